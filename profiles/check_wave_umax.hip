#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "km_common.h"
__global__ void k(const uint32_t* in, uint32_t* out) {
    const uint32_t v = in[blockIdx.x * 64 + threadIdx.x];
    const uint32_t m = km_wave_umax_last(v);
    if (threadIdx.x == 63) out[blockIdx.x] = m;
}
int main() {
    const int N = 4096;
    uint32_t* h = (uint32_t*)malloc(N * 64 * 4); uint32_t* o = (uint32_t*)malloc(N * 4);
    srand(1);
    for (int i = 0; i < N * 64; ++i) h[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
    for (int b = 0; b < 64; ++b) { for (int l = 0; l < 64; ++l) h[b * 64 + l] = 5; h[b * 64 + b] = 1000000 + b; }  // the maximum in every lane position once
    uint32_t *di, *dout; hipMalloc(&di, N * 64 * 4); hipMalloc(&dout, N * 4);
    hipMemcpy(di, h, N * 64 * 4, hipMemcpyHostToDevice);
    k<<<N, 64>>>(di, dout);
    hipMemcpy(o, dout, N * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < N; ++b) { uint32_t m = 0; for (int l = 0; l < 64; ++l) m = h[b * 64 + l] > m ? h[b * 64 + l] : m; if (m != o[b]) ++bad; }
    printf("km_wave_umax_last: %d of %d waves wrong\n", bad, N);
    return bad != 0;
}
