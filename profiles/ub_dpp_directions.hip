#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* o) {
    int v = threadIdx.x * 10;
    int a = __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);  // wave_shl:1
    int b = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);  // wave_shr:1
    int c = __builtin_amdgcn_update_dpp(v, v, 0x101, 0xf, 0xf, false);  // row_shl:1
    int d = __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    o[threadIdx.x] = a; o[64 + threadIdx.x] = b; o[128 + threadIdx.x] = c; o[192 + threadIdx.x] = d;
}
int main() { int* d; hipMalloc(&d, 1024); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); int h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  const char* nm[4] = {"wave_shl:1", "wave_shr:1", "row_shl:1", "row_shr:1"}; for (int k = 0; k < 4; ++k) { printf("%s:", nm[k]); for (int i = 0; i < 20; ++i) printf(" %d", h[64 * k + i]); printf(" ... %d %d\n", h[64 * k + 62], h[64 * k + 63]); } return 0; }
