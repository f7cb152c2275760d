// Device check of the shared-reciprocal division of km_lean.h (kml_rcp_refined + kml_div_by) against IEEE division - the device's own expansion
// of `n / d` AND the host's: 1 / d for EVERY fp32 mantissa at several exponents and both signs (homography mode's s = 1 / (Z + eps)), n / d for 2^27
// random operand pairs inside kml_div_operands_ok's range (the projective map's two quotients).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I kornia_amd/csrc profiles/check_div.hip -o profiles/bin/check_div
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "km_lean.h"

__global__ void k_recip(uint32_t expo_sign, float* fast, float* dev) {
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;  // 2^23 mantissas
    const float d = __uint_as_float(expo_sign | m);
    fast[m] = kml_div_by(1.0f, d, kml_rcp_refined(d));
    dev[m] = 1.0f / d;
}
__global__ void k_quot(const float* n, const float* d, float* fast, float* dev, uint32_t count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    fast[i] = kml_div_by(n[i], d[i], kml_rcp_refined(d[i]));
    dev[i] = n[i] / d[i];
}
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 24); }

int main() {
    const uint32_t NM = 1u << 23;
    float *fast, *dev; hipMalloc(&fast, (size_t)NM * 16 * 4); hipMalloc(&dev, (size_t)NM * 16 * 4);
    float* hf = (float*)malloc((size_t)NM * 16 * 4); float* hd = (float*)malloc((size_t)NM * 16 * 4);
    long bad_fast = 0, bad_dev = 0, total = 0;
    const int expos[] = {127, 126, 128, 120, 134, 100, 150};  // 2^0, 2^-1, 2^1, 2^-7, 2^7, 2^-27, 2^23
    for (int e : expos) for (uint32_t sign = 0; sign < 2; ++sign) {
        const uint32_t es = ((uint32_t)e << 23) | (sign << 31);
        k_recip<<<NM / 256, 256>>>(es, fast, dev);
        hipMemcpy(hf, fast, (size_t)NM * 4, hipMemcpyDeviceToHost); hipMemcpy(hd, dev, (size_t)NM * 4, hipMemcpyDeviceToHost);
        long bf = 0, bd = 0;
        for (uint32_t m = 0; m < NM; ++m) {
            uint32_t u = es | m; float d; memcpy(&d, &u, 4);
            const volatile float one = 1.0f; const float ex = one / d;
            if (memcmp(&hf[m], &ex, 4)) ++bf;
            if (memcmp(&hd[m], &ex, 4)) ++bd;
        }
        printf("1 / d, exponent %4d sign %u: shared-reciprocal form %ld wrong, device IEEE %ld wrong of %u\n", e - 127, sign, bf, bd, NM);
        bad_fast += bf; bad_dev += bd; total += NM;
    }
    // n / d
    const uint32_t NQ = 1u << 27;
    float* hn = (float*)malloc((size_t)NQ * 4); float* hdd = (float*)malloc((size_t)NQ * 4);
    for (uint32_t i = 0; i < NQ; ++i) {
        // exponents within +-30 of 1 (kml_div_operands_ok allows +-40), any mantissa, any sign
        uint32_t a = rnd(), b = rnd();
        uint32_t un = ((97u + a % 61u) << 23) | (rnd() & 0x7fffffu) | ((a >> 9 & 1u) << 31);
        uint32_t ud = ((97u + b % 61u) << 23) | (rnd() & 0x7fffffu) | ((b >> 9 & 1u) << 31);
        memcpy(&hn[i], &un, 4); memcpy(&hdd[i], &ud, 4);
    }
    float *dn, *dd; hipMalloc(&dn, (size_t)NQ * 4); hipMalloc(&dd, (size_t)NQ * 4);
    hipMemcpy(dn, hn, (size_t)NQ * 4, hipMemcpyHostToDevice); hipMemcpy(dd, hdd, (size_t)NQ * 4, hipMemcpyHostToDevice);
    k_quot<<<NQ / 256, 256>>>(dn, dd, fast, dev, NQ);
    hipMemcpy(hf, fast, (size_t)NQ * 4, hipMemcpyDeviceToHost); hipMemcpy(hd, dev, (size_t)NQ * 4, hipMemcpyDeviceToHost);
    long qf = 0, qd = 0;
    for (uint32_t i = 0; i < NQ; ++i) {
        const volatile float n = hn[i]; const float ex = n / hdd[i];
        if (memcmp(&hf[i], &ex, 4)) ++qf;
        if (memcmp(&hd[i], &ex, 4)) ++qd;
    }
    printf("n / d, %u random pairs: shared-reciprocal form %ld wrong, device IEEE %ld wrong\n", NQ, qf, qd);
    printf("TOTAL 1 / d: %ld and %ld wrong of %ld;  n / d: %ld and %ld wrong of %u\n", bad_fast, bad_dev, total, qf, qd, NQ);
    return (bad_fast || bad_dev || qf || qd) ? 1 : 0;
}
