// kornia_amd - fused ColorJitter for gfx950.
//
// Reference: kornia/augmentation/_2d/intensity/color_jitter.py:126-159 applies up to four whole-tensor stages in a
// per-call order, each a chain of elementwise torch ops (8-20 launches and a full HBM round trip per stage):
//   0 brightness  clamp(x * f, 0, 1)                                     enhance/adjust.py:542-593
//   1 contrast    clamp(x * f + mean * (1 - f), 0, 1), mean = per-image mean of rgb_to_grayscale(x)   adjust.py:414-469
//   2 saturation  clamp((1 - f) * gray(x) + f * x, 0, 1)                 adjust.py:80-134, color/gray.py:60-92
//   3 hue         rgb_to_hsv -> h = fmod(h + f, 2 pi) -> hsv_to_rgb      adjust.py:179-209, color/hsv.py:27-131
// Here the whole sequence is ONE pass over the image (read x once, write y once = 2e bytes / element) computed in fp32
// registers, preceded - only when a contrast stage is present - by one reduction pass that evaluates the stages in
// front of it and accumulates the per-image gray mean in fp64 (+1e read).  Planar RGB, 4 pixels per thread.
#include "km_regtile.h"

#define KMC_MAX_STAGES 4
enum { KMC_BRIGHTNESS = 0, KMC_CONTRAST = 1, KMC_SATURATION = 2, KMC_HUE = 3 };

template <typename T>
struct KmColorArgs {
    const T* x;            // (B,3,H,W)
    T* y;                  // (B,3,H,W)
    const float* params;   // (B,4): brightness, contrast, saturation, hue shift in radians
    double* gray_sum;      // (B) fp64 accumulators for the contrast stage's mean, pre-zeroed (nullable if no contrast stage)
    const uint8_t* enable; // (4) per-stage-kind switches on the DEVICE (nullable = all on): the reference's `(factor != neutral).any()` guards
    const uint8_t* apply;  // (B) per-sample switch of the augmentation layer (nullable = all on): a sample whose entry is 0 is copied
    int stages[KMC_MAX_STAGES];
    int n_stages;
    int HW;                // pixels per plane
    uint32_t blocks_per_image, nblocks;
};

__device__ __forceinline__ float kmc_clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ float kmc_gray(float r, float g, float b) {
    // r * w_r, then addcmul twice (color/gray.py:86-89)
    float o = r * 0.299f;
    o = o + g * 0.587f;
    o = o + b * 0.114f;
    return o;
}
// torch.remainder(a, b) for b > 0 = fmod(a, b), plus b when that is negative.  fmod is exact, and for the operands
// of this kernel |a| < 2b almost always, where it reduces to one exact subtraction (Sterbenz): no libm loop.
__device__ __forceinline__ float kmc_fmod_small(float a, float b) {
    const float m = fabsf(a);
    if (m < b) return a;
    if (m < 2.0f * b) return a < 0.0f ? a + b : a - b;  // b <= |a| < 2b: exact
    return fmodf(a, b);
}
__device__ __forceinline__ float kmc_pymod(float a, float b) {
    const float r = kmc_fmod_small(a, b);
    return (r != 0.0f && r < 0.0f) ? r + b : r;
}

__device__ __forceinline__ void kmc_hue(float& r, float& g, float& b, float shift) {
    const float two_pi = 6.283185307179586f;
    // rgb_to_hsv (color/hsv.py:54-75)
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    float deltac = mx - mn;
    const float v = mx;
    const float s = deltac / (mx + 1e-8f);
    deltac = (deltac == 0.0f) ? 1.0f : deltac;
    const float rc = mx - r, gc = mx - g, bc = mx - b;
    const float h1 = bc - gc, h2 = (rc - bc) + 2.0f * deltac, h3 = (gc - rc) + 4.0f * deltac;
    float h = ((r >= g) && (r >= b)) ? h1 : ((g >= b) ? h2 : h3);
    h = h / deltac;
    h = kmc_pymod(h * (1.0f / 6.0f), 1.0f);  // scalar divisors: multiply by the reciprocal, as PyTorch's GPU div kernel does
    h = two_pi * h;
    // adjust_hue_raw (adjust.py:203-204)
    h = kmc_fmod_small(h + shift, two_pi);
    // hsv_to_rgb (color/hsv.py:104-131)
    const float hn = h * (1.0f / two_pi);
    const float h6 = hn * 6.0f;
    float hi = kmc_pymod(floorf(h6), 6.0f);
    const float f = kmc_pymod(h6, 6.0f) - hi;
    const float p = v * (1.0f - s);
    const float q = v * (1.0f - f * s);
    const float t = v * (1.0f - (1.0f - f) * s);
    int k = (int)hi;
    k = k < 0 ? 0 : (k > 5 ? 5 : k);
    r = (k == 0) ? v : (k == 1) ? q : (k == 2) ? p : (k == 3) ? p : (k == 4) ? t : v;
    g = (k == 0) ? t : (k == 1) ? v : (k == 2) ? v : (k == 3) ? q : (k == 4) ? p : p;
    b = (k == 0) ? p : (k == 1) ? p : (k == 2) ? t : (k == 3) ? v : (k == 4) ? v : q;
}

// applies stages [first, last) to one pixel; `mean` is used by the contrast stage
__device__ __forceinline__ void kmc_apply(float& r, float& g, float& b, const int (&stages)[KMC_MAX_STAGES], int first, int last,
                                          const float (&f)[4], float mean, uint32_t enable_mask) {
    for (int s = first; s < last; ++s) {
        const int st = stages[s];  // wave-uniform
        if (!((enable_mask >> st) & 1u)) continue;
        if (st == KMC_BRIGHTNESS) {
            r = kmc_clamp01(r * f[0]); g = kmc_clamp01(g * f[0]); b = kmc_clamp01(b * f[0]);
        } else if (st == KMC_CONTRAST) {
            const float off = mean * (1.0f - f[1]);
            r = kmc_clamp01(r * f[1] + off); g = kmc_clamp01(g * f[1] + off); b = kmc_clamp01(b * f[1] + off);
        } else if (st == KMC_SATURATION) {
            const float gr = kmc_gray(r, g, b), a = (1.0f - f[2]) * gr;
            r = kmc_clamp01(a + f[2] * r); g = kmc_clamp01(a + f[2] * g); b = kmc_clamp01(a + f[2] * b);
        } else {
            kmc_hue(r, g, b, f[3]);
        }
    }
}


// MODE 0: reduction pass (gray mean in front of the contrast stage);  MODE 1: apply pass.  VEC = 4 needs HW % 4 == 0
// and 4-element aligned planes; VEC = 1 otherwise.
template <typename T, int MODE, int VEC>
__global__ __launch_bounds__(256) void km_color_jitter_kernel(const KmColorArgs<T> a) {
    const uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t b = bid / a.blocks_per_image, chunk = bid % a.blocks_per_image;
    const int HW = a.HW;
    const T* xr = a.x + (size_t)b * 3 * HW;
    float f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) f[k] = a.params[(size_t)b * 4 + k];
    int ci = a.n_stages;  // index of the contrast stage
    for (int s = 0; s < a.n_stages; ++s)
        if (a.stages[s] == KMC_CONTRAST) { ci = s; break; }
    float mean = 0.0f;
    if (MODE == 1 && ci < a.n_stages) mean = (float)(a.gray_sum[b] / (double)HW);
    const int last = (MODE == 0) ? ci : a.n_stages;
    uint32_t enable_mask = 0xfu;
    if (a.enable) enable_mask = (a.enable[0] ? 1u : 0u) | (a.enable[1] ? 2u : 0u) | (a.enable[2] ? 4u : 0u) | (a.enable[3] ? 8u : 0u);
    if (a.apply && !a.apply[b]) {  // block-uniform: this sample is not jittered - every stage off, the apply pass is a copy
        if (MODE == 0) return;
        enable_mask = 0u;
    }

    float acc = 0.0f;
    const int per_block = 256 * VEC * 4;  // 4 iterations per thread
    const int start = (int)chunk * per_block;
    // (not unrolled: four inlined copies of the stage machine - exact fmod paths included - are ~45 KB of code, more than the
    // instruction cache of a CU pair holds)
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
        const int p0 = start + (it * 256 + (int)threadIdx.x) * VEC;
        if (p0 >= HW) break;
        float r[VEC], g[VEC], bl[VEC];
        if (VEC == 4) {
            float t4[4];
            km_ld4(xr + p0, t4);
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = t4[q];
            km_ld4(xr + HW + p0, t4);
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = t4[q];
            km_ld4(xr + 2 * (size_t)HW + p0, t4);
#pragma unroll
            for (int q = 0; q < 4; ++q) bl[q] = t4[q];
        } else {
            r[0] = (float)km_ld(xr + p0); g[0] = (float)km_ld(xr + HW + p0); bl[0] = (float)km_ld(xr + 2 * (size_t)HW + p0);
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            kmc_apply(r[q], g[q], bl[q], a.stages, 0, last, f, mean, enable_mask);
            if (MODE == 0) acc += kmc_gray(r[q], g[q], bl[q]);
        }
        if (MODE == 1) {
            T* yr = a.y + (size_t)b * 3 * HW;
            if (VEC == 4) {
                float t4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) t4[q] = r[q];
                km_st4(yr + p0, t4);
#pragma unroll
                for (int q = 0; q < 4; ++q) t4[q] = g[q];
                km_st4(yr + HW + p0, t4);
#pragma unroll
                for (int q = 0; q < 4; ++q) t4[q] = bl[q];
                km_st4(yr + 2 * (size_t)HW + p0, t4);
            } else {
                km_st(yr + p0, r[0]); km_st(yr + HW + p0, g[0]); km_st(yr + 2 * (size_t)HW + p0, bl[0]);
            }
        }
    }
    if (MODE == 0) {
        __shared__ double red[4];
        const double s = km_wave_sum((double)acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) km_atomic_add(a.gray_sum + b, (red[0] + red[1]) + (red[2] + red[3]));
    }
}

template <typename T>
static int kmc_run(const void* x, void* y, const void* params, double* gray_sum, const void* enable, const void* apply, const int* stages, int n_stages,
                   int B, int H, int W, hipStream_t s) {
    KmColorArgs<T> a;
    a.x = (const T*)x; a.y = (T*)y; a.params = (const float*)params; a.gray_sum = gray_sum; a.enable = (const uint8_t*)enable;
    a.apply = (const uint8_t*)apply;
    bool has_contrast = false;
    for (int k = 0; k < KMC_MAX_STAGES; ++k) {
        a.stages[k] = k < n_stages ? stages[k] : -1;
        if (k < n_stages && stages[k] == KMC_CONTRAST) has_contrast = true;
    }
    a.n_stages = n_stages;
    a.HW = H * W;
    const size_t esz = sizeof(T);
    const bool vec = (a.HW % 4 == 0) && ((uintptr_t)x % (4 * esz) == 0) && ((uintptr_t)y % (4 * esz) == 0);
    const int per_block = 256 * (vec ? 4 : 1) * 4;
    a.blocks_per_image = (uint32_t)((a.HW + per_block - 1) / per_block);
    const uint64_t nb = (uint64_t)a.blocks_per_image * (uint64_t)B;
    KM_REQUIRE(nb < (1ull << 31), "km_color_jitter: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    if (has_contrast) {
        KM_REQUIRE(gray_sum, "km_color_jitter_fwd: a contrast stage needs the gray_sum workspace");
        if (vec) hipLaunchKernelGGL((km_color_jitter_kernel<T, 0, 4>), dim3(a.nblocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((km_color_jitter_kernel<T, 0, 1>), dim3(a.nblocks), dim3(256), 0, s, a);
        const int rc = km_check_launch("km_color_jitter_fwd(mean)");
        if (rc) return rc;
    }
    if (vec) hipLaunchKernelGGL((km_color_jitter_kernel<T, 1, 4>), dim3(a.nblocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((km_color_jitter_kernel<T, 1, 1>), dim3(a.nblocks), dim3(256), 0, s, a);
    return km_check_launch("km_color_jitter_fwd");
}

static int kmc_entry(const void* x, void* y, const void* params, double* gray_sum, const void* enable, const void* apply, const int* stages,
                     int n_stages, int B, int H, int W, int dtype, void* stream) {
    if (B == 0 || H == 0 || W == 0) return 0;
    KM_REQUIRE(x && y && params, "km_color_jitter_fwd: null pointer");
    KM_REQUIRE(B > 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 30), "km_color_jitter_fwd: bad shape B=%d H=%d W=%d", B, H, W);
    KM_REQUIRE(n_stages >= 0 && n_stages <= KMC_MAX_STAGES && (n_stages == 0 || stages), "km_color_jitter_fwd: bad stage list");
    int n_contrast = 0;
    for (int k = 0; k < n_stages; ++k) {
        KM_REQUIRE(stages[k] >= 0 && stages[k] <= 3, "km_color_jitter_fwd: stage id %d out of range", stages[k]);
        n_contrast += stages[k] == KMC_CONTRAST;
    }
    KM_REQUIRE(n_contrast <= 1, "km_color_jitter_fwd: at most one contrast stage");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case KM_F32: return kmc_run<float>(x, y, params, gray_sum, enable, apply, stages, n_stages, B, H, W, s);
        case KM_BF16: return kmc_run<km_bf16>(x, y, params, gray_sum, enable, apply, stages, n_stages, B, H, W, s);
        case KM_F16: return kmc_run<km_f16>(x, y, params, gray_sum, enable, apply, stages, n_stages, B, H, W, s);
        default: km_set_error("km_color_jitter_fwd: dtype must be f32 / bf16 / f16"); return -1;
    }
}

extern "C" {

// x, y: (B,3,H,W) RGB in `dtype` (f32 / bf16 / f16); params: (B,4) fp32 on the device - brightness factor, contrast
// factor, saturation factor, hue shift in RADIANS; stages: HOST array of n_stages (<= 4) ids in application order
// (0 brightness, 1 contrast, 2 saturation, 3 hue; at most one contrast stage); gray_sum: (B) fp64 device workspace,
// zeroed by the caller, required iff a contrast stage is present; enable: (4) uint8 on the DEVICE indexed by stage id,
// 0 = skip every stage of that kind (the reference's `(factor != neutral).any()` guards without a host sync), nullable.
int km_color_jitter_fwd(const void* x, void* y, const void* params, double* gray_sum, const void* enable, const int* stages,
                        int n_stages, int B, int H, int W, int dtype, void* stream) {
    return kmc_entry(x, y, params, gray_sum, enable, nullptr, stages, n_stages, B, H, W, dtype, stream);
}

// ... with the augmentation layer's per-sample switch (kornia/augmentation/base.py:348-393): apply (B) uint8 on the device, a sample
// whose entry is 0 is copied unchanged (and does not enter the contrast mean pass)
int km_color_jitter_fwd_masked(const void* x, void* y, const void* params, double* gray_sum, const void* enable, const void* apply,
                               const int* stages, int n_stages, int B, int H, int W, int dtype, void* stream) {
    return kmc_entry(x, y, params, gray_sum, enable, apply, stages, n_stages, B, H, W, dtype, stream);
}

}  // extern "C"
