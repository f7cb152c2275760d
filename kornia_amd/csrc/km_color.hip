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

// ---- 16-bit storage: the hue round trip WITHIN THE TOLERANCE THE STORAGE TYPE HAS (round 6) ------------------------------------------
// kmc_hue above is the reference's operation sequence rounding for rounding (two IEEE divisions, three remainders with their exact fmod
// paths, a 15-way select): ~120 vector instructions of the ~200 a pixel costs, for a result that a bf16 / f16 image then rounds to 8 / 11
// bits and that BASELINE.json compares at 1e-2 - config 3's colour pass sat on its issue time (58 us of the sequence's 213, 2.6 TB/s).
// The same function of (r, g, b, shift) with the arithmetic a GPU wants: v_rcp_f32 for the two quotients (1 ulp), hue kept in TURNS so
// that both reductions are x - floor(x), and the standard branch-free HSV -> RGB  c_n = v - v s clamp(min(k, 4 - k), 0, 1),
// k = (n + 6 h) mod 6, n = 5 / 3 / 1 for r / g / b.  ~45 instructions; within ~1e-6 of kmc_hue before the storage rounding (a different
// 16-bit neighbour at ties: tests/test_gpu_color.py compares at the storage type's 1e-2).  fp32 storage keeps kmc_hue.
#ifndef KMC_FAST16
#define KMC_FAST16 1
#endif
__device__ __forceinline__ float kmc_rcp(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(v);
#else
    return 1.0f / v;
#endif
}
__device__ __forceinline__ void kmc_hue_fast(float& r, float& g, float& b, float shift) {
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    const float delta = mx - mn, v = mx;
    const float vs = v * (delta * kmc_rcp(mx + 1e-8f));             // v * s
    const float id = kmc_rcp(delta == 0.0f ? 1.0f : delta);
    const float rc = mx - r, gc = mx - g, bc = mx - b;
    float h6 = ((r >= g) && (r >= b)) ? (bc - gc) * id : ((g >= b) ? km_fma(rc - bc, id, 2.0f) : km_fma(gc - rc, id, 4.0f));  // [-1, 5]
    float ht = km_fma(h6, 1.0f / 6.0f, shift * 0.15915494309189535f);  // hue + shift in turns
    ht = ht - floorf(ht);                                              // [0, 1)
    h6 = ht * 6.0f;
    float k, m;
    k = h6 + 5.0f; k = k >= 6.0f ? k - 6.0f : k; m = kmc_clamp01(fminf(k, 4.0f - k)); r = km_fma(-vs, m, v);
    k = h6 + 3.0f; k = k >= 6.0f ? k - 6.0f : k; m = kmc_clamp01(fminf(k, 4.0f - k)); g = km_fma(-vs, m, v);
    k = h6 + 1.0f; k = k >= 6.0f ? k - 6.0f : k; m = kmc_clamp01(fminf(k, 4.0f - k)); b = km_fma(-vs, m, v);
}

// applies stages [first, last) to one pixel; `mean` is used by the contrast stage.  FAST: kmc_hue_fast (16-bit storage)
template <bool FAST = false>
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
            if (FAST) kmc_hue_fast(r, g, b, f[3]);
            else kmc_hue(r, g, b, f[3]);
        }
    }
}


template <typename T>
__device__ __forceinline__ float kmc_storage_round(float v) {  // v as it reads back from the storage dtype
    T t;
    km_st(&t, v);
    return (float)km_ld(&t);
}

// MODE 0: the pass in front of a contrast stage - applies the stages that precede it, accumulates the per-image gray sum and, when
// there are such stages, stores the intermediate image in y (storage dtype);  MODE 1: the apply pass - without a contrast stage
// the whole chain from x; with one, the contrast stage and what follows it, from the intermediate in y (in place: a thread reads
// and writes the same pixels), so that every stage is evaluated once (the hue round trip through HSV is ~200 VALU instructions
// per pixel: evaluating it in both passes made the 224 x 224 bf16 pass ALU-bound at 0.45 TB/s).
// VEC = 4 needs HW % 4 == 0 and 4-element aligned planes; VEC = 1 otherwise.
template <typename T, int MODE, int VEC>
__global__ __launch_bounds__(256) void km_color_jitter_kernel(const KmColorArgs<T> a) {
    const uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t b = bid / a.blocks_per_image, chunk = bid % a.blocks_per_image;
    const int HW = a.HW;
    const T* xr = a.x + (size_t)b * 3 * HW;
    T* yr = a.y + (size_t)b * 3 * HW;
    float f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) f[k] = a.params[(size_t)b * 4 + k];
    int ci = a.n_stages;  // index of the contrast stage
    for (int s = 0; s < a.n_stages; ++s)
        if (a.stages[s] == KMC_CONTRAST) { ci = s; break; }
    const bool has_contrast = ci < a.n_stages;
    float mean = 0.0f;
    if (MODE == 1 && has_contrast) mean = (float)(a.gray_sum[b] / (double)HW);
    uint32_t enable_mask = 0xfu;
    if (a.enable) enable_mask = (a.enable[0] ? 1u : 0u) | (a.enable[1] ? 2u : 0u) | (a.enable[2] ? 4u : 0u) | (a.enable[3] ? 8u : 0u);
    bool skipped = false;
    if (a.apply && !a.apply[b]) {  // block-uniform: this sample is not jittered - every stage off, the apply pass is a copy of x
        if (MODE == 0) return;
        enable_mask = 0u;
        skipped = true;
    }
    // MODE 0 applies [0, ci) ; MODE 1 applies [ci, n) of the intermediate when MODE 0 stored one, else [0, n) of x
    const bool staged = has_contrast && ci > 0 && !skipped;
    const int first = (MODE == 1 && staged) ? ci : 0;
    const int last = (MODE == 0) ? ci : a.n_stages;
    const T* in = (MODE == 1 && staged) ? (const T*)yr : xr;

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
        if constexpr (VEC == 4) {
            km_ld4(in + p0, r);
            km_ld4(in + HW + p0, g);
            km_ld4(in + 2 * (size_t)HW + p0, bl);
        } else {
            r[0] = (float)km_ld(in + p0); g[0] = (float)km_ld(in + HW + p0); bl[0] = (float)km_ld(in + 2 * (size_t)HW + p0);
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            kmc_apply<(KMC_FAST16 && sizeof(T) == 2)>(r[q], g[q], bl[q], a.stages, first, last, f, mean, enable_mask);
            if (MODE == 0) {
                if (ci > 0) {  // the gray sum is taken of the values the apply pass will read back: the stored ones
                    r[q] = kmc_storage_round<T>(r[q]); g[q] = kmc_storage_round<T>(g[q]); bl[q] = kmc_storage_round<T>(bl[q]);
                }
                acc += kmc_gray(r[q], g[q], bl[q]);
            }
        }
        if (MODE == 1 || ci > 0) {
            if constexpr (VEC == 4) {
                km_st4(yr + p0, r);
                km_st4(yr + HW + p0, g);
                km_st4(yr + 2 * (size_t)HW + p0, bl);
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

// ------------------------------------------------------------------------------------------------
// Backward (the reference's ops are differentiable: kornia/enhance/adjust.py:80-593 through autograd).  One thread recomputes the
// stage chain of its pixel, keeps every stage's input in registers and walks the stages in reverse with the transposed Jacobians
// that autograd applies to the reference's op sequence:
//   clamp            grad * (0 <= u <= 1)                                    (ATen clamp_backward: bounds inclusive)
//   brightness       gx = f g m ;  df = sum g m x
//   saturation       gx_k = (1 - f) w_k sum_c(g m)_c + f (g m)_k ;  df = sum (g m)_c (x_c - gray)
//   contrast         gx_k = f (g m)_k + (1 - f) w_k S / HW,  S = sum over the IMAGE of (g m) - the mean couples every pixel, so a first
//                    pass (MODE 0) back-propagates down to the contrast stage and accumulates S per image (fp64), the second pass
//                    (MODE 1) uses it ;  df = sum (g m)_c (x_c - mean)
//   hue              the 3 x 4 Jacobian d(r', g', b') / d(r, g, b, shift) by forward-mode differentiation of kmc_hue's own operation
//                    sequence (KmcD: value + 4 tangents) with autograd's rules: amax / amin split the gradient evenly among ties,
//                    where() passes it to the selected side, floor / remainder-by-constant / fmod have slope 0 / 1 / 1
struct KmcD {
    float v, d[4];
};
__device__ __forceinline__ KmcD kd_c(float v) { return {v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ KmcD kd_add(const KmcD& a, const KmcD& b) { return {a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2], a.d[3] + b.d[3]}}; }
__device__ __forceinline__ KmcD kd_sub(const KmcD& a, const KmcD& b) { return {a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2], a.d[3] - b.d[3]}}; }
__device__ __forceinline__ KmcD kd_scale(const KmcD& a, float c) { return {a.v * c, {a.d[0] * c, a.d[1] * c, a.d[2] * c, a.d[3] * c}}; }
__device__ __forceinline__ KmcD kd_mul(const KmcD& a, const KmcD& b) {
    return {a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2], a.d[3] * b.v + a.v * b.d[3]}};
}
__device__ __forceinline__ KmcD kd_div(const KmcD& a, const KmcD& b) {
    const float q = a.v / b.v, ib = 1.0f / b.v;
    return {q, {(a.d[0] - q * b.d[0]) * ib, (a.d[1] - q * b.d[1]) * ib, (a.d[2] - q * b.d[2]) * ib, (a.d[3] - q * b.d[3]) * ib}};
}
__device__ __forceinline__ KmcD kd_one_minus(const KmcD& a) { return {1.0f - a.v, {-a.d[0], -a.d[1], -a.d[2], -a.d[3]}}; }

// J[c][k] = d out_c / d (r, g, b, shift)_k of kmc_hue at (r, g, b)
__device__ __forceinline__ void kmc_hue_jacobian(float r, float g, float b, float shift, float (&J)[3][4]) {
    const float two_pi = 6.283185307179586f;
    const KmcD R = {r, {1.f, 0.f, 0.f, 0.f}}, G = {g, {0.f, 1.f, 0.f, 0.f}}, B = {b, {0.f, 0.f, 1.f, 0.f}};
    const float mxv = fmaxf(r, fmaxf(g, b)), mnv = fminf(r, fminf(g, b));
    const float xr = (r == mxv) ? 1.f : 0.f, xg = (g == mxv) ? 1.f : 0.f, xb = (b == mxv) ? 1.f : 0.f, xc = 1.0f / fmaxf(xr + xg + xb, 1.0f);
    const float nr = (r == mnv) ? 1.f : 0.f, ng = (g == mnv) ? 1.f : 0.f, nb = (b == mnv) ? 1.f : 0.f, nc = 1.0f / fmaxf(nr + ng + nb, 1.0f);
    const KmcD MX = {mxv, {xr * xc, xg * xc, xb * xc, 0.f}}, MN = {mnv, {nr * nc, ng * nc, nb * nc, 0.f}};
    const KmcD delta = kd_sub(MX, MN);
    const KmcD V = MX;
    KmcD mxe = MX;
    mxe.v = MX.v + 1e-8f;
    const KmcD S = kd_div(delta, mxe);
    const KmcD dc = (delta.v == 0.0f) ? kd_c(1.0f) : delta;
    const KmcD rc = kd_sub(MX, R), gc = kd_sub(MX, G), bc = kd_sub(MX, B);
    const KmcD h1 = kd_sub(bc, gc), h2 = kd_add(kd_sub(rc, bc), kd_scale(dc, 2.0f)), h3 = kd_add(kd_sub(gc, rc), kd_scale(dc, 4.0f));
    KmcD h = ((r >= g) && (r >= b)) ? h1 : ((g >= b) ? h2 : h3);
    h = kd_div(h, dc);
    h = kd_scale(h, 1.0f / 6.0f);
    h.v = kmc_pymod(h.v, 1.0f);
    h = kd_scale(h, two_pi);
    h.v = kmc_fmod_small(h.v + shift, two_pi);
    h.d[3] = h.d[3] + 1.0f;
    const KmcD h6 = kd_scale(kd_scale(h, 1.0f / two_pi), 6.0f);
    const float hi = kmc_pymod(floorf(h6.v), 6.0f);
    KmcD f = h6;
    f.v = kmc_pymod(h6.v, 6.0f) - hi;
    const KmcD p = kd_mul(V, kd_one_minus(S));
    const KmcD q = kd_mul(V, kd_one_minus(kd_mul(f, S)));
    const KmcD t = kd_mul(V, kd_one_minus(kd_mul(kd_one_minus(f), S)));
    int k = (int)hi;
    k = k < 0 ? 0 : (k > 5 ? 5 : k);
    const KmcD& orr = (k == 0) ? V : (k == 1) ? q : (k == 2) ? p : (k == 3) ? p : (k == 4) ? t : V;
    const KmcD& og = (k == 0) ? t : (k == 1) ? V : (k == 2) ? V : (k == 3) ? q : (k == 4) ? p : p;
    const KmcD& ob = (k == 0) ? p : (k == 1) ? p : (k == 2) ? t : (k == 3) ? V : (k == 4) ? V : q;
#pragma unroll
    for (int j = 0; j < 4; ++j) { J[0][j] = orr.d[j]; J[1][j] = og.d[j]; J[2][j] = ob.d[j]; }
}

template <typename T>
struct KmColorBwdArgs {
    const T* x;             // (B,3,H,W) forward input
    const T* gy;            // (B,3,H,W) gradient wrt the output
    T* gx;                  // (B,3,H,W) gradient wrt the input
    const float* params;    // (B,4)
    const double* gray_sum; // (B) the forward's gray sums (contrast stage only)
    double* gsum;           // (B) fp64 workspace, pre-zeroed: S of the contrast stage
    double* gparams;        // (B,4) fp64 accumulators, pre-zeroed: gradient wrt the parameters (nullable)
    const uint8_t* enable;
    const uint8_t* apply;
    int stages[KMC_MAX_STAGES];
    int n_stages;
    int HW;
    uint32_t blocks_per_image, nblocks;
};

__device__ __forceinline__ float kmc_pass01(float u) { return ((u >= 0.0f) && (u <= 1.0f)) ? 1.0f : 0.0f; }

// MODE 0: S of the contrast stage;  MODE 1: the gradients
template <typename T, int MODE, int VEC>
__global__ __launch_bounds__(256) void km_color_jitter_bwd_kernel(const KmColorBwdArgs<T> a) {
    const uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t b = bid / a.blocks_per_image, chunk = bid % a.blocks_per_image;
    const int HW = a.HW;
    const size_t img = (size_t)b * 3 * HW;
    float f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) f[k] = a.params[(size_t)b * 4 + k];
    uint32_t enable_mask = 0xfu;
    if (a.enable) enable_mask = (a.enable[0] ? 1u : 0u) | (a.enable[1] ? 2u : 0u) | (a.enable[2] ? 4u : 0u) | (a.enable[3] ? 8u : 0u);
    if (a.apply && !a.apply[b]) {  // block-uniform: the forward copied this sample
        if (MODE == 0) return;
        enable_mask = 0u;
    }
    int st[KMC_MAX_STAGES];  // stage kinds in application order, -1 = none / disabled
    bool has_contrast = false;
#pragma unroll
    for (int s = 0; s < KMC_MAX_STAGES; ++s) {
        const int k = (s < a.n_stages) ? a.stages[s] : -1;
        st[s] = (k >= 0 && ((enable_mask >> k) & 1u)) ? k : -1;
        has_contrast = has_contrast || (st[s] == KMC_CONTRAST);
    }
    if (MODE == 0 && !has_contrast) return;
    float mean = 0.0f, s_over_hw = 0.0f;
    if (has_contrast) {
        mean = (float)(a.gray_sum[b] / (double)HW);
        if (MODE == 1) s_over_hw = (float)(a.gsum[b] / (double)HW);
    }
    const float wgt[3] = {0.299f, 0.587f, 0.114f};
    float acc_s = 0.0f, gf[4] = {0.f, 0.f, 0.f, 0.f};
    const int per_block = 256 * VEC * 4;
    const int start = (int)chunk * per_block;
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
        const int p0 = start + (it * 256 + (int)threadIdx.x) * VEC;
        if (p0 >= HW) break;
        float xin[3][VEC], g[3][VEC];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if constexpr (VEC == 4) {
                km_ld4(a.x + img + (size_t)c * HW + p0, xin[c]);
                km_ld4(a.gy + img + (size_t)c * HW + p0, g[c]);
            } else {
                xin[c][0] = (float)km_ld(a.x + img + (size_t)c * HW + p0);
                g[c][0] = (float)km_ld(a.gy + img + (size_t)c * HW + p0);
            }
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            // forward: the input of every stage
            float in[KMC_MAX_STAGES][3];
            float r = xin[0][q], gg = xin[1][q], bb = xin[2][q];
#pragma unroll
            for (int s = 0; s < KMC_MAX_STAGES; ++s) {
                in[s][0] = r; in[s][1] = gg; in[s][2] = bb;
                if (st[s] == KMC_BRIGHTNESS) {
                    r = kmc_clamp01(r * f[0]); gg = kmc_clamp01(gg * f[0]); bb = kmc_clamp01(bb * f[0]);
                } else if (st[s] == KMC_CONTRAST) {
                    const float off = mean * (1.0f - f[1]);
                    r = kmc_clamp01(r * f[1] + off); gg = kmc_clamp01(gg * f[1] + off); bb = kmc_clamp01(bb * f[1] + off);
                } else if (st[s] == KMC_SATURATION) {
                    const float gr = kmc_gray(r, gg, bb), aa = (1.0f - f[2]) * gr;
                    r = kmc_clamp01(aa + f[2] * r); gg = kmc_clamp01(aa + f[2] * gg); bb = kmc_clamp01(aa + f[2] * bb);
                } else if (st[s] == KMC_HUE) {
                    kmc_hue(r, gg, bb, f[3]);
                }
            }
            // backward
            float gr3[3] = {g[0][q], g[1][q], g[2][q]};
            bool done = false;
#pragma unroll
            for (int s = KMC_MAX_STAGES - 1; s >= 0; --s) {
                if (done) continue;
                const float x0 = in[s][0], x1 = in[s][1], x2 = in[s][2];
                if (st[s] == KMC_BRIGHTNESS) {
                    const float m0 = gr3[0] * kmc_pass01(x0 * f[0]), m1 = gr3[1] * kmc_pass01(x1 * f[0]), m2 = gr3[2] * kmc_pass01(x2 * f[0]);
                    gf[0] += (m0 * x0 + m1 * x1) + m2 * x2;
                    gr3[0] = m0 * f[0]; gr3[1] = m1 * f[0]; gr3[2] = m2 * f[0];
                } else if (st[s] == KMC_CONTRAST) {
                    const float off = mean * (1.0f - f[1]);
                    const float m0 = gr3[0] * kmc_pass01(x0 * f[1] + off), m1 = gr3[1] * kmc_pass01(x1 * f[1] + off), m2 = gr3[2] * kmc_pass01(x2 * f[1] + off);
                    if (MODE == 0) {
                        acc_s += (m0 + m1) + m2;
                        done = true;
                    } else {
                        gf[1] += (m0 * (x0 - mean) + m1 * (x1 - mean)) + m2 * (x2 - mean);
                        const float spread = (1.0f - f[1]) * s_over_hw;
                        gr3[0] = f[1] * m0 + wgt[0] * spread; gr3[1] = f[1] * m1 + wgt[1] * spread; gr3[2] = f[1] * m2 + wgt[2] * spread;
                    }
                } else if (st[s] == KMC_SATURATION) {
                    const float gray = kmc_gray(x0, x1, x2), aa = (1.0f - f[2]) * gray;
                    const float m0 = gr3[0] * kmc_pass01(aa + f[2] * x0), m1 = gr3[1] * kmc_pass01(aa + f[2] * x1), m2 = gr3[2] * kmc_pass01(aa + f[2] * x2);
                    gf[2] += (m0 * (x0 - gray) + m1 * (x1 - gray)) + m2 * (x2 - gray);
                    const float sm = (1.0f - f[2]) * ((m0 + m1) + m2);
                    gr3[0] = wgt[0] * sm + f[2] * m0; gr3[1] = wgt[1] * sm + f[2] * m1; gr3[2] = wgt[2] * sm + f[2] * m2;
                } else if (st[s] == KMC_HUE) {
                    float J[3][4];
                    kmc_hue_jacobian(x0, x1, x2, f[3], J);
                    const float o0 = (gr3[0] * J[0][0] + gr3[1] * J[1][0]) + gr3[2] * J[2][0];
                    const float o1 = (gr3[0] * J[0][1] + gr3[1] * J[1][1]) + gr3[2] * J[2][1];
                    const float o2 = (gr3[0] * J[0][2] + gr3[1] * J[1][2]) + gr3[2] * J[2][2];
                    gf[3] += (gr3[0] * J[0][3] + gr3[1] * J[1][3]) + gr3[2] * J[2][3];
                    gr3[0] = o0; gr3[1] = o1; gr3[2] = o2;
                }
            }
            g[0][q] = gr3[0]; g[1][q] = gr3[1]; g[2][q] = gr3[2];
        }
        if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if constexpr (VEC == 4) km_st4(a.gx + img + (size_t)c * HW + p0, g[c]);
                else km_st(a.gx + img + (size_t)c * HW + p0, g[c][0]);
            }
        }
    }
    __shared__ double red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (MODE == 0) {
        const double sw = km_wave_sum((double)acc_s);
        if (lane == 0) red[0][wave] = sw;
        __syncthreads();
        if (threadIdx.x == 0) km_atomic_add(a.gsum + b, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    } else if (a.gparams) {  // kernel-uniform
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double sw = km_wave_sum((double)gf[k]);
            if (lane == 0) red[k][wave] = sw;
        }
        __syncthreads();
        if (threadIdx.x < 4) km_atomic_add(a.gparams + (size_t)b * 4 + threadIdx.x, (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]));
    }
}

template <typename T>
static int kmc_bwd_run(const void* x, const void* gy, void* gx, const void* params, const double* gray_sum, double* gsum, double* gparams, const void* enable,
                       const void* apply, const int* stages, int n_stages, int B, int H, int W, hipStream_t s) {
    KmColorBwdArgs<T> a;
    a.x = (const T*)x; a.gy = (const T*)gy; a.gx = (T*)gx; a.params = (const float*)params; a.gray_sum = gray_sum; a.gsum = gsum; a.gparams = gparams;
    a.enable = (const uint8_t*)enable; a.apply = (const uint8_t*)apply;
    bool has_contrast = false;
    for (int k = 0; k < KMC_MAX_STAGES; ++k) {
        a.stages[k] = k < n_stages ? stages[k] : -1;
        if (k < n_stages && stages[k] == KMC_CONTRAST) has_contrast = true;
    }
    a.n_stages = n_stages;
    a.HW = H * W;
    const size_t esz = sizeof(T);
    const bool vec = (a.HW % 4 == 0) && ((uintptr_t)x % (4 * esz) == 0) && ((uintptr_t)gy % (4 * esz) == 0) && ((uintptr_t)gx % (4 * esz) == 0);
    const int per_block = 256 * (vec ? 4 : 1) * 4;
    a.blocks_per_image = (uint32_t)((a.HW + per_block - 1) / per_block);
    const uint64_t nb = (uint64_t)a.blocks_per_image * (uint64_t)B;
    KM_REQUIRE(nb < (1ull << 31), "km_color_jitter_bwd: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    if (has_contrast) {
        KM_REQUIRE(gray_sum && gsum, "km_color_jitter_bwd: a contrast stage needs the forward's gray sums and the gsum workspace");
        if (vec) hipLaunchKernelGGL((km_color_jitter_bwd_kernel<T, 0, 4>), dim3(a.nblocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((km_color_jitter_bwd_kernel<T, 0, 1>), dim3(a.nblocks), dim3(256), 0, s, a);
        const int rc = km_check_launch("km_color_jitter_bwd(contrast sum)");
        if (rc) return rc;
    }
    if (vec) hipLaunchKernelGGL((km_color_jitter_bwd_kernel<T, 1, 4>), dim3(a.nblocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((km_color_jitter_bwd_kernel<T, 1, 1>), dim3(a.nblocks), dim3(256), 0, s, a);
    return km_check_launch("km_color_jitter_bwd");
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

// Backward of km_color_jitter_fwd(_masked): gx = d loss / d x given gy = d loss / d y, same stage list / params / enable / apply as the
// forward.  gray_sum: (B) fp64, the forward's workspace AFTER the forward ran (the per-image gray sums) - required iff a contrast
// stage is present, as is gsum: (B) fp64 workspace zeroed by the caller.  gparams: (B,4) fp64 accumulators zeroed by the caller,
// receives d loss / d params (hue in radians), or NULL.  Reference: autograd through kornia/enhance/adjust.py:80-593.
int km_color_jitter_bwd(const void* x, const void* gy, void* gx, const void* params, const double* gray_sum, double* gsum, double* gparams,
                        const void* enable, const void* apply, const int* stages, int n_stages, int B, int H, int W, int dtype, void* stream) {
    if (B == 0 || H == 0 || W == 0) return 0;
    KM_REQUIRE(x && gy && gx && params, "km_color_jitter_bwd: null pointer");
    KM_REQUIRE(B > 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 30), "km_color_jitter_bwd: bad shape B=%d H=%d W=%d", B, H, W);
    KM_REQUIRE(n_stages >= 0 && n_stages <= KMC_MAX_STAGES && (n_stages == 0 || stages), "km_color_jitter_bwd: bad stage list");
    int n_contrast = 0;
    for (int k = 0; k < n_stages; ++k) {
        KM_REQUIRE(stages[k] >= 0 && stages[k] <= 3, "km_color_jitter_bwd: stage id %d out of range", stages[k]);
        n_contrast += stages[k] == KMC_CONTRAST;
    }
    KM_REQUIRE(n_contrast <= 1, "km_color_jitter_bwd: at most one contrast stage");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case KM_F32: return kmc_bwd_run<float>(x, gy, gx, params, gray_sum, gsum, gparams, enable, apply, stages, n_stages, B, H, W, s);
        case KM_BF16: return kmc_bwd_run<km_bf16>(x, gy, gx, params, gray_sum, gsum, gparams, enable, apply, stages, n_stages, B, H, W, s);
        case KM_F16: return kmc_bwd_run<km_f16>(x, gy, gx, params, gray_sum, gsum, gparams, enable, apply, stages, n_stages, B, H, W, s);
        default: km_set_error("km_color_jitter_bwd: dtype must be f32 / bf16 / f16"); return -1;
    }
}

}  // extern "C"
