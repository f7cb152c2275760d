// kornia_amd - small kernels of the augmentation layer's fast path: the pieces between the
// reference's parameter dictionaries and the hot kernels that cost it a dozen tiny launches each.
//
//   km_gaussian_taps_fwd   per-sample sigma (B,2) -> normalised 1-D Gaussian taps (B,kx), (B,ky) in one launch
//                          (kornia/filters/kernels.py:77-120 `gaussian`: x = arange(k) - k // 2 (+ 0.5 if k is even),
//                          g = exp(-x^2 / (2 sigma^2)), g / sum(g); called by gaussian_blur2d, kornia/filters/gaussian.py:111-114,
//                          with sigma[:, 1] for the horizontal and sigma[:, 0] for the vertical kernel) - the reference spends
//                          ~8 elementwise launches per axis on B x 5 numbers;
//   km_select_samples_fwd  out[b] = apply[b] ? transformed[b] : original[b] - the per-sample probability blend of
//                          _AugmentationBase.transform_inputs (kornia/augmentation/base.py:348-393, `torch.where` on a
//                          broadcast mask), reading only the side that is kept: 2e bytes per element instead of 3e.
#include "km_regtile.h"

// taps rounded to the image's storage type (what filter2d's cast of its kernel to the input dtype does, kornia/filters/filter.py:126), kept as fp32
__device__ __forceinline__ float km_taps_round(float v, int round_dtype) {
    if (round_dtype == KM_BF16) return __uint_as_float(((uint32_t)km_f32_to_bf16_bits(v)) << 16);
    if (round_dtype == KM_F16) return km_round_as(v, (const km_f16*)nullptr);
    return v;
}
// per_axis: sigma is (B,2) = (sigma_y, sigma_x); else (B): one sigma for both axes.  apply (B) uint8 / prob (B) fp32 (applied when > 0.5):
// at most one of them; round_dtype: KM_F32 (none) / KM_BF16 / KM_F16.
template <int MAXK>
__global__ __launch_bounds__(64) void km_gaussian_taps_kernel(const float* __restrict__ sigma, const uint8_t* __restrict__ apply, float* __restrict__ taps_x,
                                                              float* __restrict__ taps_y, int B, int kx, int ky, int per_axis = 1,
                                                              const float* __restrict__ prob = nullptr, int round_dtype = KM_F32) {
    const int t = blockIdx.x * 64 + threadIdx.x;  // one thread per (sample, axis)
    if (t >= 2 * B) return;
    const int b = t >> 1, axis = t & 1;           // axis 0: horizontal taps from sigma[:, 1]; axis 1: vertical from sigma[:, 0]
    const int k = axis ? ky : kx;
    const float s = per_axis ? sigma[(size_t)b * 2 + (axis ? 0 : 1)] : sigma[b];
    float* out = (axis ? taps_y : taps_x) + (size_t)b * k;
    if ((apply && !apply[b]) || (prob && !(prob[b] > 0.5f))) {
        // a sample that is not blurred gets the identity kernel: 1 * x + 0 * neighbours reproduces a finite image bit for bit, so the
        // probability blend of the augmentation layer costs no pass of its own (odd kernel sizes; the caller blends otherwise)
        for (int i = 0; i < k; ++i) out[i] = (i == k / 2) ? 1.0f : 0.0f;
        return;
    }
    const float mean = (float)(k / 2);
    const float den = 2.0f * (s * s);
    if (den == 0.0f) {
        // sigma == 0: the reference refuses it on the host (kornia/filters/gaussian.py:103-108, a device-to-host read this path does not
        // make); here the formula would give 0 / 0 taps and a NaN image.  The limit of the Gaussian for sigma -> 0 is what is emitted
        // instead: the identity kernel (odd sizes), two taps of one half (even sizes).  A NaN sigma still gives NaN taps.
        for (int i = 0; i < k; ++i) out[i] = (k & 1) ? ((i == k / 2) ? 1.0f : 0.0f) : ((i == k / 2 - 1 || i == k / 2) ? 0.5f : 0.0f);
        return;
    }
    float g[MAXK];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
        if (i < k) {
            float x = (float)i - mean;
            if ((k & 1) == 0) x = x + 0.5f;
            g[i] = expf(-(x * x) / den);
            sum = sum + g[i];
        }
    }
#pragma unroll
    for (int i = 0; i < MAXK; ++i)
        if (i < k) out[i] = km_taps_round(g[i] / sum, round_dtype);
}

template <typename T>
__global__ __launch_bounds__(256) void km_select_samples_kernel(const T* __restrict__ transformed, const T* __restrict__ original, const uint8_t* __restrict__ apply,
                                                                T* __restrict__ out, uint32_t chunks_per_sample, size_t n_per_sample, int vec) {
    const uint32_t b = blockIdx.x / chunks_per_sample, chunk = blockIdx.x % chunks_per_sample;
    const T* __restrict__ src = (apply[b] ? transformed : original) + (size_t)b * n_per_sample;  // block-uniform
    T* __restrict__ dst = out + (size_t)b * n_per_sample;
    if (vec) {
        const size_t i = ((size_t)chunk * 256 + threadIdx.x) * 4;
        if (i < n_per_sample) {
            float v[4];
            km_ld4(src + i, v);
            km_st4(dst + i, v);
        }
    } else {
        const size_t i0 = (size_t)chunk * 1024 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i = i0 + (size_t)k * 256;
            if (i < n_per_sample) dst[i] = src[i];
        }
    }
}

template <typename T>
static int km_select_run(const void* transformed, const void* original, const void* apply, void* out, int B, size_t n, hipStream_t s) {
    const bool vec = (n % 4 == 0) && ((uintptr_t)transformed % (4 * sizeof(T)) == 0) && ((uintptr_t)original % (4 * sizeof(T)) == 0) &&
                     ((uintptr_t)out % (4 * sizeof(T)) == 0);
    const uint64_t chunks = (n + 1023) / 1024;
    KM_REQUIRE(chunks * (uint64_t)B < (1ull << 31), "km_select_samples_fwd: grid too large");
    hipLaunchKernelGGL(km_select_samples_kernel<T>, dim3((uint32_t)(chunks * B)), dim3(256), 0, s, (const T*)transformed, (const T*)original,
                       (const uint8_t*)apply, (T*)out, (uint32_t)chunks, n, vec ? 1 : 0);
    return km_check_launch("km_select_samples_fwd");
}

// ColorJitter's sampled factors -> what km_color_jitter_fwd reads, in one launch (one block): the (B,4) parameter table with the
// hue factor turned from turns into radians (x float(2 pi), color_jitter.py:147), the four stage switches of the module's
// `(factor != neutral).any()` guards (color_jitter.py:137-148: brightness != 0, contrast != 1, saturation != 1, hue != 0) and,
// when the parameters carry a probability draw, the per-sample switch `batch_prob > 0.5` (augmentation/base.py:380).
__global__ __launch_bounds__(256) void km_color_params_kernel(const float* bf, const float* cf, const float* sf, const float* hf, const float* prob,
                                                              float* params, uint8_t* enable, uint8_t* apply, int B, double* gray_sum = nullptr) {
    int any_b = 0, any_c = 0, any_s = 0, any_h = 0;
    for (int b = threadIdx.x; b < B; b += 256) {
        if (gray_sum) gray_sum[b] = 0.0;  // (km_color_params_ws_fwd: the contrast stage's accumulators start at zero - no fill launch of the caller's)
        const float vb = bf[b], vc = cf[b], vs = sf[b], vh = hf[b];
        params[4 * b + 0] = vb; params[4 * b + 1] = vc; params[4 * b + 2] = vs; params[4 * b + 3] = vh * 6.283185307179586f;
        any_b |= (vb != 0.0f); any_c |= (vc != 1.0f); any_s |= (vs != 1.0f); any_h |= (vh != 0.0f);
        if (apply) apply[b] = (prob[b] > 0.5f) ? 1 : 0;
    }
    any_b = __syncthreads_or(any_b); any_c = __syncthreads_or(any_c); any_s = __syncthreads_or(any_s); any_h = __syncthreads_or(any_h);
    if (threadIdx.x == 0) { enable[0] = any_b ? 1 : 0; enable[1] = any_c ? 1 : 0; enable[2] = any_s ? 1 : 0; enable[3] = any_h ? 1 : 0; }
}

extern "C" {

// sigma (B,2) fp32 on the device, (sigma_y, sigma_x) per sample like gaussian_blur2d's argument; taps_x (B,kx), taps_y (B,ky) fp32;
// apply (B) uint8 on the device, nullable: a sample whose entry is 0 gets the identity kernel (needs odd kx, ky).
int km_gaussian_taps_fwd(const void* sigma, const void* apply, void* taps_x, void* taps_y, int B, int kx, int ky, void* stream) {
    if (B == 0) return 0;
    KM_REQUIRE(sigma && taps_x && taps_y, "km_gaussian_taps_fwd: null pointer");
    KM_REQUIRE(B > 0 && kx > 0 && ky > 0 && kx <= 64 && ky <= 64, "km_gaussian_taps_fwd: bad sizes B=%d kx=%d ky=%d (1..64)", B, kx, ky);
    KM_REQUIRE(!apply || ((kx & 1) && (ky & 1)), "km_gaussian_taps_fwd: the per-sample switch needs odd kernel sizes (%d, %d)", kx, ky);
    const dim3 grid((2 * B + 63) / 64);
    if (kx <= 8 && ky <= 8)
        hipLaunchKernelGGL(km_gaussian_taps_kernel<8>, grid, dim3(64), 0, (hipStream_t)stream, (const float*)sigma, (const uint8_t*)apply, (float*)taps_x, (float*)taps_y, B, kx, ky,
                           1, (const float*)nullptr, (int)KM_F32);
    else
        hipLaunchKernelGGL(km_gaussian_taps_kernel<64>, grid, dim3(64), 0, (hipStream_t)stream, (const float*)sigma, (const uint8_t*)apply, (float*)taps_x, (float*)taps_y, B, kx, ky,
                           1, (const float*)nullptr, (int)KM_F32);
    return km_check_launch("km_gaussian_taps_fwd");
}

// km_gaussian_taps_fwd for the augmentation layer's own call (RandomGaussianBlur.apply_transform: ONE sigma per sample, an image of a 16-bit
// type): sigma (B,2) [per_axis != 0] or (B) [per_axis == 0] fp32; batch_prob (B) fp32 on the device, nullable - the layer's probability draw,
// thresholded here (> 0.5: blurred; else the identity kernel, odd sizes); round_dtype: the image's dtype code - the taps are rounded to it
// (and stay fp32 values), which is what the reference's cast of its kernels to the input dtype does (kornia/filters/filter.py:126).  Replaces
// the expand / contiguous copy of sigma, the comparison and the two cast round trips the host layer otherwise spends ATen launches on.
int km_gaussian_taps_dtype_fwd(const void* sigma, int per_axis, const void* batch_prob, void* taps_x, void* taps_y, int B, int kx, int ky, int round_dtype,
                               void* stream) {
    if (B == 0) return 0;
    KM_REQUIRE(sigma && taps_x && taps_y, "km_gaussian_taps_dtype_fwd: null pointer");
    KM_REQUIRE(B > 0 && kx > 0 && ky > 0 && kx <= 64 && ky <= 64, "km_gaussian_taps_dtype_fwd: bad sizes B=%d kx=%d ky=%d (1..64)", B, kx, ky);
    KM_REQUIRE(!batch_prob || ((kx & 1) && (ky & 1)), "km_gaussian_taps_dtype_fwd: the per-sample switch needs odd kernel sizes (%d, %d)", kx, ky);
    KM_REQUIRE(round_dtype == KM_F32 || round_dtype == KM_BF16 || round_dtype == KM_F16, "km_gaussian_taps_dtype_fwd: round_dtype must be f32 / bf16 / f16");
    const dim3 grid((2 * B + 63) / 64);
    if (kx <= 8 && ky <= 8)
        hipLaunchKernelGGL(km_gaussian_taps_kernel<8>, grid, dim3(64), 0, (hipStream_t)stream, (const float*)sigma, (const uint8_t*)nullptr, (float*)taps_x, (float*)taps_y, B, kx, ky,
                           per_axis, (const float*)batch_prob, round_dtype);
    else
        hipLaunchKernelGGL(km_gaussian_taps_kernel<64>, grid, dim3(64), 0, (hipStream_t)stream, (const float*)sigma, (const uint8_t*)nullptr, (float*)taps_x, (float*)taps_y, B, kx, ky,
                           per_axis, (const float*)batch_prob, round_dtype);
    return km_check_launch("km_gaussian_taps_dtype_fwd");
}

// transformed, original, out: (B, n_per_sample) elements of `dtype`; apply: (B) uint8 on the device (non-zero: keep the transformed sample).
int km_select_samples_fwd(const void* transformed, const void* original, const void* apply, void* out, int B, long long n_per_sample, int dtype,
                          void* stream) {
    if (B == 0 || n_per_sample == 0) return 0;
    KM_REQUIRE(transformed && original && apply && out, "km_select_samples_fwd: null pointer");
    KM_REQUIRE(B > 0 && n_per_sample > 0, "km_select_samples_fwd: bad sizes B=%d n=%lld", B, n_per_sample);
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case KM_F32: return km_select_run<float>(transformed, original, apply, out, B, (size_t)n_per_sample, s);
        case KM_BF16: return km_select_run<km_bf16>(transformed, original, apply, out, B, (size_t)n_per_sample, s);
        case KM_F16: return km_select_run<km_f16>(transformed, original, apply, out, B, (size_t)n_per_sample, s);
        default: km_set_error("km_select_samples_fwd: dtype must be f32 / bf16 / f16"); return -1;
    }
}

// brightness / contrast / saturation / hue factors (B) fp32 on the device as ColorJitter's generator samples them (hue in turns),
// batch_prob (B) fp32 or NULL -> params (B,4) fp32 (hue in radians), enable (4) uint8, apply (B) uint8 (iff batch_prob is given).
int km_color_params_fwd(const void* brightness, const void* contrast, const void* saturation, const void* hue, const void* batch_prob, void* params,
                        void* enable, void* apply, int B, void* stream) {
    KM_REQUIRE(B >= 0 && enable && (B == 0 || (brightness && contrast && saturation && hue && params)), "km_color_params_fwd: null pointer");
    KM_REQUIRE((batch_prob == nullptr) == (apply == nullptr) || B == 0, "km_color_params_fwd: batch_prob and apply go together");
    hipLaunchKernelGGL(km_color_params_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)brightness, (const float*)contrast,
                       (const float*)saturation, (const float*)hue, (const float*)batch_prob, (float*)params, (uint8_t*)enable, (uint8_t*)apply, B);
    return km_check_launch("km_color_params_fwd");
}

// km_color_params_fwd that also ZEROES gray_sum (B) fp64, the workspace km_color_jitter_fwd's contrast stage accumulates the per-image gray
// sums in (its caller would otherwise spend a fill launch on it).
int km_color_params_ws_fwd(const void* brightness, const void* contrast, const void* saturation, const void* hue, const void* batch_prob, void* params,
                           void* enable, void* apply, void* gray_sum, int B, void* stream) {
    KM_REQUIRE(B >= 0 && enable && (B == 0 || (brightness && contrast && saturation && hue && params && gray_sum)), "km_color_params_ws_fwd: null pointer");
    KM_REQUIRE((batch_prob == nullptr) == (apply == nullptr) || B == 0, "km_color_params_ws_fwd: batch_prob and apply go together");
    hipLaunchKernelGGL(km_color_params_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)brightness, (const float*)contrast,
                       (const float*)saturation, (const float*)hue, (const float*)batch_prob, (float*)params, (uint8_t*)enable, (uint8_t*)apply, B, (double*)gray_sum);
    return km_check_launch("km_color_params_ws_fwd");
}

}  // extern "C"
