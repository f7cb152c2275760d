// kornia_amd - spatial_gradient / sobel kernels for gfx950.
//
// Reference: kornia/filters/sobel.py:59-72 (replicate pad + conv2d of (B*C,1,H,W) with the
// (n_out,1,k,k) derivative stack -> (B,C,n_out,H,W)) and :164-171 (sobel magnitude
// sqrt(gx*gx + gy*gy + eps)).  One launch computes every output channel from one set of taps
// (9 or 25 loads served by L1) and, for sobel(), fuses the magnitude so the (B,C,2,H,W) stack is
// never written.  fma chain in (p,q) order == oracle/ko_impl.h ko_spatial_gradient_fwd.
#include "km_common.h"

#define KM_SG_MAX_K 5
#define KM_SG_MAX_OUT 3

template <typename T>
struct KmGradArgs {
    typedef typename KmTraits<T>::R R;
    const T* x;      // fwd: (B*C,H,W)
    const T* gout;   // bwd: (B*C,n_out,H,W)
    T* out;          // fwd: (B*C,n_out,H,W) nullable ; bwd: (B*C,H,W)
    T* mag;          // fwd: (B*C,H,W) nullable
    R kern[KM_SG_MAX_OUT * KM_SG_MAX_K * KM_SG_MAX_K];  // taps by value (tiny)
    R eps;
    int BC, H, W, n_out, kS;
    uint32_t tiles_x, tiles_y, nblocks;
};

__device__ __forceinline__ int km_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <typename T>
__global__ __launch_bounds__(256) void km_spatial_gradient_fwd_kernel(const KmGradArgs<T> a) {
    typedef typename KmTraits<T>::R R;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int j = tx * 64 + (threadIdx.x & 63);
    const int i0 = ty * 16 + (threadIdx.x >> 6) * 4;
    if (j >= a.W) return;
    const int pd = a.kS / 2;
    const size_t plane = (size_t)a.H * a.W;
    const T* img = a.x + (size_t)bc * plane;
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + r;
        if (i >= a.H) break;
        R acc[KM_SG_MAX_OUT] = {0, 0, 0};
        for (int p = 0; p < a.kS; ++p) {
            const int sy = km_clampi(i + p - pd, 0, a.H - 1);  // replicate border
            for (int q = 0; q < a.kS; ++q) {
                const int sx = km_clampi(j + q - pd, 0, a.W - 1);
                const R v = km_ld(img + (size_t)sy * a.W + sx);
#pragma unroll
                for (int o = 0; o < KM_SG_MAX_OUT; ++o)
                    if (o < a.n_out) acc[o] = km_fma(a.kern[(o * a.kS + p) * a.kS + q], v, acc[o]);
            }
        }
        if (a.out) {
#pragma unroll
            for (int o = 0; o < KM_SG_MAX_OUT; ++o)
                if (o < a.n_out) km_st(a.out + ((size_t)bc * a.n_out + o) * plane + (size_t)i * a.W + j, acc[o]);
        }
        if (a.mag) km_st(a.mag + (size_t)bc * plane + (size_t)i * a.W + j, km_sqrt((acc[0] * acc[0] + acc[1] * acc[1]) + a.eps));
    }
}

// adjoint wrt input (gather form): gx[p] = sum_o sum_{s in pre(p)} sum_{t} k_o[t] * gout_o[s + pd - t]
// replicate pre-images: p itself, plus every s < 0 for p == 0 and every s >= n for p == n-1.
template <typename T>
__global__ __launch_bounds__(256) void km_spatial_gradient_bwd_kernel(const KmGradArgs<T> a) {
    typedef typename KmTraits<T>::R R;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int px = tx * 64 + (threadIdx.x & 63);
    const int py0 = ty * 16 + (threadIdx.x >> 6) * 4;
    if (px >= a.W) return;
    const int pd = a.kS / 2;
    const size_t plane = (size_t)a.H * a.W;
    const int sx_lo = (px == 0) ? -pd : px, sx_hi = (px == a.W - 1) ? a.W - 1 + pd : px;
    for (int r = 0; r < 4; ++r) {
        const int py = py0 + r;
        if (py >= a.H) break;
        const int sy_lo = (py == 0) ? -pd : py, sy_hi = (py == a.H - 1) ? a.H - 1 + pd : py;
        R acc = 0;
        for (int o = 0; o < a.n_out; ++o) {
            const T* go = a.gout + ((size_t)bc * a.n_out + o) * plane;
            for (int sy = sy_lo; sy <= sy_hi; ++sy)
                for (int sx = sx_lo; sx <= sx_hi; ++sx)
                    for (int p = 0; p < a.kS; ++p) {
                        const int oy = sy + pd - p;
                        if (oy < 0 || oy >= a.H) continue;
                        for (int q = 0; q < a.kS; ++q) {
                            const int ox = sx + pd - q;
                            if (ox < 0 || ox >= a.W) continue;
                            acc = km_fma(a.kern[(o * a.kS + p) * a.kS + q], (R)km_ld(go + (size_t)oy * a.W + ox), acc);
                        }
                    }
        }
        km_st(a.out + (size_t)bc * plane + (size_t)py * a.W + px, acc);
    }
}

template <typename T>
static int km_grad_run(bool bwd, const void* x, const void* gout, const void* kern_host, void* out, void* mag, int BC, int H,
                       int W, int n_out, int kS, double eps, hipStream_t s) {
    typedef typename KmTraits<T>::R R;
    KmGradArgs<T> a;
    a.x = (const T*)x; a.gout = (const T*)gout; a.out = (T*)out; a.mag = (T*)mag;
    const R* kh = (const R*)kern_host;
    for (int t = 0; t < n_out * kS * kS; ++t) a.kern[t] = kh[t];
    a.eps = (R)eps; a.BC = BC; a.H = H; a.W = W; a.n_out = n_out; a.kS = kS;
    a.tiles_x = (W + 63) / 64;
    a.tiles_y = (H + 15) / 16;
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)BC;
    KM_REQUIRE(nb < (1ull << 31), "km_spatial_gradient: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    if (bwd)
        hipLaunchKernelGGL(km_spatial_gradient_bwd_kernel<T>, dim3(a.nblocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(km_spatial_gradient_fwd_kernel<T>, dim3(a.nblocks), dim3(256), 0, s, a);
    return km_check_launch(bwd ? "km_spatial_gradient_bwd" : "km_spatial_gradient_fwd");
}

static int km_grad_validate(const char* fn, int B, int C, int H, int W, int n_out, int kS, int dtype) {
    KM_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0, "%s: bad shape", fn);
    KM_REQUIRE((int64_t)H * W < (1ll << 31), "%s: image plane exceeds 2^31 elements", fn);
    KM_REQUIRE(n_out >= 1 && n_out <= KM_SG_MAX_OUT, "%s: n_out must be 1..3, got %d", fn, n_out);
    KM_REQUIRE(kS >= 1 && kS <= KM_SG_MAX_K && (kS & 1), "%s: kernel size must be odd and <= 5, got %d", fn, kS);
    KM_REQUIRE(dtype >= 0 && dtype <= 3, "%s: bad dtype", fn);
    return 0;
}

extern "C" {

// kern_host: HOST pointer to the (n_out,kS,kS) derivative stack in the compute dtype (fp32 / fp64 for
// f64 data) - it is passed to the kernel by value.  out: (B,C,n_out,H,W) or null; mag: (B,C,H,W) or
// null (sobel magnitude, needs n_out == 2).
int km_spatial_gradient_fwd(const void* x, const void* kern_host, void* out, void* mag, int B, int C, int H, int W,
                            int n_out, int kS, double eps, int dtype, void* stream) {
    if (B == 0 || C == 0) return 0;
    if (km_grad_validate("km_spatial_gradient_fwd", B, C, H, W, n_out, kS, dtype)) return -1;
    KM_REQUIRE(x && kern_host && (out || mag), "km_spatial_gradient_fwd: null pointer");
    KM_REQUIRE(!mag || n_out == 2, "km_spatial_gradient_fwd: magnitude needs n_out == 2");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case KM_F32: return km_grad_run<float>(false, x, nullptr, kern_host, out, mag, B * C, H, W, n_out, kS, eps, s);
        case KM_F64: return km_grad_run<double>(false, x, nullptr, kern_host, out, mag, B * C, H, W, n_out, kS, eps, s);
        case KM_BF16: return km_grad_run<km_bf16>(false, x, nullptr, kern_host, out, mag, B * C, H, W, n_out, kS, eps, s);
        default: return km_grad_run<km_f16>(false, x, nullptr, kern_host, out, mag, B * C, H, W, n_out, kS, eps, s);
    }
}

// gout: (B,C,n_out,H,W) -> gx: (B,C,H,W)
int km_spatial_gradient_bwd(const void* gout, const void* kern_host, void* gx, int B, int C, int H, int W, int n_out,
                            int kS, int dtype, void* stream) {
    if (B == 0 || C == 0) return 0;
    if (km_grad_validate("km_spatial_gradient_bwd", B, C, H, W, n_out, kS, dtype)) return -1;
    KM_REQUIRE(gout && kern_host && gx, "km_spatial_gradient_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case KM_F32: return km_grad_run<float>(true, nullptr, gout, kern_host, gx, nullptr, B * C, H, W, n_out, kS, 0.0, s);
        case KM_F64: return km_grad_run<double>(true, nullptr, gout, kern_host, gx, nullptr, B * C, H, W, n_out, kS, 0.0, s);
        case KM_BF16: return km_grad_run<km_bf16>(true, nullptr, gout, kern_host, gx, nullptr, B * C, H, W, n_out, kS, 0.0, s);
        default: return km_grad_run<km_f16>(true, nullptr, gout, kern_host, gx, nullptr, B * C, H, W, n_out, kS, 0.0, s);
    }
}

}  // extern "C"
