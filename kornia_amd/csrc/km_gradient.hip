// kornia_amd - spatial_gradient / sobel kernels for gfx950.
//
// Reference: kornia/filters/sobel.py:59-72 (replicate pad + conv2d of (B*C,1,H,W) with the
// (n_out,1,k,k) derivative stack -> (B,C,n_out,H,W)) and :164-171 (sobel magnitude
// sqrt(gx*gx + gy*gy + eps)).  One launch computes every output channel from one set of taps
// (9 or 25 loads served by L1) and, for sobel(), fuses the magnitude so the (B,C,2,H,W) stack is
// never written.  fma chain in (p,q) order == oracle/ko_impl.h ko_spatial_gradient_fwd.
#include <stdlib.h>

#include "km_regtile.h"

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
__device__ __forceinline__ typename KmTraits<T>::R km_sg_adjoint_pixel(const KmGradArgs<T>& a, uint32_t bc, int py, int px) {
    typedef typename KmTraits<T>::R R;
    const int pd = a.kS / 2;
    const size_t plane = (size_t)a.H * a.W;
    const int sx_lo = (px == 0) ? -pd : px, sx_hi = (px == a.W - 1) ? a.W - 1 + pd : px;
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
    return acc;
}

template <typename T>
__global__ __launch_bounds__(256) void km_spatial_gradient_bwd_kernel(const KmGradArgs<T> a) {
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int px = tx * 64 + (threadIdx.x & 63);
    const int py0 = ty * 16 + (threadIdx.x >> 6) * 4;
    if (px >= a.W) return;
    const size_t plane = (size_t)a.H * a.W;
    for (int r = 0; r < 4; ++r) {
        const int py = py0 + r;
        if (py >= a.H) break;
        km_st(a.out + (size_t)bc * plane + (size_t)py * a.W + px, km_sg_adjoint_pixel<T>(a, bc, py, px));
    }
}

// replicate padding folds every padded coordinate outside the image onto the border pixels, so the adjoint differs from
// "correlate the zero-extended gradients with the rotated stacks" only on the 1-pixel frame: recomputed here exactly
template <typename T>
__global__ __launch_bounds__(256) void km_spatial_gradient_bwd_frame_kernel(const KmGradArgs<T> a, uint32_t frame_px, uint32_t blocks_per_plane) {
    const uint32_t bc = blockIdx.x / blocks_per_plane;
    const uint32_t idx = (blockIdx.x % blocks_per_plane) * 256u + threadIdx.x;
    if (idx >= frame_px) return;
    int py, px;
    if (idx < 2u * (uint32_t)a.W) {
        px = (int)(idx % (uint32_t)a.W);
        py = idx < (uint32_t)a.W ? 0 : a.H - 1;
    } else {
        const uint32_t i2 = idx - 2u * (uint32_t)a.W;
        py = 1 + (int)(i2 >> 1);
        px = (i2 & 1u) ? a.W - 1 : 0;
    }
    km_st(a.out + (size_t)bc * a.H * a.W + (size_t)py * a.W + px, km_sg_adjoint_pixel<T>(a, bc, py, px));
}

// ------------------------------------------------------------------------------------------------
// Register-tiled forward for the shapes the reference produces (3x3 or 5x5 stacks, 2 or 3 outputs): the same
// organisation as the separable blur (km_blur_fast.hip) - a lane owns 4 adjacent columns (one 16-byte load per
// input row + PD clamped scalar halo loads per side, 16-byte stores), a wave walks a strip of rows keeping the last
// KS input rows in registers.  Replicate border = clamped row / column addresses.  Each output keeps the
// reference's fma chain in (p, q) order over the full stack (zero taps included: 0 * inf must stay NaN).
// HBM traffic = read x once + write n_out (or 1 for the fused magnitude) planes = the algorithmic (1 + n_out) e.
#define KM_SG_ROWS 32


template <typename T, int KS, int NOUT>
__global__ __launch_bounds__(256) void km_spatial_gradient_reg_kernel(const KmGradArgs<T> a) {
    constexpr int PD = KS / 2, NV = 4 + 2 * PD;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tbx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t tby = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx = (int)tbx * 64 + lane;                     // column group (4 px)
    const int r0 = ((int)tby * 4 + wave) * KM_SG_ROWS;       // first output row of this wave's strip
    const int H = a.H, W = a.W;
    if (gx * 4 >= W || r0 >= H) return;
    const int c0 = gx * 4;
    const size_t plane = (size_t)H * W;
    const T* img = a.x + (size_t)bc * plane;
    // clamped halo columns (replicate border)
    int hl[PD], hr[PD];
#pragma unroll
    for (int q = 0; q < PD; ++q) {
        hl[q] = max(c0 - PD + q, 0);
        hr[q] = min(c0 + 4 + q, W - 1);
    }
    float k[NOUT][KS][KS];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int p = 0; p < KS; ++p)
#pragma unroll
            for (int q = 0; q < KS; ++q) k[o][p][q] = a.kern[(o * KS + p) * KS + q];

    float ring[KS][NV];  // last KS input rows: ring[.][i] = column c0 - PD + i
    const int n_rows = (r0 + KM_SG_ROWS <= H ? KM_SG_ROWS : H - r0);
    const int total = n_rows + KS - 1;
    for (int it0 = 0; it0 < total; it0 += KS) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int it = it0 + kk;
            if (it < total) {
                const int rin = min(max(r0 - PD + it, 0), H - 1);
                const T* rowp = img + (size_t)rin * W;
                float o4[4];
                km_ld4(rowp + c0, o4);
#pragma unroll
                for (int q = 0; q < PD; ++q) {
                    ring[kk][q] = (float)km_ld(rowp + hl[q]);
                    ring[kk][PD + 4 + q] = (float)km_ld(rowp + hr[q]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) ring[kk][PD + q] = o4[q];
                if (it >= KS - 1) {
                    const int r = r0 + it - (KS - 1);
                    float acc[NOUT][4];
#pragma unroll
                    for (int o = 0; o < NOUT; ++o)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float s = 0.f;
#pragma unroll
                            for (int p = 0; p < KS; ++p)
#pragma unroll
                                for (int q = 0; q < KS; ++q) s = km_fma(k[o][p][q], ring[(kk + 1 + p) % KS][c + q], s);
                            acc[o][c] = s;
                        }
                    if (a.out) {
#pragma unroll
                        for (int o = 0; o < NOUT; ++o) km_st4(a.out + ((size_t)bc * NOUT + o) * plane + (size_t)r * W + c0, acc[o]);
                    }
                    if (NOUT == 2 && a.mag) {
                        float mg[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) mg[c] = km_sqrt((acc[0][c] * acc[0][c] + acc[1][c] * acc[1][c]) + a.eps);
                        km_st4(a.mag + (size_t)bc * plane + (size_t)r * W + c0, mg);
                    }
                }
            }
        }
    }
}


// interior of the adjoint: gx = sum_o correlate(zero-extended gout_o, k_o rotated by 180 degrees), register-tiled like the forward
template <typename T, int KS, int NOUT>
__global__ __launch_bounds__(256) void km_spatial_gradient_bwd_reg_kernel(const KmGradArgs<T> a) {
    constexpr int PD = KS / 2, NV = 4 + 2 * PD;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tbx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t tby = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx = (int)tbx * 64 + lane;
    const int r0 = ((int)tby * 4 + wave) * KM_SG_ROWS;
    const int H = a.H, W = a.W;
    if (gx * 4 >= W || r0 >= H) return;
    const int c0 = gx * 4;
    const size_t plane = (size_t)H * W;
    // halo columns: zero outside the image
    int hl[PD], hr[PD];
    bool okl[PD], okr[PD];
#pragma unroll
    for (int q = 0; q < PD; ++q) {
        const int il = c0 - PD + q, ir = c0 + 4 + q;
        okl[q] = il >= 0; hl[q] = okl[q] ? il : 0;
        okr[q] = ir < W; hr[q] = okr[q] ? ir : 0;
    }
    float k[NOUT][KS][KS];  // rotated by 180 degrees
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int p = 0; p < KS; ++p)
#pragma unroll
            for (int q = 0; q < KS; ++q) k[o][p][q] = a.kern[(o * KS + (KS - 1 - p)) * KS + (KS - 1 - q)];

    float ring[NOUT][KS][NV];
    const int n_rows = (r0 + KM_SG_ROWS <= H ? KM_SG_ROWS : H - r0);
    const int total = n_rows + KS - 1;
    for (int it0 = 0; it0 < total; it0 += KS) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int it = it0 + kk;
            if (it < total) {
                const int rin = r0 - PD + it;
                const bool rv = (rin >= 0 && rin < H);
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    if (rv) {
                        const T* rowp = a.gout + ((size_t)bc * NOUT + o) * plane + (size_t)rin * W;
                        float o4[4];
                        km_ld4(rowp + c0, o4);
#pragma unroll
                        for (int q = 0; q < PD; ++q) {
                            const float vl = (float)km_ld(rowp + hl[q]), vr = (float)km_ld(rowp + hr[q]);
                            ring[o][kk][q] = okl[q] ? vl : 0.f;
                            ring[o][kk][PD + 4 + q] = okr[q] ? vr : 0.f;
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) ring[o][kk][PD + q] = o4[q];
                    } else {
#pragma unroll
                        for (int q = 0; q < NV; ++q) ring[o][kk][q] = 0.f;
                    }
                }
                if (it >= KS - 1) {
                    const int r = r0 + it - (KS - 1);
                    float acc[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float sacc = 0.f;
#pragma unroll
                        for (int o = 0; o < NOUT; ++o)
#pragma unroll
                            for (int p = 0; p < KS; ++p)
#pragma unroll
                                for (int q = 0; q < KS; ++q) sacc = km_fma(k[o][p][q], ring[o][(kk + 1 + p) % KS][c + q], sacc);
                        acc[c] = sacc;
                    }
                    km_st4(a.out + (size_t)bc * plane + (size_t)r * W + c0, acc);
                }
            }
        }
    }
}

template <typename T>
static int km_sg_fast_bwd_launch(const KmGradArgs<T>& a0, hipStream_t s) {
    KmGradArgs<T> a = a0;
    a.tiles_x = (uint32_t)((a.W / 4 + 63) / 64);
    a.tiles_y = (uint32_t)((a.H + 4 * KM_SG_ROWS - 1) / (4 * KM_SG_ROWS));
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)a.BC;
    KM_REQUIRE(nb < (1ull << 31), "km_spatial_gradient: grid too large");
    a.nblocks = (uint32_t)nb;
    if (a.kS == 3 && a.n_out == 2)
        hipLaunchKernelGGL((km_spatial_gradient_bwd_reg_kernel<T, 3, 2>), dim3(a.nblocks), dim3(256), 0, s, a);
    else if (a.kS == 3)
        hipLaunchKernelGGL((km_spatial_gradient_bwd_reg_kernel<T, 3, 3>), dim3(a.nblocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((km_spatial_gradient_bwd_reg_kernel<T, 5, 3>), dim3(a.nblocks), dim3(256), 0, s, a);
    int rc = km_check_launch("km_spatial_gradient_bwd(reg)");
    if (rc) return rc;
    const uint64_t frame = 2ull * a.W + 2ull * (uint64_t)(a.H - 2);
    const uint32_t bpp = (uint32_t)((frame + 255) / 256);
    hipLaunchKernelGGL(km_spatial_gradient_bwd_frame_kernel<T>, dim3(bpp * (uint32_t)a.BC), dim3(256), 0, s, a, (uint32_t)frame, bpp);
    return km_check_launch("km_spatial_gradient_bwd(frame)");
}
template <>
int km_sg_fast_bwd_launch<double>(const KmGradArgs<double>&, hipStream_t) { return -1; }

// 1 if the register-tiled forward handles this problem
template <typename T>
static bool km_sg_fast_ok(const void* x, const void* out, const void* mag, int H, int W, int n_out, int kS) {
    if (!((kS == 3 && (n_out == 2 || n_out == 3)) || (kS == 5 && n_out == 3))) return false;
    if ((W & 3) != 0 || W < 8 || H < 1) return false;
    const size_t al = 4 * sizeof(T);
    if (((uintptr_t)x % al) || (out && ((uintptr_t)out % al)) || (mag && ((uintptr_t)mag % al))) return false;
    return (((size_t)H * W * sizeof(T)) % al) == 0;
}

template <typename T>
static int km_sg_fast_launch(const KmGradArgs<T>& a0, hipStream_t s) {
    KmGradArgs<T> a = a0;
    a.tiles_x = (uint32_t)((a.W / 4 + 63) / 64);
    a.tiles_y = (uint32_t)((a.H + 4 * KM_SG_ROWS - 1) / (4 * KM_SG_ROWS));
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)a.BC;
    KM_REQUIRE(nb < (1ull << 31), "km_spatial_gradient: grid too large");
    a.nblocks = (uint32_t)nb;
    if (a.kS == 3 && a.n_out == 2)
        hipLaunchKernelGGL((km_spatial_gradient_reg_kernel<T, 3, 2>), dim3(a.nblocks), dim3(256), 0, s, a);
    else if (a.kS == 3)
        hipLaunchKernelGGL((km_spatial_gradient_reg_kernel<T, 3, 3>), dim3(a.nblocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((km_spatial_gradient_reg_kernel<T, 5, 3>), dim3(a.nblocks), dim3(256), 0, s, a);
    return km_check_launch("km_spatial_gradient_fwd(reg)");
}
template <>
int km_sg_fast_launch<double>(const KmGradArgs<double>&, hipStream_t) { return -1; }  // fp64 stays on the generic kernel

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
    {
        const int generic = km_config().sg_generic;  // KM_SG_ALGO=generic: A/B timing
        if (!bwd && !generic && sizeof(T) != 8 && km_sg_fast_ok<T>(x, out, mag, H, W, n_out, kS)) return km_sg_fast_launch<T>(a, s);
        if (bwd && !generic && sizeof(T) != 8 && H > 2 && km_sg_fast_ok<T>(gout, out, nullptr, H, W, n_out, kS)) return km_sg_fast_bwd_launch<T>(a, s);
    }
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
