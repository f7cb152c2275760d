// kornia_amd - image pyramid kernels (reference: kornia/geometry/transform/pyramid.py).
//
//   km_pyrdown_fwd          pyrdown (pyramid.py:409-453) in ONE pass:  blur with the fixed 5x5 binomial kernel / 256
//                           (filter2d, any border mode) followed by the bilinear resize to (oh, ow).  The blurred image is
//                           never written: HBM traffic = read x once + write y once = (1 + 1/f^2) e bytes per input
//                           element instead of (3 + 1/f^2) e for blur + resize as two kernels (f = 2: 1.25 e vs 3.25 e).
//   km_resize_bilinear_fwd  F.interpolate(mode='bilinear') with ATen's arithmetic (the resize leg of pyrup, :494-496).
//
// Arithmetic (bit-identical to oracle/ko_impl.h ko_filter2d_fwd + ko_resize_bilinear_fwd):
//   * blur: fma chain over the 5x5 taps in (p, q) order from 0, taps r[p]*r[q]/256 (exact in binary floating point);
//     the blurred value is rounded to the storage dtype before it is interpolated, as the unfused pipeline does;
//   * resize: scale = align ? (in-1)/(out-1) : in/out ; src = align ? scale*d : max(scale*(d+0.5)-0.5, 0) ;
//     i0 = (int)src ; i1 = i0 + (i0 < in-1) ; l1 = src - i0 ; l0 = 1 - l1 ;
//     out = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)   (no contraction: the library is built with -ffp-contract=off).
//
// Two mappings.  The pyramid case proper (factor 2, align_corners = False, even H, W % 4 == 0: every interpolation weight
// is exactly 1/2 and every blurred pixel is used exactly once) runs register-tiled like the 5x5 filter2d kernel
// (km_filter2d_fast.hip): a lane owns 4 adjacent input columns = 2 output columns, a wave walks a 32-row strip keeping the
// last 5 input rows in registers, every second blurred row is combined with the previous one and stored (one 16-byte load
// per input row and lane, one 8-byte store per output row and lane; 25 fma per input pixel).  Everything else (odd sizes,
// other factors, align_corners = True) takes the general mapping: one lane per OUTPUT pixel, a wave covers 64 adjacent output columns (for f = 2 its 6 loads per window row
// cover one contiguous 520-byte span of the input row; the 6x6 window overlaps between neighbours are served by L1/L2),
// a 256-thread workgroup covers a 64 x 4 output tile, tiles of one image stay on one XCD (km_xcd_remap).
#include <stdlib.h>

#include "km_regtile.h"



// value as the storage dtype would hold it
__device__ __forceinline__ float kmp_round(float v, const float*) { return v; }
__device__ __forceinline__ double kmp_round(double v, const double*) { return v; }
__device__ __forceinline__ float kmp_round(float v, const km_bf16*) { return __uint_as_float(((uint32_t)km_f32_to_bf16_bits(v)) << 16); }
__device__ __forceinline__ float kmp_round(float v, const km_f16*) { KM_OPAQUE(v); return (float)(km_f16)v; }

// ATen area_pixel_compute_scale / area_pixel_compute_source_index (bilinear: negative sources clamp to 0)
template <typename R>
__device__ __forceinline__ void kmp_axis(int d, int n_in, int n_out, int align, int& i0, int& i1, R& l0, R& l1) {
    R src;
    if (align) {
        const R scale = n_out > 1 ? (R)(n_in - 1) / (R)(n_out - 1) : (R)0;
        src = scale * (R)d;
    } else {
        const R scale = (R)n_in / (R)n_out;
        src = scale * ((R)d + (R)0.5) - (R)0.5;
        if (src < (R)0) src = (R)0;
    }
    i0 = (int)src;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = src - (R)i0;
    l0 = (R)1 - l1;
}

template <typename T>
struct KmPyrArgs {
    const T* x;
    T* y;
    int H, W, oh, ow, border, align;
    uint32_t tiles_x, tiles_y, nblocks;
};

template <typename T>
__device__ __forceinline__ bool kmp_tile(const KmPyrArgs<T>& a, int& ox, int& oy, uint32_t& bc) {
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tbx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t tby = bid % a.tiles_y;
    bc = bid / a.tiles_y;
    ox = (int)tbx * 64 + (int)(threadIdx.x & 63);
    oy = (int)tby * 4 + (int)(threadIdx.x >> 6);
    return ox < a.ow && oy < a.oh;
}

template <typename T>
__global__ __launch_bounds__(256) void km_pyrdown_kernel(const KmPyrArgs<T> a) {
    typedef typename KmTraits<T>::R R;
    int ox, oy;
    uint32_t bc;
    if (!kmp_tile(a, ox, oy, bc)) return;
    const int H = a.H, W = a.W;
    const T* img = a.x + (size_t)bc * H * W;

    int y0, y1, x0, x1;
    R h0, h1, w0, w1;
    kmp_axis<R>(oy, H, a.oh, a.align, y0, y1, h0, h1);
    kmp_axis<R>(ox, W, a.ow, a.align, x0, x1, w0, w1);
    const int dy = y1 - y0, dx = x1 - x0;  // 0 or 1

    // 6 x 6 input window: rows y0-2 .. y0+3, columns x0-2 .. x0+3, through the border map (-1: zero padding)
    int cm[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) cm[c] = km_border_map(x0 - 2 + c, W, a.border);
    R v[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int rm = km_border_map(y0 - 2 + r, H, a.border);
        const T* rowp = img + (size_t)(rm < 0 ? 0 : rm) * W;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const R val = (R)km_ld(rowp + (cm[c] < 0 ? 0 : cm[c]));
            v[r][c] = (rm >= 0 && cm[c] >= 0) ? val : (R)0;
        }
    }

    // the second blurred column / row sits 0 or 1 pixel further (0 only on the last column / row): shift the window
    // once with selects instead of indexing registers dynamically
    R vc[6][5];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int q = 0; q < 5; ++q) vc[r][q] = dx ? v[r][q + 1] : v[r][q];
    const R taps[5] = {(R)1, (R)4, (R)6, (R)4, (R)1};
    R s00 = (R)0, s01 = (R)0, s10 = (R)0, s11 = (R)0;
#pragma unroll
    for (int p = 0; p < 5; ++p)
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const R k = (taps[p] * taps[q]) / (R)256;
            s00 = km_fma(k, v[p][q], s00);
            s01 = km_fma(k, vc[p][q], s01);
            s10 = km_fma(k, dy ? v[p + 1][q] : v[p][q], s10);
            s11 = km_fma(k, dy ? vc[p + 1][q] : vc[p][q], s11);
        }
    R blur[2][2];
    blur[0][0] = kmp_round(s00, (const T*)nullptr); blur[0][1] = kmp_round(s01, (const T*)nullptr);
    blur[1][0] = kmp_round(s10, (const T*)nullptr); blur[1][1] = kmp_round(s11, (const T*)nullptr);
    const R top = w0 * blur[0][0] + w1 * blur[0][1];
    const R bot = w0 * blur[1][0] + w1 * blur[1][1];
    km_st(a.y + ((size_t)bc * a.oh + oy) * a.ow + ox, h0 * top + h1 * bot);
}

template <typename T>
__global__ __launch_bounds__(256) void km_resize_bilinear_kernel(const KmPyrArgs<T> a) {
    typedef typename KmTraits<T>::R R;
    int ox, oy;
    uint32_t bc;
    if (!kmp_tile(a, ox, oy, bc)) return;
    const int H = a.H, W = a.W;
    const T* img = a.x + (size_t)bc * H * W;
    int y0, y1, x0, x1;
    R h0, h1, w0, w1;
    kmp_axis<R>(oy, H, a.oh, a.align, y0, y1, h0, h1);
    kmp_axis<R>(ox, W, a.ow, a.align, x0, x1, w0, w1);
    const T* r0 = img + (size_t)y0 * W;
    const T* r1 = img + (size_t)y1 * W;
    const R v00 = (R)km_ld(r0 + x0), v01 = (R)km_ld(r0 + x1), v10 = (R)km_ld(r1 + x0), v11 = (R)km_ld(r1 + x1);
    const R top = w0 * v00 + w1 * v01;
    const R bot = w0 * v10 + w1 * v11;
    km_st(a.y + ((size_t)bc * a.oh + oy) * a.ow + ox, h0 * top + h1 * bot);
}

// ---- factor-2 fast path -----------------------------------------------------------------------------------------------
#define KMP_ROWS 32



template <typename T>
__global__ __launch_bounds__(256) void km_pyrdown2_kernel(const KmPyrArgs<T> a) {
    constexpr int K = 5, PD = 2, NV = 4 + 2 * PD;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tbx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t tby = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx = (int)tbx * 64 + lane;              // column group (4 input px = 2 output px)
    const int r0 = ((int)tby * 4 + wave) * KMP_ROWS;  // first input row of this wave's strip (even)
    const int H = a.H, W = a.W, border = a.border;
    if (gx * 4 >= W || r0 >= H) return;
    const int c0 = gx * 4;
    const T* img = a.x + (size_t)bc * H * W;
    T* out = a.y + (size_t)bc * a.oh * a.ow;

    int hl[PD], hr[PD];
    bool okl[PD], okr[PD];
#pragma unroll
    for (int q = 0; q < PD; ++q) {
        const int il = km_border_map(c0 - PD + q, W, border), ir = km_border_map(c0 + 4 + q, W, border);
        okl[q] = il >= 0; hl[q] = okl[q] ? il : 0;
        okr[q] = ir >= 0; hr[q] = okr[q] ? ir : 0;
    }
    const float taps[5] = {1.f, 4.f, 6.f, 4.f, 1.f};
    float ring[K][NV];  // last K input rows: ring[.][i] = column c0 - PD + i
    float prev[4] = {0.f, 0.f, 0.f, 0.f};
    const int n_rows = (r0 + KMP_ROWS <= H ? KMP_ROWS : H - r0);  // even: H is even and r0 a multiple of 32
    const int total = n_rows + K - 1;
    for (int it0 = 0; it0 < total; it0 += K) {
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int it = it0 + kk;
            if (it < total) {
                const int srow = km_border_map(r0 - PD + it, H, border);  // wave-uniform
                if (srow >= 0) {
                    const T* rowp = img + (size_t)srow * W;
                    float o4[4];
                    km_ld4(rowp + c0, o4);
#pragma unroll
                    for (int q = 0; q < PD; ++q) {
                        const float vl = (float)km_ld(rowp + hl[q]), vr = (float)km_ld(rowp + hr[q]);
                        ring[kk][q] = okl[q] ? vl : 0.f;
                        ring[kk][PD + 4 + q] = okr[q] ? vr : 0.f;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) ring[kk][PD + q] = o4[q];
                } else {
#pragma unroll
                    for (int q = 0; q < NV; ++q) ring[kk][q] = 0.f;
                }
                if (it >= K - 1) {
                    const int r = r0 + it - (K - 1);  // blurred row, wave-uniform
                    float cur[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float s = 0.f;
#pragma unroll
                        for (int p = 0; p < K; ++p)
#pragma unroll
                            for (int q = 0; q < K; ++q) s = km_fma((taps[p] * taps[q]) / 256.f, ring[(kk + 1 + p) % K][c + q], s);
                        cur[c] = kmp_round(s, (const T*)nullptr);
                    }
                    if (r & 1) {
                        // all four bilinear weights are exactly 1/2 here: h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)
                        const float o0 = 0.5f * (0.5f * prev[0] + 0.5f * prev[1]) + 0.5f * (0.5f * cur[0] + 0.5f * cur[1]);
                        const float o1 = 0.5f * (0.5f * prev[2] + 0.5f * prev[3]) + 0.5f * (0.5f * cur[2] + 0.5f * cur[3]);
                        km_st2(out + (size_t)(r >> 1) * a.ow + (c0 >> 1), o0, o1);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) prev[c] = cur[c];
                    }
                }
            }
        }
    }
}

// ---- factor-2 fast path, separable evaluation (opt-in: KM_PYRDOWN_ALGO=separable) ---------------------------------------------
// The binomial kernel is the outer product of [1 4 6 4 1] / 16 with itself: a horizontal 5-tap pass on every loaded row (kept in
// the rolling window) and a vertical 5-tap pass per output row need 40 multiply-adds per lane and row instead of 100.  The sum
// is the same real number in another rounding order (<= 2 ulp from the 25-tap chain of filter2d that the default path and the
// oracle follow), so it stays opt-in until a device measurement says the 25-tap chain is what bounds the kernel.
template <typename T>
__global__ __launch_bounds__(256) void km_pyrdown2_sep_kernel(const KmPyrArgs<T> a) {
    constexpr int K = 5, PD = 2, NV = 4 + 2 * PD;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tbx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t tby = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx = (int)tbx * 64 + lane;
    const int r0 = ((int)tby * 4 + wave) * KMP_ROWS;
    const int H = a.H, W = a.W, border = a.border;
    if (gx * 4 >= W || r0 >= H) return;
    const int c0 = gx * 4;
    const T* img = a.x + (size_t)bc * H * W;
    T* out = a.y + (size_t)bc * a.oh * a.ow;

    int hl[PD], hr[PD];
    bool okl[PD], okr[PD];
#pragma unroll
    for (int q = 0; q < PD; ++q) {
        const int il = km_border_map(c0 - PD + q, W, border), ir = km_border_map(c0 + 4 + q, W, border);
        okl[q] = il >= 0; hl[q] = okl[q] ? il : 0;
        okr[q] = ir >= 0; hr[q] = okr[q] ? ir : 0;
    }
    const float taps[5] = {1.f / 16.f, 4.f / 16.f, 6.f / 16.f, 4.f / 16.f, 1.f / 16.f};
    float ring[K][4];  // last K input rows after the horizontal pass
    float prev[4] = {0.f, 0.f, 0.f, 0.f};
    const int n_rows = (r0 + KMP_ROWS <= H ? KMP_ROWS : H - r0);
    const int total = n_rows + K - 1;
    for (int it0 = 0; it0 < total; it0 += K) {
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int it = it0 + kk;
            if (it < total) {
                const int srow = km_border_map(r0 - PD + it, H, border);  // wave-uniform
                if (srow >= 0) {
                    const T* rowp = img + (size_t)srow * W;
                    float v[NV];
                    float o4[4];
                    km_ld4(rowp + c0, o4);
#pragma unroll
                    for (int q = 0; q < PD; ++q) {
                        const float vl = (float)km_ld(rowp + hl[q]), vr = (float)km_ld(rowp + hr[q]);
                        v[q] = okl[q] ? vl : 0.f;
                        v[PD + 4 + q] = okr[q] ? vr : 0.f;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[PD + q] = o4[q];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float s = 0.f;
#pragma unroll
                        for (int q = 0; q < K; ++q) s = km_fma(taps[q], v[c + q], s);
                        ring[kk][c] = s;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) ring[kk][c] = 0.f;
                }
                if (it >= K - 1) {
                    const int r = r0 + it - (K - 1);
                    float cur[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float s = 0.f;
#pragma unroll
                        for (int p = 0; p < K; ++p) s = km_fma(taps[p], ring[(kk + 1 + p) % K][c], s);
                        cur[c] = kmp_round(s, (const T*)nullptr);
                    }
                    if (r & 1) {
                        const float o0 = 0.5f * (0.5f * prev[0] + 0.5f * prev[1]) + 0.5f * (0.5f * cur[0] + 0.5f * cur[1]);
                        const float o1 = 0.5f * (0.5f * prev[2] + 0.5f * prev[3]) + 0.5f * (0.5f * cur[2] + 0.5f * cur[3]);
                        km_st2(out + (size_t)(r >> 1) * a.ow + (c0 >> 1), o0, o1);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) prev[c] = cur[c];
                    }
                }
            }
        }
    }
}

// ---- exact x2 upsampling (the resize leg of pyrup) ------------------------------------------------------------------------
// align_corners = False, (oh, ow) = (2H, 2W): output columns 4g .. 4g+3 read source columns 2g-1 .. 2g+2 and output rows
// 2s-1, 2s read source rows s-1, s.  A lane owns 2 source columns (4 output columns, one 16-byte store), a wave walks a strip
// of source rows keeping the horizontally interpolated previous row in registers: every source row is loaded once per
// lane (+2 halo rows per strip) instead of 4 gathered loads per output pixel.  Indices and weights still come from
// kmp_axis, so the borders (weights (1, 0) on the first row / column, repeated index on the last) need no special case;
// where kmp_axis gives weight 0 the neighbour loaded here may differ from ATen's, which is exact for finite data.
template <typename T>
__global__ __launch_bounds__(256) void km_resize2x_kernel(const KmPyrArgs<T> a) {
    typedef typename KmTraits<T>::R R;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tbx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t tby = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = (int)tbx * 64 + lane;               // source column pair
    const int s0 = ((int)tby * 4 + wave) * KMP_ROWS;  // first source row of this wave's strip
    const int H = a.H, W = a.W;
    if (g * 2 >= W || s0 >= H) return;
    const int c0 = g * 2;
    const T* img = a.x + (size_t)bc * H * W;
    T* out = a.y + (size_t)bc * a.oh * a.ow;

    // horizontal weights of the 4 output columns; their source pairs are (c0-1, c0), (c0, c0+1), (c0, c0+1), (c0+1, c0+2)
    R w0[4], w1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int x0, x1;
        kmp_axis<R>(c0 * 2 + t, W, a.ow, 0, x0, x1, w0[t], w1[t]);
    }
    const int cl = c0 > 0 ? c0 - 1 : 0, cr = c0 + 2 < W ? c0 + 2 : W - 1;
    const int s1 = s0 + KMP_ROWS < H ? s0 + KMP_ROWS : H;  // strip = source rows [s0, s1) = output rows [2 s0, 2 s1)
    R prev[4] = {(R)0, (R)0, (R)0, (R)0};
    for (int s = s0 - 1; s <= s1; ++s) {
        const int sr = s < 0 ? 0 : (s > H - 1 ? H - 1 : s);
        const T* rowp = img + (size_t)sr * W;
        R m0, m1;
        km_ld2(rowp + c0, m0, m1);
        const R vl = (R)km_ld(rowp + cl), vr = (R)km_ld(rowp + cr);
        R cur[4];
        cur[0] = w0[0] * vl + w1[0] * m0;
        cur[1] = w0[1] * m0 + w1[1] * m1;
        cur[2] = w0[2] * m0 + w1[2] * m1;
        cur[3] = w0[3] * m1 + w1[3] * vr;
        if (s >= s0) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int oy = 2 * s - 1 + half;  // rows 2s-1 and 2s are interpolated between source rows s-1 and s
                if (oy >= 2 * s0 && oy < 2 * s1) {
                    int y0, y1;
                    R h0, h1;
                    kmp_axis<R>(oy, H, a.oh, 0, y0, y1, h0, h1);
                    R o[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = h0 * prev[t] + h1 * cur[t];
                    km_st4(out + (size_t)oy * a.ow + c0 * 2, o);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) prev[t] = cur[t];
    }
}

template <typename T>
static int kmp_run_up2(const void* x, void* y, int B, int C, int H, int W, hipStream_t s) {
    KmPyrArgs<T> a;
    a.x = (const T*)x; a.y = (T*)y;
    a.H = H; a.W = W; a.oh = 2 * H; a.ow = 2 * W; a.border = 0; a.align = 0;
    a.tiles_x = (uint32_t)((W / 2 + 63) / 64);
    a.tiles_y = (uint32_t)((H + 4 * KMP_ROWS - 1) / (4 * KMP_ROWS));
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B * (uint64_t)C;
    KM_REQUIRE(nb < (1ull << 31), "km_resize_bilinear_fwd: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    hipLaunchKernelGGL((km_resize2x_kernel<T>), dim3(a.nblocks), dim3(256), 0, s, a);
    return km_check_launch("km_resize_bilinear_fwd(x2)");
}

// exact x2, align_corners = False, even W >= 2, H >= 2, 16-byte aligned output rows, not fp64
static bool kmp_up2_ok(const void* x, const void* y, int H, int W, int oh, int ow, int align, int dtype) {
    if (align || dtype == KM_F64 || (W & 1) || W < 2 || H < 2 || oh != 2 * H || ow != 2 * W) return false;
    const size_t esz = (dtype == KM_F32) ? 4 : 2;
    return ((uintptr_t)x % (2 * esz)) == 0 && ((uintptr_t)y % (4 * esz)) == 0;
}

template <typename T>
static int kmp_run2(const void* x, void* y, int B, int C, int H, int W, int border, hipStream_t s) {
    KmPyrArgs<T> a;
    a.x = (const T*)x; a.y = (T*)y;
    a.H = H; a.W = W; a.oh = H / 2; a.ow = W / 2; a.border = border; a.align = 0;
    a.tiles_x = (uint32_t)((W / 4 + 63) / 64);
    a.tiles_y = (uint32_t)((H + 4 * KMP_ROWS - 1) / (4 * KMP_ROWS));
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B * (uint64_t)C;
    KM_REQUIRE(nb < (1ull << 31), "km_pyrdown_fwd: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    if (km_config().pyrdown_separable) {  // KM_PYRDOWN_ALGO=separable: the 5 + 5 tap evaluation (A/B timing, see above)
        hipLaunchKernelGGL((km_pyrdown2_sep_kernel<T>), dim3(a.nblocks), dim3(256), 0, s, a);
        return km_check_launch("km_pyrdown_fwd(x2, separable)");
    }
    hipLaunchKernelGGL((km_pyrdown2_kernel<T>), dim3(a.nblocks), dim3(256), 0, s, a);
    return km_check_launch("km_pyrdown_fwd(x2)");
}

// factor 2, align_corners = False, even H, W % 4 == 0, 16-byte aligned rows, not fp64
static bool kmp_fast_ok(const void* x, const void* y, int H, int W, int oh, int ow, int align, int dtype) {
    if (align || dtype == KM_F64 || (H & 1) || (W & 3) || W < 8 || oh * 2 != H || ow * 2 != W) return false;
    const size_t esz = (dtype == KM_F32) ? 4 : 2;
    return ((uintptr_t)x % (4 * esz)) == 0 && ((uintptr_t)y % (2 * esz)) == 0;
}

template <typename T>
static int kmp_run(bool blur, const void* x, void* y, int B, int C, int H, int W, int oh, int ow, int border, int align, hipStream_t s) {
    KmPyrArgs<T> a;
    a.x = (const T*)x; a.y = (T*)y;
    a.H = H; a.W = W; a.oh = oh; a.ow = ow; a.border = border; a.align = align;
    a.tiles_x = (uint32_t)((ow + 63) / 64);
    a.tiles_y = (uint32_t)((oh + 3) / 4);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B * (uint64_t)C;
    KM_REQUIRE(nb < (1ull << 31), "km_pyramid: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    if (blur) {
        hipLaunchKernelGGL((km_pyrdown_kernel<T>), dim3(a.nblocks), dim3(256), 0, s, a);
        return km_check_launch("km_pyrdown_fwd");
    }
    hipLaunchKernelGGL((km_resize_bilinear_kernel<T>), dim3(a.nblocks), dim3(256), 0, s, a);
    return km_check_launch("km_resize_bilinear_fwd");
}

// ------------------------------------------------------------------------------------------------
// Adjoint of the bilinear resize (aten::upsample_bilinear2d_backward): grad_x[i][j] = sum over the output pixels (d, e) whose
// 2 x 2 footprint holds (i, j) of  wy(d -> i) * wx(e -> j) * grad_y[d][e].  Gather form - one lane per INPUT pixel, no atomics, no
// zero-fill, bit-reproducible: the source index of the forward (kmp_axis) is non-decreasing in the output index, so the outputs
// that touch input row i are one contiguous range, found from the inverse of the index map and then corrected with the
// forward's own arithmetic (so that rounding can neither drop nor double-count an output).
template <typename R>
__device__ __forceinline__ void kmp_adjoint_range(int i, int n_in, int n_out, int align, int& lo, int& hi) {
    // estimate of the first output whose i0 >= i - 1 / last whose i0 <= i
    const R scale = align ? (n_out > 1 ? (R)(n_in - 1) / (R)(n_out - 1) : (R)0) : (R)n_in / (R)n_out;
    int a, b;
    if (scale > (R)0) {
        const R inv = (R)1 / scale;
        a = (int)km_floor(align ? ((R)(i - 1)) * inv : ((R)(i - 1) + (R)0.5) * inv - (R)0.5) - 1;
        b = (int)km_floor(align ? ((R)(i + 1)) * inv : ((R)(i + 1) + (R)0.5) * inv - (R)0.5) + 1;
    } else {
        a = 0; b = n_out - 1;
    }
    lo = min(max(a, 0), n_out - 1);
    hi = min(max(b, 0), n_out - 1);
    int i0, i1; R l0, l1;
    // exact correction with the forward's index arithmetic
    while (lo > 0) { kmp_axis<R>(lo - 1, n_in, n_out, align, i0, i1, l0, l1); if (i1 >= i) --lo; else break; }
    while (lo < hi) { kmp_axis<R>(lo, n_in, n_out, align, i0, i1, l0, l1); if (i1 < i) ++lo; else break; }
    while (hi < n_out - 1) { kmp_axis<R>(hi + 1, n_in, n_out, align, i0, i1, l0, l1); if (i0 <= i) ++hi; else break; }
    while (hi > lo) { kmp_axis<R>(hi, n_in, n_out, align, i0, i1, l0, l1); if (i0 > i) --hi; else break; }
}
template <typename R>
__device__ __forceinline__ R kmp_adjoint_weight(int d, int i, int n_in, int n_out, int align) {
    int i0, i1; R l0, l1;
    kmp_axis<R>(d, n_in, n_out, align, i0, i1, l0, l1);
    return (i0 == i ? l0 : (R)0) + (i1 == i ? l1 : (R)0);  // (i1 == i0 on the last row: both weights land on it)
}

template <typename T>
__global__ __launch_bounds__(256) void km_resize_bilinear_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int H, int W, int oh, int ow, int align,
                                                                     uint32_t tiles_x) {
    typedef typename KmTraits<T>::R R;
    const uint32_t tbx = blockIdx.x % tiles_x, i = blockIdx.x / tiles_x;  // one input row per block-row
    const uint32_t bc = blockIdx.y;
    const int j = (int)tbx * 256 + (int)threadIdx.x;
    if (j >= W) return;
    int ylo, yhi, xlo, xhi;
    kmp_adjoint_range<R>((int)i, H, oh, align, ylo, yhi);
    kmp_adjoint_range<R>(j, W, ow, align, xlo, xhi);
    const T* g = gy + (size_t)bc * oh * ow;
    R acc = (R)0;
    for (int d = ylo; d <= yhi; ++d) {
        const R wy = kmp_adjoint_weight<R>(d, (int)i, H, oh, align);
        if (wy == (R)0) continue;
        R row = (R)0;
        for (int e = xlo; e <= xhi; ++e) {
            const R wx = kmp_adjoint_weight<R>(e, j, W, ow, align);
            row = row + wx * (R)km_ld(g + (size_t)d * ow + e);
        }
        acc = acc + wy * row;
    }
    km_st(gx + ((size_t)bc * H + i) * W + j, acc);
}

template <typename T>
static int kmp_run_resize_bwd(const void* gy, void* gx, int B, int C, int H, int W, int oh, int ow, int align, hipStream_t s) {
    const uint32_t tiles_x = (uint32_t)((W + 255) / 256);
    const uint64_t planes = (uint64_t)B * C;
    KM_REQUIRE((uint64_t)tiles_x * H < (1ull << 31) && planes < 65536ull * 65535ull, "km_resize_bilinear_bwd: grid too large");
    // grid.y carries the planes (<= 65535 per launch)
    for (uint64_t p0 = 0; p0 < planes; p0 += 65535) {
        const uint32_t np = (uint32_t)(planes - p0 < 65535 ? planes - p0 : 65535);
        hipLaunchKernelGGL((km_resize_bilinear_bwd_kernel<T>), dim3(tiles_x * (uint32_t)H, np), dim3(256), 0, s, (const T*)gy + p0 * (size_t)oh * ow,
                           (T*)gx + p0 * (size_t)H * W, H, W, oh, ow, align, tiles_x);
    }
    return km_check_launch("km_resize_bilinear_bwd");
}

static int kmp_validate(const char* what, const void* x, const void* y, int B, int C, int H, int W, int oh, int ow, int dtype) {
    KM_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && oh >= 0 && ow >= 0, "%s: bad shape B=%d C=%d H=%d W=%d -> %dx%d", what, B, C, H, W, oh, ow);
    KM_REQUIRE(dtype >= KM_F32 && dtype <= KM_F16, "%s: unknown dtype code %d", what, dtype);
    if ((uint64_t)B * C * oh * ow == 0) return 0;
    KM_REQUIRE(x && y, "%s: null pointer", what);
    return 0;
}

static int kmp_dispatch(bool blur, const void* x, void* y, int B, int C, int H, int W, int oh, int ow, int border, int align, int dtype, hipStream_t s) {
    switch (dtype) {
        case KM_F32: return kmp_run<float>(blur, x, y, B, C, H, W, oh, ow, border, align, s);
        case KM_F64: return kmp_run<double>(blur, x, y, B, C, H, W, oh, ow, border, align, s);
        case KM_BF16: return kmp_run<km_bf16>(blur, x, y, B, C, H, W, oh, ow, border, align, s);
        default: return kmp_run<km_f16>(blur, x, y, B, C, H, W, oh, ow, border, align, s);
    }
}

extern "C" {

int km_pyrdown_fwd(const void* x, void* y, int B, int C, int H, int W, int oh, int ow, int border, int align, int dtype, void* stream) {
    if (kmp_validate("km_pyrdown_fwd", x, y, B, C, H, W, oh, ow, dtype)) return -1;
    KM_REQUIRE(border >= KM_BORDER_CONSTANT && border <= KM_BORDER_CIRCULAR, "km_pyrdown_fwd: unknown border code %d", border);
    // torch's reflection padding needs pad < size (filter.py:139 F.pad): the 5x5 blur pads by 2
    KM_REQUIRE(border != KM_BORDER_REFLECT || (H > 2 && W > 2), "km_pyrdown_fwd: reflect padding needs H, W > 2 (got %dx%d)", H, W);
    if ((uint64_t)B * C * oh * ow == 0) return 0;
    if (kmp_fast_ok(x, y, H, W, oh, ow, align, dtype)) {
        switch (dtype) {
            case KM_F32: return kmp_run2<float>(x, y, B, C, H, W, border, (hipStream_t)stream);
            case KM_BF16: return kmp_run2<km_bf16>(x, y, B, C, H, W, border, (hipStream_t)stream);
            default: return kmp_run2<km_f16>(x, y, B, C, H, W, border, (hipStream_t)stream);
        }
    }
    return kmp_dispatch(true, x, y, B, C, H, W, oh, ow, border, align ? 1 : 0, dtype, (hipStream_t)stream);
}

int km_resize_bilinear_fwd(const void* x, void* y, int B, int C, int H, int W, int oh, int ow, int align, int dtype, void* stream) {
    if (kmp_validate("km_resize_bilinear_fwd", x, y, B, C, H, W, oh, ow, dtype)) return -1;
    if ((uint64_t)B * C * oh * ow == 0) return 0;
    if (kmp_up2_ok(x, y, H, W, oh, ow, align, dtype)) {
        switch (dtype) {
            case KM_F32: return kmp_run_up2<float>(x, y, B, C, H, W, (hipStream_t)stream);
            case KM_BF16: return kmp_run_up2<km_bf16>(x, y, B, C, H, W, (hipStream_t)stream);
            default: return kmp_run_up2<km_f16>(x, y, B, C, H, W, (hipStream_t)stream);
        }
    }
    return kmp_dispatch(false, x, y, B, C, H, W, oh, ow, 0, align ? 1 : 0, dtype, (hipStream_t)stream);
}

// Adjoint of km_resize_bilinear_fwd: gy (B,C,oh,ow) -> gx (B,C,H,W), written completely (no zero-fill needed); replaces
// aten::upsample_bilinear2d_backward behind F.interpolate(mode='bilinear') in kornia/geometry/transform/pyramid.py:447,501.
int km_resize_bilinear_bwd(const void* gy, void* gx, int B, int C, int H, int W, int oh, int ow, int align, int dtype, void* stream) {
    KM_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && oh > 0 && ow > 0, "km_resize_bilinear_bwd: bad shape B=%d C=%d H=%d W=%d <- %dx%d", B, C, H, W, oh, ow);
    KM_REQUIRE(dtype >= KM_F32 && dtype <= KM_F16, "km_resize_bilinear_bwd: unknown dtype code %d", dtype);
    if ((uint64_t)B * C == 0) return 0;
    KM_REQUIRE(gy && gx, "km_resize_bilinear_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case KM_F32: return kmp_run_resize_bwd<float>(gy, gx, B, C, H, W, oh, ow, align ? 1 : 0, s);
        case KM_F64: return kmp_run_resize_bwd<double>(gy, gx, B, C, H, W, oh, ow, align ? 1 : 0, s);
        case KM_BF16: return kmp_run_resize_bwd<km_bf16>(gy, gx, B, C, H, W, oh, ow, align ? 1 : 0, s);
        default: return kmp_run_resize_bwd<km_f16>(gy, gx, B, C, H, W, oh, ow, align ? 1 : 0, s);
    }
}

}  // extern "C"
