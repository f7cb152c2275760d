// kornia_amd - masked photometric loss of a warp and its gradient with respect to the matrix, in ONE launch (gfx950).
//
// Replaces ImageRegistrator.get_single_level_loss (kornia/geometry/transform/image_registrator.py:225-245) together
// with its autograd backward wrt the model:
//     warped = HomographyWarper(h, w)(src, H)                      # homography_warp, bilinear, zeros padding
//     ones   = HomographyWarper(h, w)(ones_like(src), H)           # = sum of the in-bounds bilinear weights
//     loss   = loss_fn(warped, dst, reduction='none').masked_select(ones > 0.9).mean()
// The reference writes the warped image, the warped ones image, the elementwise loss, the mask and the compacted
// selection (whose size needs a device sync), and walks all of it again in backward: > 20e bytes per element.
// Here every output pixel is visited once: sampling position with the forward's exact instruction sequence
// (km_gen_coord), the four taps of each channel, the warped value (same fma chain as the forward), the mask from the
// in-bounds weights, the elementwise loss and its derivative, and the pixel's contribution to d loss / d matrix
// (km_gm_terms, as km_warp_gm.hip).  Nothing image-sized is written: HBM traffic = read src taps + read dst = 2e bytes
// per element for loss AND gradient.  Outputs are 11 fp64 accumulators per image (so that the atomics of one image's
// workgroups - which km_xcd_remap keeps on one XCD - are the only ones that meet on an address):
//     acc[b][0] = sum of the selected elementwise losses of image b, acc[b][1] = number of selected elements,
//     acc[b][2 + k] = d acc[b][0] / d mat[b][k]
// so loss = sum_b acc[b][0] / sum_b acc[b][1] and d loss / d mat[b] = acc[b][2..] / sum_b acc[b][1] (summed over b for a
// shared matrix); the mask is piecewise constant, as in autograd.
//
// Mapping (as km_warp_gm.hip, whose access pattern this is): a wave instruction covers a 32 x 2 patch of the output, a thread
// walks KML_ROWS rows two at a time with all loads of the pair in flight before the first use when every lane samples inside
// the image (RGB / grey unrolled); per-thread fp32 partial sums over <= KML_ROWS * C terms, fp64 wave / block reduction,
// 11 fp64 atomics per block.
#include "km_sampler.h"

// output rows per thread; a wave instruction covers a KML_PATCH_W x (64 / KML_PATCH_W) patch of the output, which keeps the
// gathered taps compact under rotation (measured for the same access pattern in km_warp_gm.hip)
#define KML_ROWS 16
#define KML_GROUP 2  // rows whose loads are in flight together
#define KML_PATCH_W 32
#define KML_TILE_W 64
#define KML_TILE_H (4 * KML_ROWS)

enum { KML_L1 = 0, KML_MSE = 1 };

template <typename T>
struct KmWarpLossArgs {
    const T* src;      // (B,C,H,W)
    const T* dst;      // (B,C,h,w)
    const float* mat;  // (B_M,9)
    double* acc;       // (B, 11) fp64, pre-zeroed
    KmWarpGeom<float> g;
    float threshold;
    int loss_kind;
    uint32_t tiles_x, tiles_y, nblocks;
};

// elementwise loss and its derivative wrt the warped value
__device__ __forceinline__ void kml_elem(bool mse, float diff, float& e, float& ge) {
    e = mse ? diff * diff : km_fabs(diff);
    ge = mse ? 2.0f * diff : (diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f));
}

template <typename T, int CM, int NC>  // NC = 3 / 1: RGB / grey unrolled with batched loads ; NC = 0: runtime channel loop
__global__ __launch_bounds__(256) void km_warp_loss_kernel(const KmWarpLossArgs<T> a) {
    typedef float R;
    const KmWarpGeom<R>& g = a.g;
    __shared__ double red[4][11];
    __shared__ R s_v[KML_TILE_H];
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    constexpr int PW = KML_PATCH_W, PH = 64 / PW, WA = 64 / PW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = (int)tx * KML_TILE_W + (wave % WA) * PW + (lane % PW);
    const int li_base = (wave / WA) * (PH * KML_ROWS) + lane / PW;  // row r of this thread sits at tile row li_base + r * PH
    const int i_base = (int)ty * KML_TILE_H + li_base;
    if (threadIdx.x < KML_TILE_H) s_v[threadIdx.x] = km_base_y<R, CM>(g, (int)ty * KML_TILE_H + (int)threadIdx.x);
    __syncthreads();

    R m[9];
    {
        const R* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = mp[k];
    }
    const int W = g.W, H = g.H, align = g.align;
    const int C = (NC > 0) ? NC : g.C;
    const size_t src_plane = (size_t)H * W, dst_plane = (size_t)g.h * g.w;
    const T* __restrict__ src_b = a.src + (size_t)b * C * src_plane;
    const T* __restrict__ dst_b = a.dst + (size_t)b * C * dst_plane;
    const bool col_ok = j < g.w;
    const R u = km_base_x<R, CM>(g, col_ok ? j : 0);
    const bool mse = a.loss_kind == KML_MSE;

    R S[3] = {0, 0, 0}, Sv[3] = {0, 0, 0};
    R loss_sum = 0;
    int count = 0;
    for (int r0 = 0; r0 < KML_ROWS; r0 += KML_GROUP) {
        KmCoord<R> cd[KML_GROUP];
        KmBilin<R> t[KML_GROUP];
        R mx[KML_GROUP], my[KML_GROUP], gix[KML_GROUP], giy[KML_GROUP];
        uint32_t d_off[KML_GROUP];
        bool sel[KML_GROUP], xbad[KML_GROUP], ybad[KML_GROUP];
        bool inside = true, any_sel = false;
#pragma unroll
        for (int q = 0; q < KML_GROUP; ++q) {
            const int i = i_base + (r0 + q) * PH;
            const bool ok = col_ok && (i < g.h);
            km_gen_coord<R, CM>(m, u, s_v[li_base + (r0 + q) * PH], cd[q]);
            const R x = km_unnormalize(cd[q].gx, W, align, mx[q]);
            const R y = km_unnormalize(cd[q].gy, H, align, my[q]);
            km_bilinear_setup(x, y, W, H, t[q]);
            // the warped ones image: the in-bounds weights through the forward's fma chain (nw, ne, sw, se from 0)
            R ones = km_bilinear_ones(t[q]);  // (taps outside the image: zeros that are still multiplied - NaN for a NaN / inf position)
            ones = km_round_as(ones, (const T*)nullptr);
            // threshold < 0 is "no mask" (the warped ones image is >= 0 wherever it is a number): every pixel counts, a NaN one included
            sel[q] = ok && ((a.threshold < 0.f) || (ones > a.threshold));
            // a pixel the mask drops still sends its (zero) gradient through the warp's backward in the reference: 0 times the NaN weights of
            // a non-finite position is NaN in the grid - hence the matrix - gradient (gix through the y weights, giy through the x weights)
            xbad[q] = ok && !km_finite(x);
            ybad[q] = ok && !km_finite(y);
            d_off[q] = ok ? (uint32_t)i * (uint32_t)g.w + (uint32_t)j : 0u;
            inside = inside && t[q].b00 && t[q].b01 && t[q].b10 && t[q].b11;
            any_sel = any_sel || sel[q];
            gix[q] = 0;
            giy[q] = 0;
        }
        if (__any(any_sel)) {
            if (NC > 0 && __all(inside)) {
                // every lane samples inside the image for all rows of the group: all loads back to back before the first use
                constexpr int NCC = NC > 0 ? NC : 1;
                R dv[KML_GROUP][NCC], v[KML_GROUP][NCC][4];
#pragma unroll
                for (int q = 0; q < KML_GROUP; ++q)
#pragma unroll
                    for (int c = 0; c < NCC; ++c) {
                        dv[q][c] = (R)km_ld(km_at(dst_b + c * dst_plane, d_off[q]));
                        km_ld2(km_at(src_b + c * src_plane, (uint32_t)t[q].i00), v[q][c][0], v[q][c][1]);
                        km_ld2(km_at(src_b + c * src_plane, (uint32_t)t[q].i10), v[q][c][2], v[q][c][3]);
                    }
#pragma unroll
                for (int q = 0; q < KML_GROUP; ++q)
#pragma unroll
                    for (int c = 0; c < NCC; ++c) {
                        const R s00 = v[q][c][0], s01 = v[q][c][1], s10 = v[q][c][2], s11 = v[q][c][3];
                        R wv = km_fma(s00, t[q].w00, (R)0);
                        wv = km_fma(s01, t[q].w01, wv);
                        wv = km_fma(s10, t[q].w10, wv);
                        wv = km_fma(s11, t[q].w11, wv);
                        wv = km_round_as(wv, (const T*)nullptr);  // the reference stores the warped image in the image dtype
                        R e, ge;
                        kml_elem(mse, wv - dv[q][c], e, ge);
                        if (sel[q]) {
                            loss_sum += e;
                            count += 1;
                            gix[q] = km_fma(ge, km_fma(s01 - s00, t[q].wy1, (s11 - s10) * t[q].wy0), gix[q]);
                            giy[q] = km_fma(ge, km_fma(s10 - s00, t[q].wx1, (s11 - s01) * t[q].wx0), giy[q]);
                        }
                    }
            } else {
#pragma unroll
                for (int q = 0; q < KML_GROUP; ++q) {
                    const KmBilin<R>& tq = t[q];
                    const bool row_inside = __all(tq.b00 && tq.b01 && tq.b10 && tq.b11);
                    for (int c = 0; c < C; ++c) {
                        const T* img = src_b + (size_t)c * src_plane;
                        R s00, s01, s10, s11;
                        if (row_inside) {  // (x0, x0 + 1) of a row with one load
                            km_ld2(km_at(img, (uint32_t)tq.i00), s00, s01);
                            km_ld2(km_at(img, (uint32_t)tq.i10), s10, s11);
                        } else {           // clamped addresses; out-of-bounds taps are not part of the reference's sum
                            const R v00 = (R)km_ld(img + tq.i00), v01 = (R)km_ld(img + tq.i01), v10 = (R)km_ld(img + tq.i10), v11 = (R)km_ld(img + tq.i11);
                            s00 = tq.b00 ? v00 : (R)0; s01 = tq.b01 ? v01 : (R)0; s10 = tq.b10 ? v10 : (R)0; s11 = tq.b11 ? v11 : (R)0;
                        }
                        R wv = km_fma(s00, tq.w00, (R)0);  // (s = 0 for a tap outside the image: still multiplied)
                        wv = km_fma(s01, tq.w01, wv);
                        wv = km_fma(s10, tq.w10, wv);
                        wv = km_fma(s11, tq.w11, wv);
                        wv = km_round_as(wv, (const T*)nullptr);
                        const R d = (R)km_ld(km_at(dst_b + (size_t)c * dst_plane, d_off[q]));
                        R e, ge;
                        kml_elem(mse, wv - d, e, ge);
                        if (sel[q]) {
                            loss_sum += e;
                            count += 1;
                            gix[q] = km_fma(ge, km_fma(s01 - s00, tq.wy1, (s11 - s10) * tq.wy0), gix[q]);
                            giy[q] = km_fma(ge, km_fma(s10 - s00, tq.wx1, (s11 - s01) * tq.wx0), giy[q]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < KML_GROUP; ++q) {
            const R qnan = (R)__int_as_float(0x7fc00000);
            const R gx_ = sel[q] ? gix[q] * mx[q] : (ybad[q] ? qnan : (R)0), gy_ = sel[q] ? giy[q] * my[q] : (xbad[q] ? qnan : (R)0);
            R ax, ay, az;
            km_gm_terms<CM>(cd[q], gx_, gy_, ax, ay, az);
            if (sel[q] || xbad[q] || ybad[q]) {  // (padding lanes / rows beyond the output may have undefined coordinates: kept out of the sums)
                S[0] += ax; S[1] += ay; S[2] += az;
                Sv[0] = km_fma(ax, cd[q].v, Sv[0]); Sv[1] = km_fma(ay, cd[q].v, Sv[1]); Sv[2] = km_fma(az, cd[q].v, Sv[2]);
            }
        }
    }
    R out[11];
    out[0] = loss_sum;
    out[1] = (R)count;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        out[2 + 3 * k + 0] = S[k] * u;
        out[2 + 3 * k + 1] = Sv[k];
        out[2 + 3 * k + 2] = S[k];
    }
    if (CM == KM_COORD_AFFINE) out[8] = out[9] = out[10] = 0;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const double s = km_wave_sum_last((double)out[k]);  // (DPP ladder: valid in lane 63; every thread of the block is here)
        if (lane == 63) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 11) {
        const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (s != 0.0) km_atomic_add(a.acc + (size_t)b * 11 + threadIdx.x, s);
    }
}

template <typename T, int CM>
static void kml_launch(const KmWarpLossArgs<T>& a, hipStream_t s) {
    if (a.g.C == 3)
        hipLaunchKernelGGL((km_warp_loss_kernel<T, CM, 3>), dim3(a.nblocks), dim3(256), 0, s, a);
    else if (a.g.C == 1)
        hipLaunchKernelGGL((km_warp_loss_kernel<T, CM, 1>), dim3(a.nblocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((km_warp_loss_kernel<T, CM, 0>), dim3(a.nblocks), dim3(256), 0, s, a);
}

template <typename T>
static int kml_run(const void* src, const void* dst, const void* mat, double* acc, int B, int C, int H, int W, int h, int w, int B_M,
                   int coord_mode, int norm_coords, int align, int loss_kind, double threshold, hipStream_t s) {
    KmWarpLossArgs<T> a;
    a.src = (const T*)src; a.dst = (const T*)dst; a.mat = (const float*)mat; a.acc = acc;
    a.threshold = (float)threshold; a.loss_kind = loss_kind;
    KmWarpGeom<float>& g = a.g;
    km_geom_init(g, B, C, H, W, h, w, B_M, coord_mode, norm_coords, KM_INTERP_BILINEAR, KM_PAD_ZEROS, align);
    a.tiles_x = (uint32_t)((w + KML_TILE_W - 1) / KML_TILE_W);
    a.tiles_y = (uint32_t)((h + KML_TILE_H - 1) / KML_TILE_H);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B;
    KM_REQUIRE(nb < (1ull << 31), "km_warp_masked_loss: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    switch (coord_mode) {
        case KM_COORD_PERSPECTIVE: kml_launch<T, KM_COORD_PERSPECTIVE>(a, s); break;
        case KM_COORD_AFFINE: kml_launch<T, KM_COORD_AFFINE>(a, s); break;
        default: kml_launch<T, KM_COORD_HOMOGRAPHY>(a, s); break;
    }
    return km_check_launch("km_warp_masked_loss");
}

// ---- what follows the accumulators: the two sums over the batch and the divisions, ONE launch ------------------------------------------
// (as torch ops: sum, two index / divide pairs and casts - nine launches of 4 - 10 us around an 86 us kernel at config 5's shape)
// One workgroup; thread t sums entries t, t + 256, ... of a column in index order, the 256 partials meet in a fixed tree: deterministic.
__global__ __launch_bounds__(256) void km_warp_loss_finish_kernel(const double* __restrict__ acc, int B, int B_M, double* __restrict__ loss,
                                                                   float* __restrict__ loss_f32, double* __restrict__ gm_unit) {
    __shared__ double red[256];
    __shared__ double tot[11];
    const int tid = threadIdx.x;
    const int ncol = (B_M == 1) ? 11 : 2;  // a shared matrix: its nine gradient entries are sums over the batch as well
    for (int k = 0; k < ncol; ++k) {
        double s = 0.0;
        for (int b = tid; b < B; b += 256) s += acc[(size_t)b * 11 + k];
        red[tid] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) red[tid] += red[tid + off];
            __syncthreads();
        }
        if (tid == 0) tot[k] = red[0];
        __syncthreads();
    }
    const double n = tot[1];
    if (tid == 0) {
        const double l = tot[0] / n;  // 0 / 0 = NaN when nothing is selected, like the mean of an empty selection
        loss[0] = l;
        if (loss_f32) loss_f32[0] = (float)l;
    }
    if (B_M == 1) {
        if (tid < 9) gm_unit[tid] = tot[2 + tid] / n;
    } else {
        for (int e = tid; e < B * 9; e += 256) gm_unit[e] = acc[(size_t)(e / 9) * 11 + 2 + (e % 9)] / n;
    }
}

// out[k] = (out type)(in[k] * (double)scale[0]): the product with the upstream gradient of the scalar loss, formed in fp64 and rounded once
template <typename TS, typename TO>
__global__ __launch_bounds__(256) void km_scale_f64_kernel(const double* __restrict__ in, const TS* __restrict__ scale, TO* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (TO)(in[i] * (double)scale[0]);
}

extern "C" {

int km_warp_masked_loss_finish(const double* acc, int B, int B_M, double* loss, float* loss_f32, double* gm_unit, void* stream) {
    KM_REQUIRE(B >= 1 && (B_M == 1 || B_M == B), "km_warp_masked_loss_finish: B >= 1 and B_M in {1, B} (got B=%d, B_M=%d)", B, B_M);
    KM_REQUIRE(acc && loss && gm_unit, "km_warp_masked_loss_finish: null pointer");
    hipLaunchKernelGGL(km_warp_loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, acc, B, B_M, loss, loss_f32, gm_unit);
    return km_check_launch("km_warp_masked_loss_finish");
}

int km_scale_f64(const double* in, const void* scale, int scale_dtype, void* out, int out_dtype, long long n, void* stream) {
    KM_REQUIRE(n >= 0, "km_scale_f64: n must be >= 0");
    KM_REQUIRE((scale_dtype == KM_F32 || scale_dtype == KM_F64) && (out_dtype == KM_F32 || out_dtype == KM_F64), "km_scale_f64: dtypes must be f32 (0) or f64 (1)");
    if (n == 0) return 0;
    KM_REQUIRE(in && scale && out, "km_scale_f64: null pointer");
    KM_REQUIRE((n + 255) / 256 < (1ll << 31), "km_scale_f64: grid too large");
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (scale_dtype == KM_F32 && out_dtype == KM_F32) hipLaunchKernelGGL((km_scale_f64_kernel<float, float>), grid, block, 0, s, in, (const float*)scale, (float*)out, n);
    else if (scale_dtype == KM_F32) hipLaunchKernelGGL((km_scale_f64_kernel<float, double>), grid, block, 0, s, in, (const float*)scale, (double*)out, n);
    else if (out_dtype == KM_F32) hipLaunchKernelGGL((km_scale_f64_kernel<double, float>), grid, block, 0, s, in, (const double*)scale, (float*)out, n);
    else hipLaunchKernelGGL((km_scale_f64_kernel<double, double>), grid, block, 0, s, in, (const double*)scale, (double*)out, n);
    return km_check_launch("km_scale_f64");
}

int km_warp_masked_loss(const void* src, const void* dst, const void* mat, double* acc, int B, int C, int H, int W, int h, int w, int B_M,
                        int coord_mode, int norm_coords, int align, int loss_kind, double threshold, int dtype, void* stream) {
    KM_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && h >= 0 && w >= 0, "km_warp_masked_loss: bad shape");
    KM_REQUIRE(B_M == 1 || B_M == B, "km_warp_masked_loss: B_M must be 1 or B (got %d for B=%d)", B_M, B);
    KM_REQUIRE(coord_mode >= KM_COORD_PERSPECTIVE && coord_mode <= KM_COORD_HOMOGRAPHY, "km_warp_masked_loss: unknown coord_mode %d", coord_mode);
    KM_REQUIRE(loss_kind == KML_L1 || loss_kind == KML_MSE, "km_warp_masked_loss: loss_kind must be 0 (l1) or 1 (mse)");
    KM_REQUIRE(dtype == KM_F32 || dtype == KM_BF16 || dtype == KM_F16, "km_warp_masked_loss: dtype must be f32 / bf16 / f16");
    KM_REQUIRE(W >= 2, "km_warp_masked_loss: the source needs at least two columns");
    KM_REQUIRE((uint64_t)H * W * 4 < (1ull << 32) && (uint64_t)h * w * 4 < (1ull << 32), "km_warp_masked_loss: plane too large for 32-bit offsets");
    if ((uint64_t)B * C * h * w == 0) return 0;
    KM_REQUIRE(src && dst && mat && acc, "km_warp_masked_loss: null pointer");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case KM_F32: return kml_run<float>(src, dst, mat, acc, B, C, H, W, h, w, B_M, coord_mode, norm_coords, align ? 1 : 0, loss_kind, threshold, s);
        case KM_BF16: return kml_run<km_bf16>(src, dst, mat, acc, B, C, H, W, h, w, B_M, coord_mode, norm_coords, align ? 1 : 0, loss_kind, threshold, s);
        default: return kml_run<km_f16>(src, dst, mat, acc, B, C, H, W, h, w, B_M, coord_mode, norm_coords, align ? 1 : 0, loss_kind, threshold, s);
    }
}

}  // extern "C"
