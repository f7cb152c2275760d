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
// Mapping: a wave owns a 64-wide x KML_ROWS-tall strip of the output (lane = column, coalesced dst reads), per-thread
// fp32 partial sums over <= KML_ROWS * C terms, fp64 wave / block reduction, 11 fp64 atomics per block.
#include "km_sampler.h"

#define KML_ROWS 8
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

template <typename T, int CM>
__global__ __launch_bounds__(256) void km_warp_loss_kernel(const KmWarpLossArgs<T> a) {
    typedef float R;
    const KmWarpGeom<R>& g = a.g;
    __shared__ double red[4][11];
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = (int)tx * 64 + lane;
    const int i_base = (int)ty * KML_TILE_H + wave * KML_ROWS;

    R m[9];
    {
        const R* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = mp[k];
    }
    const int W = g.W, H = g.H, align = g.align, C = g.C;
    const size_t src_plane = (size_t)H * W, dst_plane = (size_t)g.h * g.w;
    const T* __restrict__ src_b = a.src + (size_t)b * C * src_plane;
    const T* __restrict__ dst_b = a.dst + (size_t)b * C * dst_plane;
    const bool col_ok = j < g.w;
    const R u = km_base_x<R, CM>(g, col_ok ? j : 0);
    const bool mse = a.loss_kind == KML_MSE;

    R S[3] = {0, 0, 0}, Sv[3] = {0, 0, 0};
    R loss_sum = 0;
    int count = 0;
    for (int r = 0; r < KML_ROWS; ++r) {
        const int i = i_base + r;
        const bool ok = col_ok && i < g.h;  // wave-uniform in i, per-lane in j
        const R v = km_base_y<R, CM>(g, i < g.h ? i : 0);
        KmCoord<R> cd;
        km_gen_coord<R, CM>(m, u, v, cd);
        R mx, my;
        const R x = km_unnormalize(cd.gx, W, align, mx);
        const R y = km_unnormalize(cd.gy, H, align, my);
        KmBilin<R> t;
        km_bilinear_setup(x, y, W, H, t);
        // the warped ones image: the in-bounds weights through the forward's fma chain (nw, ne, sw, se from 0)
        R ones = 0;
        if (t.b00) ones = km_fma((R)1, t.w00, ones);
        if (t.b01) ones = km_fma((R)1, t.w01, ones);
        if (t.b10) ones = km_fma((R)1, t.w10, ones);
        if (t.b11) ones = km_fma((R)1, t.w11, ones);
        ones = km_round_as(ones, (const T*)nullptr);
        const bool sel = ok && (ones > a.threshold);
        const uint32_t d_off = ok ? (uint32_t)i * (uint32_t)g.w + (uint32_t)j : 0u;
        R gix = 0, giy = 0;
        if (__any(sel)) {
            const bool inside = __all(t.b00 && t.b01 && t.b10 && t.b11);
            for (int c = 0; c < C; ++c) {
                const T* img = src_b + (size_t)c * src_plane;
                R s00, s01, s10, s11;
                if (inside) {  // (x0, x0 + 1) of a row with one load
                    km_ld2(km_at(img, (uint32_t)t.i00), s00, s01);
                    km_ld2(km_at(img, (uint32_t)t.i10), s10, s11);
                } else {       // clamped addresses, out-of-bounds taps are not part of the reference's sum
                    const R v00 = (R)km_ld(img + t.i00), v01 = (R)km_ld(img + t.i01), v10 = (R)km_ld(img + t.i10), v11 = (R)km_ld(img + t.i11);
                    s00 = t.b00 ? v00 : (R)0; s01 = t.b01 ? v01 : (R)0; s10 = t.b10 ? v10 : (R)0; s11 = t.b11 ? v11 : (R)0;
                }
                R wv = 0;
                if (t.b00) wv = km_fma(s00, t.w00, wv);
                if (t.b01) wv = km_fma(s01, t.w01, wv);
                if (t.b10) wv = km_fma(s10, t.w10, wv);
                if (t.b11) wv = km_fma(s11, t.w11, wv);
                wv = km_round_as(wv, (const T*)nullptr);  // the warped image is stored in the image dtype by the reference
                const R d = (R)km_ld(km_at(dst_b + (size_t)c * dst_plane, d_off));
                const R diff = wv - d;
                const R e = mse ? diff * diff : km_fabs(diff);
                const R ge = mse ? (R)2 * diff : (diff > (R)0 ? (R)1 : (diff < (R)0 ? (R)-1 : (R)0));
                if (sel) {
                    loss_sum += e;
                    count += 1;
                    gix = km_fma(ge, km_fma(s01 - s00, t.wy1, (s11 - s10) * t.wy0), gix);
                    giy = km_fma(ge, km_fma(s10 - s00, t.wx1, (s11 - s01) * t.wx0), giy);
                }
            }
        }
        const R gx_ = sel ? gix * mx : (R)0, gy_ = sel ? giy * my : (R)0;
        R ax, ay, az;
        km_gm_terms<CM>(cd, gx_, gy_, ax, ay, az);
        if (sel) {  // unselected pixels may have undefined coordinates (NaN * 0): keep them out of the sums
            S[0] += ax; S[1] += ay; S[2] += az;
            Sv[0] = km_fma(ax, cd.v, Sv[0]); Sv[1] = km_fma(ay, cd.v, Sv[1]); Sv[2] = km_fma(az, cd.v, Sv[2]);
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
        const double s = km_wave_sum((double)out[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 11) {
        const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (s != 0.0) km_atomic_add(a.acc + (size_t)b * 11 + threadIdx.x, s);
    }
}

template <typename T>
static int kml_run(const void* src, const void* dst, const void* mat, double* acc, int B, int C, int H, int W, int h, int w, int B_M,
                   int coord_mode, int norm_coords, int align, int loss_kind, double threshold, hipStream_t s) {
    KmWarpLossArgs<T> a;
    a.src = (const T*)src; a.dst = (const T*)dst; a.mat = (const float*)mat; a.acc = acc;
    a.threshold = (float)threshold; a.loss_kind = loss_kind;
    KmWarpGeom<float>& g = a.g;
    km_geom_init(g, B, C, H, W, h, w, B_M, coord_mode, norm_coords, KM_INTERP_BILINEAR, KM_PAD_ZEROS, align);
    a.tiles_x = (uint32_t)((w + 63) / 64);
    a.tiles_y = (uint32_t)((h + KML_TILE_H - 1) / KML_TILE_H);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B;
    KM_REQUIRE(nb < (1ull << 31), "km_warp_masked_loss: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    switch (coord_mode) {
        case KM_COORD_PERSPECTIVE: hipLaunchKernelGGL((km_warp_loss_kernel<T, KM_COORD_PERSPECTIVE>), dim3(a.nblocks), dim3(256), 0, s, a); break;
        case KM_COORD_AFFINE: hipLaunchKernelGGL((km_warp_loss_kernel<T, KM_COORD_AFFINE>), dim3(a.nblocks), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((km_warp_loss_kernel<T, KM_COORD_HOMOGRAPHY>), dim3(a.nblocks), dim3(256), 0, s, a); break;
    }
    return km_check_launch("km_warp_masked_loss");
}

extern "C" {

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
