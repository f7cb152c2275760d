// kornia_amd - gradient of the bilinear warps with respect to the (B,3,3) matrix, for gfx950.
//
// d loss / d M  =  sum over output pixels q of  J_q^T (dL/dx_q, dL/dy_q),  with
//   dL/dx_q = sum_c grad_out[q,c] * ((ne - nw)(y1 - y) + (se - sw)(y - y0))     (image gradient at the sample)
//   dL/dy_q = sum_c grad_out[q,c] * ((sw - nw)(x1 - x) + (se - ne)(x - x0))
// and J_q the Jacobian of the sampling position with respect to the matrix entries (SURVEY.md A.6;
// reference: autograd through kornia/geometry/transform/imgwarp.py:157-174 / linalg.py:219-239).
//
// This is an OUTPUT-pixel-centric reduction: it has the access pattern of the forward warp (coalesced
// grad_out reads, gathered src taps) and none of the scatter of grad_src, so it runs as its own kernel
// shaped like the specialised forward (km_warp_fwd_bz_kernel) instead of riding in the tile-owner
// scatter kernel, where it cost more than the scatter itself (owner predication, 118 VGPRs).
//   * a wave owns a 64-wide x KMG_ROWS-tall strip of the output, lane = column;
//   * per-thread partial sums in fp32 (<= KMG_ROWS * C terms), wave + block reduction and the global
//     accumulation in fp64 (9 atomics per block);
//   * HBM traffic: read grad_out once + src once = 2e bytes / element.
#include <stdlib.h>

#include "km_sampler.h"

#ifndef KMG_ROWS
#define KMG_ROWS 16
#endif
#define KMG_TILE_W 64
#ifndef KMG_GROUP
#define KMG_GROUP 2   // rows whose loads are in flight together
#endif
#ifndef KMG_PATCH_W
#define KMG_PATCH_W 32  // measured at 256x3x512^2: 64 -> 0.375 ms (0.83 at 20 deg, 1.32 at 45 deg), 32 -> 0.368 (0.60, 0.84), 16 -> 0.412 (0.57, 0.68)
#endif
#define KMG_TILE_H (4 * KMG_ROWS)

template <typename T>
struct KmWarpGmArgs {
    const T* src;       // (B,C,H,W)
    const T* gout;      // (B,C,h,w)
    const float* mat;   // (B_M,9)
    double* gmat;       // (B_M,9) fp64 accumulators, pre-zeroed
    const float* fill;  // (C), pad == fill only
    KmWarpGeom<float> g;
    uint32_t tiles_x, tiles_y, nblocks;
};

template <typename T, int CM, int NC>  // NC = 3 / 1: RGB / grey unrolled ; NC = 0: runtime channel loop
__global__ __launch_bounds__(256) void km_warp_gm_kernel(const KmWarpGmArgs<T> a) {
    typedef float R;
    const KmWarpGeom<R>& g = a.g;
    __shared__ double red[4][9];
    __shared__ R s_v[KMG_TILE_H];
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    // a wave instruction covers a KMG_PATCH_W x (64 / KMG_PATCH_W) patch of the output (see km_warp_fwd_bz_kernel)
    constexpr int PW = KMG_PATCH_W, PH = 64 / PW, WA = 64 / PW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = (int)tx * KMG_TILE_W + (wave % WA) * PW + (lane % PW);
    const int li_base = (wave / WA) * (PH * KMG_ROWS) + lane / PW;  // row r of this thread sits at tile row li_base + r * PH
    const int i_base = (int)ty * KMG_TILE_H + li_base;
    if (threadIdx.x < KMG_TILE_H) s_v[threadIdx.x] = km_base_y<R, CM>(g, (int)ty * KMG_TILE_H + (int)threadIdx.x);
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
    const T* __restrict__ gout_b = a.gout + (size_t)b * C * dst_plane;
    const bool col_ok = j < g.w;
    const R u = km_base_x<R, CM>(g, col_ok ? j : 0);
    const bool is_fill = (g.pad == KM_PAD_FILL);

    // u is fixed per thread (lane = column): accumulate S = sum(ax, ay, az) and Sv = sum(v * (ax, ay, az)) over the
    // thread's rows and multiply by u once at the end
    R S[3] = {0, 0, 0}, Sv[3] = {0, 0, 0};

    // rows are taken KMG_GROUP at a time: all their sampling positions first, then - when every lane samples inside the
    // image for all of them - all their loads back to back before the first use (memory-latency bound otherwise)
    for (int r0 = 0; r0 < KMG_ROWS; r0 += KMG_GROUP) {
        KmCoord<R> cd[KMG_GROUP];
        KmBilin<R> t[KMG_GROUP];
        R mx[KMG_GROUP], my[KMG_GROUP], gix[KMG_GROUP], giy[KMG_GROUP];
        uint32_t go_off[KMG_GROUP];
        bool ok[KMG_GROUP];
        bool inside = true;
#pragma unroll
        for (int q = 0; q < KMG_GROUP; ++q) {
            const int i = i_base + (r0 + q) * PH;
            ok[q] = col_ok && (i < g.h);
            km_gen_coord<R, CM>(m, u, s_v[li_base + (r0 + q) * PH], cd[q]);
            const R x = km_unnormalize(cd[q].gx, W, align, mx[q]);
            const R y = km_unnormalize(cd[q].gy, H, align, my[q]);
            km_bilinear_setup(x, y, W, H, t[q]);
            go_off[q] = ok[q] ? (uint32_t)i * (uint32_t)g.w + (uint32_t)j : 0u;
            inside = inside && t[q].b00 && t[q].b01 && t[q].b10 && t[q].b11;
            gix[q] = 0;
            giy[q] = 0;
        }
        if (NC > 0 && __all(inside)) {
            constexpr int NCC = NC > 0 ? NC : 1;
            R go[KMG_GROUP][NCC], v[KMG_GROUP][NCC][4];
#pragma unroll
            for (int q = 0; q < KMG_GROUP; ++q)
#pragma unroll
                for (int c = 0; c < NCC; ++c) {
                    go[q][c] = (R)km_ld(km_at(gout_b + c * dst_plane, go_off[q]));
                    km_ld2(km_at(src_b + c * src_plane, (uint32_t)t[q].i00), v[q][c][0], v[q][c][1]);
                    km_ld2(km_at(src_b + c * src_plane, (uint32_t)t[q].i10), v[q][c][2], v[q][c][3]);
                }
#pragma unroll
            for (int q = 0; q < KMG_GROUP; ++q)
#pragma unroll
                for (int c = 0; c < NCC; ++c) {
                    R s00 = v[q][c][0], s01 = v[q][c][1], s10 = v[q][c][2], s11 = v[q][c][3];
                    if (is_fill) {  // same rounding sequence as the oracle: (v - fill) first
                        const R f = a.fill[c];
                        s00 -= f; s01 -= f; s10 -= f; s11 -= f;
                    }
                    gix[q] = km_fma(go[q][c], km_fma(s01 - s00, t[q].wy1, (s11 - s10) * t[q].wy0), gix[q]);
                    giy[q] = km_fma(go[q][c], km_fma(s10 - s00, t[q].wx1, (s11 - s01) * t[q].wx0), giy[q]);
                }
        } else {
#pragma unroll
            for (int q = 0; q < KMG_GROUP; ++q) {
                const KmBilin<R>& tq = t[q];
                if (__all(tq.b00 && tq.b01 && tq.b10 && tq.b11)) {
                    // the whole wave samples inside the image for this row: (x0, x0 + 1) come with one load per row
                    for (int c = 0; c < C; ++c) {
                        const R gv = (R)km_ld(km_at(gout_b + (size_t)c * dst_plane, go_off[q]));
                        const T* img = src_b + (size_t)c * src_plane;
                        R s00, s01, s10, s11;
                        km_ld2(km_at(img, (uint32_t)tq.i00), s00, s01);
                        km_ld2(km_at(img, (uint32_t)tq.i10), s10, s11);
                        if (is_fill) {
                            const R f = a.fill[c];
                            s00 -= f; s01 -= f; s10 -= f; s11 -= f;
                        }
                        gix[q] = km_fma(gv, km_fma(s01 - s00, tq.wy1, (s11 - s10) * tq.wy0), gix[q]);
                        giy[q] = km_fma(gv, km_fma(s10 - s00, tq.wx1, (s11 - s01) * tq.wx0), giy[q]);
                    }
                } else {
                    for (int c = 0; c < C; ++c) {
                        const R gv = (R)km_ld(km_at(gout_b + (size_t)c * dst_plane, go_off[q]));
                        const T* img = src_b + (size_t)c * src_plane;
                        const R f = is_fill ? a.fill[c] : (R)0;
                        // unconditional loads (clamped indices); out-of-bounds taps do not exist in the reference's sum
                        const R v00 = (R)km_ld(img + tq.i00), v01 = (R)km_ld(img + tq.i01), v10 = (R)km_ld(img + tq.i10), v11 = (R)km_ld(img + tq.i11);
                        const R s00 = tq.b00 ? v00 - f : (R)0, s01 = tq.b01 ? v01 - f : (R)0;
                        const R s10 = tq.b10 ? v10 - f : (R)0, s11 = tq.b11 ? v11 - f : (R)0;
                        gix[q] = km_fma(gv, km_fma(s01 - s00, tq.wy1, (s11 - s10) * tq.wy0), gix[q]);
                        giy[q] = km_fma(gv, km_fma(s10 - s00, tq.wx1, (s11 - s01) * tq.wx0), giy[q]);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < KMG_GROUP; ++q) {
            // pixels outside the output (padding lanes / rows of the last tiles) contribute nothing
            const R gx_ = ok[q] ? gix[q] * mx[q] : (R)0, gy_ = ok[q] ? giy[q] * my[q] : (R)0;
            R ax, ay, az;
            km_gm_terms<CM>(cd[q], gx_, gy_, ax, ay, az);
            S[0] += ax; S[1] += ay; S[2] += az;
            Sv[0] = km_fma(ax, cd[q].v, Sv[0]); Sv[1] = km_fma(ay, cd[q].v, Sv[1]); Sv[2] = km_fma(az, cd[q].v, Sv[2]);
        }
    }
    R gm[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gm[3 * k + 0] = S[k] * u;
        gm[3 * k + 1] = Sv[k];
        gm[3 * k + 2] = S[k];
    }
    if (CM == KM_COORD_AFFINE) gm[6] = gm[7] = gm[8] = 0;

#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double s = km_wave_sum((double)gm[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (s != 0.0) km_atomic_add(a.gmat + (size_t)(g.B_M == 1 ? 0 : b) * 9 + threadIdx.x, s);
    }
}

template <typename T, int CM>
static int kmg_launch(const KmWarpGmArgs<T>& a, hipStream_t s) {
    if (a.g.C == 3)
        hipLaunchKernelGGL((km_warp_gm_kernel<T, CM, 3>), dim3(a.nblocks), dim3(256), 0, s, a);
    else if (a.g.C == 1)
        hipLaunchKernelGGL((km_warp_gm_kernel<T, CM, 1>), dim3(a.nblocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((km_warp_gm_kernel<T, CM, 0>), dim3(a.nblocks), dim3(256), 0, s, a);
    return km_check_launch("km_warp2d_bwd(matrix gradient)");
}

template <typename T>
static int kmg_run(const void* gout, const void* src, const void* mat, double* gmat, int B, int C, int H, int W, int h, int w, int B_M,
                   int coord_mode, int norm_coords, int pad, int align, const void* fill, hipStream_t s) {
    KmWarpGmArgs<T> a;
    a.src = (const T*)src; a.gout = (const T*)gout; a.mat = (const float*)mat; a.gmat = gmat; a.fill = (const float*)fill;
    KmWarpGeom<float>& g = a.g;
    km_geom_init(g, B, C, H, W, h, w, B_M, coord_mode, norm_coords, KM_INTERP_BILINEAR, pad, align);
    a.tiles_x = (uint32_t)((w + KMG_TILE_W - 1) / KMG_TILE_W);
    a.tiles_y = (uint32_t)((h + KMG_TILE_H - 1) / KMG_TILE_H);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B;
    KM_REQUIRE(nb < (1ull << 31), "km_warp2d_bwd: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    switch (coord_mode) {
        case KM_COORD_PERSPECTIVE: return kmg_launch<T, KM_COORD_PERSPECTIVE>(a, s);
        case KM_COORD_AFFINE: return kmg_launch<T, KM_COORD_AFFINE>(a, s);
        default: return kmg_launch<T, KM_COORD_HOMOGRAPHY>(a, s);
    }
}

// 1 if this kernel computes the matrix gradient for these modes (bilinear, zeros / fill padding, fp32 compute)
int km_warp_gm_supported(int interp, int pad, int dtype, int H, int W, int h, int w) {
    static int disabled = -1;
    if (disabled < 0) {
        const char* e = getenv("KM_WARP_GM_ALGO");  // "generic": the atomic scatter kernel computes it (A/B timing)
        disabled = (e && e[0] == 'g') ? 1 : 0;
    }
    if (disabled) return 0;
    if (!(interp == KM_INTERP_BILINEAR && (pad == KM_PAD_ZEROS || pad == KM_PAD_FILL) && dtype != KM_F64)) return 0;
    if (W < 2) return 0;  // the pair loads need two columns
    // 32-bit byte offsets inside a plane
    return ((uint64_t)H * W * 4 < (1ull << 32) && (uint64_t)h * w * 4 < (1ull << 32)) ? 1 : 0;
}

int km_warp_gm_run(const void* gout, const void* src, const void* mat, double* gmat, int B, int C, int H, int W, int h, int w, int B_M,
                   int coord_mode, int norm_coords, int pad, int align, const void* fill, int dtype, hipStream_t s) {
    switch (dtype) {
        case KM_F32: return kmg_run<float>(gout, src, mat, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, s);
        case KM_BF16: return kmg_run<km_bf16>(gout, src, mat, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, s);
        default: return kmg_run<km_f16>(gout, src, mat, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, s);
    }
}
