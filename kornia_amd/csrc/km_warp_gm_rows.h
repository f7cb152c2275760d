// kornia_amd - the per-thread rows of the matrix gradient of the bilinear warps and its block reduction, shared by the
// matrix-gradient kernel (km_warp_gm.hip) and its LDS-staged variant; kmg_terms also serves the one-read backward (km_warp_bwd_fused.hip).
#pragma once

#include "km_warp_stage.h"

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
#ifndef KMG_BOUNDS
#define KMG_BOUNDS __launch_bounds__(256)
#endif

template <typename T>
struct KmWarpGmArgs {
    const T* src;       // (B,C,H,W)
    const T* gout;      // (B,C,h,w)
    const float* mat;   // (B_M,9)
    double* gmat;       // (B_M,9) fp64 accumulators, pre-zeroed
    const float* fill;  // (C), pad == fill only
    KmWarpGeom<float> g;
    uint32_t tiles_x, tiles_y, nblocks;
    uint32_t reverse;   // the XCDs walk their block ranges backwards (km_traversal_next)
};

// matrix-gradient terms of one output pixel from the lean position record (km_gm_terms on KmlPos)
template <int CM, bool FAST>
__device__ __forceinline__ void kmg_terms(const KmlPos& p, float gix, float giy, float& ax, float& ay, float& az) {
    if (CM == KM_COORD_PERSPECTIVE) {
        const float inv = FAST ? p.rinv : __frcp_rn(p.den);  // the refined reciprocal is within 1 ulp of 1 / den
        ax = gix * inv;
        ay = giy * inv;
        az = -km_fma(gix, p.gx, giy * p.gy) * inv;
    } else if (CM == KM_COORD_AFFINE) {
        ax = gix;
        ay = giy;
        az = 0;
    } else {
        const float s = p.den;
        ax = gix * s;
        ay = giy * s;
        az = p.live ? -km_fma(gix, p.X, giy * p.Y) * s * s : 0.0f;
    }
}

template <typename T, int CM, int NC, int ALIGN, bool FAST, int NROWS = KMG_ROWS, int PH = 64 / KMG_PATCH_W, int GROUP = KMG_GROUP>  // rows per thread, tile rows between them, rows in flight
__device__ __forceinline__ void km_warp_gm_rows(const KmWarpGmArgs<T>& a, const float (&m)[9], const float4* s_rv, uint32_t b, int j, int li_base, int i_base,
                                                float (&S)[3], float (&Sv)[3]) {
    const KmWarpGeom<float>& g = a.g;
    const int W = g.W, H = g.H;
    const int C = (NC > 0) ? NC : g.C;
    const size_t src_plane = (size_t)H * W, dst_plane = (size_t)g.h * g.w;
    const T* __restrict__ src_b = a.src + (size_t)b * C * src_plane;
    const T* __restrict__ gout_b = a.gout + (size_t)b * C * dst_plane;
    const bool col_ok = j < g.w;
    const float Wm1 = (float)(W - 1), Hm1 = (float)(H - 1), hW = (float)W / 2, hH = (float)H / 2, Wm2 = (float)(W - 2), Hm2 = (float)(H - 2);
    const float mx = ALIGN ? Wm1 / 2 : hW, my = ALIGN ? Hm1 / 2 : hH;  // d (pixel) / d (normalised): km_unnormalize's multiplier
    const KmlHalf cu = kml_col_half<CM>(m, km_base_x<float, CM>(g, col_ok ? j : 0));
    const bool is_fill = (g.pad == KM_PAD_FILL);

    // rows are taken GROUP at a time: all their sampling positions first, then - when every lane samples inside the
    // image for all of them - all their loads back to back before the first use
    for (int r0 = 0; r0 < NROWS; r0 += GROUP) {
        KmlPos p[GROUP];
        KmlTaps t[GROUP];
        float xs[GROUP], ys[GROUP], vrow[GROUP], gix[GROUP], giy[GROUP];
        uint32_t go_off[GROUP];
        bool ok[GROUP];
        bool inside = true;
#pragma unroll
        for (int q = 0; q < GROUP; ++q) {
            const int i = i_base + (r0 + q) * PH;
            ok[q] = col_ok & (i < g.h);
            const float4 rv4 = s_rv[li_base + (r0 + q) * PH];
            KmlHalf rv;
            rv.a = rv4.x; rv.b = rv4.y; rv.c = rv4.z;
            vrow[q] = rv4.w;
            kml_position<CM, FAST>(m, cu, rv, p[q]);
            xs[q] = kml_unnormalize<ALIGN>(p[q].gx, Wm1, hW);
            ys[q] = kml_unnormalize<ALIGN>(p[q].gy, Hm1, hH);
            kml_taps(xs[q], ys[q], t[q]);
            go_off[q] = ok[q] ? (uint32_t)i * (uint32_t)g.w + (uint32_t)j : 0u;
            inside = inside & kml_inside(t[q], Wm2, Hm2);
            gix[q] = 0;
            giy[q] = 0;
        }
        if (NC > 0 && __all(inside)) {
            constexpr int NCC = NC > 0 ? NC : 1;
            const T* __restrict__ sp[NCC];
            const T* __restrict__ gp[NCC];
#pragma unroll
            for (int c = 0; c < NCC; ++c) { sp[c] = src_b + c * src_plane; gp[c] = gout_b + c * dst_plane; }  // wave-uniform plane bases
            // (taking the x0 + 1 column from the next lane, as the forward does, measured slower here: 0.40 vs 0.38 ms.  Ablations on one
            // box, wrong results, timing only: 0.397 ms as is; 4-byte instead of 8-byte tap loads 0.345; one source row 0.349; no source
            // taps at all - grad_out, the position arithmetic and the reductions only - 0.266: two thirds of the kernel's time is not its
            // tap gathers but the dependent chain position -> address -> load -> terms of each group of rows.)
            float go[GROUP][NCC], v[GROUP][NCC][4];
#pragma unroll
            for (int q = 0; q < GROUP; ++q) {
                const uint32_t off = (uint32_t)__mul24((int)t[q].yf, W) + (uint32_t)(int)t[q].xf;
#pragma unroll
                for (int c = 0; c < NCC; ++c) {
                    go[q][c] = (float)km_ld(km_at(gp[c], go_off[q]));
                    km_ld2(km_at(sp[c], off), v[q][c][0], v[q][c][1]);
                    km_ld2(km_at(sp[c], off + (uint32_t)W), v[q][c][2], v[q][c][3]);
                }
            }
#pragma unroll
            for (int q = 0; q < GROUP; ++q)
#pragma unroll
                for (int c = 0; c < NCC; ++c) {
                    float s00 = v[q][c][0], s01 = v[q][c][1], s10 = v[q][c][2], s11 = v[q][c][3];
                    if (is_fill) {  // same rounding sequence as the oracle: (v - fill) first
                        const float f = a.fill[c];
                        s00 -= f; s01 -= f; s10 -= f; s11 -= f;
                    }
                    gix[q] = km_fma(go[q][c], km_fma(s01 - s00, t[q].wy1, (s11 - s10) * t[q].wy0), gix[q]);
                    giy[q] = km_fma(go[q][c], km_fma(s10 - s00, t[q].wx1, (s11 - s01) * t[q].wx0), giy[q]);
                }
        } else {
#pragma unroll
            for (int q = 0; q < GROUP; ++q) {
                KmBilin<float> tq;
                km_bilinear_setup(xs[q], ys[q], W, H, tq);
                if (__all(tq.b00 && tq.b01 && tq.b10 && tq.b11)) {
                    // the whole wave samples inside the image for this row: (x0, x0 + 1) come with one load per row
                    for (int c = 0; c < C; ++c) {
                        const float gv = (float)km_ld(km_at(gout_b + (size_t)c * dst_plane, go_off[q]));
                        const T* img = src_b + (size_t)c * src_plane;
                        float s00, s01, s10, s11;
                        km_ld2(km_at(img, (uint32_t)tq.i00), s00, s01);
                        km_ld2(km_at(img, (uint32_t)tq.i10), s10, s11);
                        if (is_fill) {
                            const float f = a.fill[c];
                            s00 -= f; s01 -= f; s10 -= f; s11 -= f;
                        }
                        gix[q] = km_fma(gv, km_fma(s01 - s00, tq.wy1, (s11 - s10) * tq.wy0), gix[q]);
                        giy[q] = km_fma(gv, km_fma(s10 - s00, tq.wx1, (s11 - s01) * tq.wx0), giy[q]);
                    }
                } else {
                    for (int c = 0; c < C; ++c) {
                        const float gv = (float)km_ld(km_at(gout_b + (size_t)c * dst_plane, go_off[q]));
                        const T* img = src_b + (size_t)c * src_plane;
                        const float f = is_fill ? a.fill[c] : 0.0f;
                        // unconditional loads (clamped indices); out-of-bounds taps do not exist in the reference's sum
                        const float v00 = (float)km_ld(img + tq.i00), v01 = (float)km_ld(img + tq.i01), v10 = (float)km_ld(img + tq.i10), v11 = (float)km_ld(img + tq.i11);
                        const float s00 = tq.b00 ? v00 - f : 0.0f, s01 = tq.b01 ? v01 - f : 0.0f;
                        const float s10 = tq.b10 ? v10 - f : 0.0f, s11 = tq.b11 ? v11 - f : 0.0f;
                        gix[q] = km_fma(gv, km_fma(s01 - s00, tq.wy1, (s11 - s10) * tq.wy0), gix[q]);
                        giy[q] = km_fma(gv, km_fma(s10 - s00, tq.wx1, (s11 - s01) * tq.wx0), giy[q]);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < GROUP; ++q) {
            // pixels outside the output (padding lanes / rows of the last tiles) contribute nothing
            const float gx_ = ok[q] ? gix[q] * mx : 0.0f, gy_ = ok[q] ? giy[q] * my : 0.0f;
            float ax, ay, az;
            kmg_terms<CM, FAST>(p[q], gx_, gy_, ax, ay, az);
            // select the TERMS, not only their inputs: a padding lane / row whose position is not finite (den == 0 beyond the output's
            // last row) would otherwise add 0 * inf = NaN to the whole image's gradient
            ax = ok[q] ? ax : 0.0f; ay = ok[q] ? ay : 0.0f; az = ok[q] ? az : 0.0f;
            S[0] += ax; S[1] += ay; S[2] += az;
            Sv[0] = km_fma(ax, vrow[q], Sv[0]); Sv[1] = km_fma(ay, vrow[q], Sv[1]); Sv[2] = km_fma(az, vrow[q], Sv[2]);
        }
    }
}

// block reduction of the per-thread sums and the 9 fp64 atomics of the block (u = this thread's column coordinate).
// tid: index of the thread in its group of 256 (the one-launch backward runs two such groups per workgroup, each with its own `red`);
// the barrier is the workgroup's, so every thread of the workgroup must call this.
template <int CM>
__device__ __forceinline__ void kmg_block_reduce(const float (&S)[3], const float (&Sv)[3], float u, double* gmat_b, double (*red)[9], int tid = threadIdx.x) {
    const int lane = tid & 63, wave = tid >> 6;
    float gm[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gm[3 * k + 0] = S[k] * u;
        gm[3 * k + 1] = Sv[k];
        gm[3 * k + 2] = S[k];
    }
    if (CM == KM_COORD_AFFINE) gm[6] = gm[7] = gm[8] = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double s = km_wave_sum_last((double)gm[k]);  // (DPP: valid in lane 63)
        if (lane == 63) red[wave][k] = s;
    }
    __syncthreads();
    if (tid < 9) {
        const double s = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        if (s != 0.0) km_atomic_add(gmat_b + tid, s);
    }
}
