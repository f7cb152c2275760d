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

template <typename T, int CM, int NC, int ALIGN, bool FAST, int NROWS = KMG_ROWS, int PH = 64 / KMG_PATCH_W>  // rows per thread, tile rows between them
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

    // rows are taken KMG_GROUP at a time: all their sampling positions first, then - when every lane samples inside the
    // image for all of them - all their loads back to back before the first use
    for (int r0 = 0; r0 < NROWS; r0 += KMG_GROUP) {
        KmlPos p[KMG_GROUP];
        KmlTaps t[KMG_GROUP];
        float xs[KMG_GROUP], ys[KMG_GROUP], vrow[KMG_GROUP], gix[KMG_GROUP], giy[KMG_GROUP];
        uint32_t go_off[KMG_GROUP];
        bool ok[KMG_GROUP];
        bool inside = true;
#pragma unroll
        for (int q = 0; q < KMG_GROUP; ++q) {
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
            // (taking the x0 + 1 column from the next lane, as the forward does, measured slower here: 0.40 vs 0.38 ms)
            float go[KMG_GROUP][NCC], v[KMG_GROUP][NCC][4];
#pragma unroll
            for (int q = 0; q < KMG_GROUP; ++q) {
                const uint32_t off = (uint32_t)__mul24((int)t[q].yf, W) + (uint32_t)(int)t[q].xf;
#pragma unroll
                for (int c = 0; c < NCC; ++c) {
                    go[q][c] = (float)km_ld(km_at(gp[c], go_off[q]));
                    km_ld2(km_at(sp[c], off), v[q][c][0], v[q][c][1]);
                    km_ld2(km_at(sp[c], off + (uint32_t)W), v[q][c][2], v[q][c][3]);
                }
            }
#pragma unroll
            for (int q = 0; q < KMG_GROUP; ++q)
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
            for (int q = 0; q < KMG_GROUP; ++q) {
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
        for (int q = 0; q < KMG_GROUP; ++q) {
            // pixels outside the output (padding lanes / rows of the last tiles) contribute nothing
            const float gx_ = ok[q] ? gix[q] * mx : 0.0f, gy_ = ok[q] ? giy[q] * my : 0.0f;
            float ax, ay, az;
            kmg_terms<CM, FAST>(p[q], gx_, gy_, ax, ay, az);
            S[0] += ax; S[1] += ay; S[2] += az;
            Sv[0] = km_fma(ax, vrow[q], Sv[0]); Sv[1] = km_fma(ay, vrow[q], Sv[1]); Sv[2] = km_fma(az, vrow[q], Sv[2]);
        }
    }
}

// block reduction of the per-thread sums and the 9 fp64 atomics of the block (u = this thread's column coordinate)
template <int CM>
__device__ __forceinline__ void kmg_block_reduce(const float (&S)[3], const float (&Sv)[3], float u, double* gmat_b, double (*red)[9]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
        const double s = km_wave_sum((double)gm[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (s != 0.0) km_atomic_add(gmat_b + threadIdx.x, s);
    }
}

template <typename T, int CM, int NC, int ALIGN>  // NC = 3 / 1: RGB / grey unrolled ; NC = 0: runtime channel loop
__global__ KMG_BOUNDS void km_warp_gm_kernel(const KmWarpGmArgs<T> a) {
    const KmWarpGeom<float>& g = a.g;
    __shared__ double red[4][9];
    __shared__ float4 s_rv[KMG_TILE_H];  // per row of the tile: (m1 v, m4 v, m7 v, v)
    __shared__ int s_fast;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks, a.reverse);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    // a wave instruction covers a KMG_PATCH_W x (64 / KMG_PATCH_W) patch of the output (see km_warp_fwd_lean_kernel)
    constexpr int PW = KMG_PATCH_W, PH = 64 / PW, WA = 64 / PW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = (int)tx * KMG_TILE_W + (wave % WA) * PW + (lane % PW);
    const int li_base = (wave / WA) * (PH * KMG_ROWS) + lane / PW;  // row r of this thread sits at tile row li_base + r * PH
    const int i_base = (int)ty * KMG_TILE_H + li_base;

    float m[9];
    {
        const float* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = mp[k];
    }
    static_assert(KMG_TILE_H <= 64, "the row table is filled by one wave");
    if (wave == 0) {  // row halves of the numerators + the per-row division guard, combined over the tile's rows
        bool okr = true;
        if (lane < KMG_TILE_H) {
            const float v = km_base_y<float, CM>(g, (int)ty * KMG_TILE_H + lane);
            const KmlHalf h = kml_row_half<CM>(m, v);
            s_rv[lane] = make_float4(h.a, h.b, h.c, v);
            okr = kml_row_guard<CM>(g, m, v);
        }
        const bool all_ok = __all(okr);
        if (lane == 0) s_fast = all_ok ? 1 : 0;
    }
    __syncthreads();

    // u is fixed per thread (lane = column): accumulate S = sum(ax, ay, az) and Sv = sum(v * (ax, ay, az)) over the
    // thread's rows and multiply by u once at the end
    float S[3] = {0, 0, 0}, Sv[3] = {0, 0, 0};
    if (__builtin_amdgcn_readfirstlane(s_fast))
        km_warp_gm_rows<T, CM, NC, ALIGN, true>(a, m, s_rv, b, j, li_base, i_base, S, Sv);
    else
        km_warp_gm_rows<T, CM, NC, ALIGN, false>(a, m, s_rv, b, j, li_base, i_base, S, Sv);
    kmg_block_reduce<CM>(S, Sv, km_base_x<float, CM>(g, j < g.w ? j : 0), a.gmat + (size_t)(g.B_M == 1 ? 0 : b) * 9, red);
}

// LDS-staged matrix gradient (RGB / grey): the forward-shaped reduction with the source taps read from a staged box instead
// of gathered from global memory (km_warp_stage.h) - the gather kernel above is bound by the texture-address path like the
// gather forward.  Per pixel the memory pipeline sees NC coalesced grad_out loads instead of those + 2 NC gathers.
template <typename T, int CM, int NC, int ALIGN>
__global__ __launch_bounds__(256) void km_warp_gm_lds_kernel(const KmWarpGmArgs<T> a) {
    const KmWarpGeom<float>& g = a.g;
    __shared__ double red[4][9];
    __shared__ float4 s_rv[KMF_T];  // per row of the tile: (m1 v, m4 v, m7 v, v)
    __shared__ int s_info[8];
    __shared__ __attribute__((aligned(16))) float s_src[KMF_ROWS * NC * KMF_PITCH];
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    const int tid = threadIdx.x;
    const int j = (int)tx * KMF_T + (tid % KMF_T);
    const int li_base = tid / KMF_T;
    const int i_base = (int)ty * KMF_T + li_base;

    float m[9];
    {
        const float* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = mp[k];
    }
    kmf_tile_setup<CM, ALIGN>(g, m, (int)tx * KMF_T, (int)ty * KMF_T, s_rv, s_info, true);
    __syncthreads();
    const KmfBox bx = kmf_read_box(s_info);
    const bool col_ok = j < g.w;
    const float u = km_base_x<float, CM>(g, col_ok ? j : 0);
    double* gmat_b = a.gmat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
    float S[3] = {0, 0, 0}, Sv[3] = {0, 0, 0};
    bool done = false;
    if (bx.staged) {  // block-uniform
        const int W = g.W, H = g.H;
        const float Wm1 = (float)(W - 1), Hm1 = (float)(H - 1), hW = (float)W / 2, hH = (float)H / 2;
        const float mx = ALIGN ? Wm1 / 2 : hW, my = ALIGN ? Hm1 / 2 : hH;
        const size_t src_plane = (size_t)H * W, dst_plane = (size_t)g.h * g.w;
        const T* __restrict__ src_b = a.src + (size_t)b * NC * src_plane;
        const T* __restrict__ gout_b = a.gout + (size_t)b * NC * dst_plane;
        const bool is_fill = (g.pad == KM_PAD_FILL);
        float oob[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) oob[c] = is_fill ? a.fill[c] : 0.f;  // (v - fill) of a tap outside the image is 0: it does not exist
        // grad_out of this thread's pixels: issued before the staging so that both are in flight together
        float go[KMF_RPT][NC];
        bool ok[KMF_RPT];
#pragma unroll
        for (int r = 0; r < KMF_RPT; ++r) {
            const int i = i_base + r * KMF_RSTEP;
            ok[r] = col_ok & (i < g.h);
            const uint32_t off = ok[r] ? (uint32_t)i * (uint32_t)g.w + (uint32_t)j : 0u;
#pragma unroll
            for (int c = 0; c < NC; ++c) go[r][c] = (float)km_ld(km_at(gout_b + c * dst_plane, off));
        }
        kmf_stage_box<T, NC>(src_b, src_plane, W, H, bx, s_src, oob);
        __syncthreads();
        const KmlHalf cu = kml_col_half<CM>(m, u);
        KmlPos p[KMF_RPT];
        KmlTaps t[KMF_RPT];
        float vrow[KMF_RPT];
        bool inbox = true;
#pragma unroll
        for (int r = 0; r < KMF_RPT; ++r) {
            const float4 rv4 = s_rv[li_base + r * KMF_RSTEP];
            KmlHalf rv;
            rv.a = rv4.x; rv.b = rv4.y; rv.c = rv4.z;
            vrow[r] = rv4.w;
            if (bx.fast) kml_position<CM, true>(m, cu, rv, p[r]);
            else kml_position<CM, false>(m, cu, rv, p[r]);
            kml_taps(kml_unnormalize<ALIGN>(p[r].gx, Wm1, hW), kml_unnormalize<ALIGN>(p[r].gy, Hm1, hH), t[r]);
            inbox = inbox & (kmf_in_box(t[r], bx) | !ok[r]);
        }
        if (__syncthreads_and((int)inbox)) {  // block-uniform (the reduction below has a barrier)
            done = true;
#pragma unroll
            for (int r = 0; r < KMF_RPT; ++r) {
                const float* q0 = kmf_tap_ptr<NC>(s_src, t[r], bx, ok[r]);
                const float* q1 = q0 + NC * KMF_PITCH;
                float gix = 0, giy = 0;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float s00 = q0[c * KMF_PITCH], s01 = q0[c * KMF_PITCH + 1], s10 = q1[c * KMF_PITCH], s11 = q1[c * KMF_PITCH + 1];
                    if (is_fill) {  // same rounding sequence as the oracle: (v - fill) first
                        const float f = a.fill[c];
                        s00 -= f; s01 -= f; s10 -= f; s11 -= f;
                    }
                    gix = km_fma(go[r][c], km_fma(s01 - s00, t[r].wy1, (s11 - s10) * t[r].wy0), gix);
                    giy = km_fma(go[r][c], km_fma(s10 - s00, t[r].wx1, (s11 - s01) * t[r].wx0), giy);
                }
                const float gx_ = ok[r] ? gix * mx : 0.0f, gy_ = ok[r] ? giy * my : 0.0f;
                float ax, ay, az;
                if (bx.fast) kmg_terms<CM, true>(p[r], gx_, gy_, ax, ay, az);
                else kmg_terms<CM, false>(p[r], gx_, gy_, ax, ay, az);
                S[0] += ax; S[1] += ay; S[2] += az;
                Sv[0] = km_fma(ax, vrow[r], Sv[0]); Sv[1] = km_fma(ay, vrow[r], Sv[1]); Sv[2] = km_fma(az, vrow[r], Sv[2]);
            }
        }
    }
    if (!done) {  // the box does not fit / a footprint is not covered: gathers from global memory
        if (bx.fast) km_warp_gm_rows<T, CM, NC, ALIGN, true, KMF_RPT, KMF_RSTEP>(a, m, s_rv, b, j, li_base, i_base, S, Sv);
        else km_warp_gm_rows<T, CM, NC, ALIGN, false, KMF_RPT, KMF_RSTEP>(a, m, s_rv, b, j, li_base, i_base, S, Sv);
    }
    kmg_block_reduce<CM>(S, Sv, u, gmat_b, red);
}

template <typename T, int CM, int NC>
static void kmg_launch_nc(const KmWarpGmArgs<T>& a, hipStream_t s) {
    if (a.g.align)
        hipLaunchKernelGGL((km_warp_gm_kernel<T, CM, NC, 1>), dim3(a.nblocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((km_warp_gm_kernel<T, CM, NC, 0>), dim3(a.nblocks), dim3(256), 0, s, a);
}
static int kmg_algo() {  // KM_WARP_GM_ALGO: "generic" | "lds" (LDS-staged kernel, A/B timing) | default: the lean gather kernel
    static int v = -1;
    if (v < 0) { const char* e = getenv("KM_WARP_GM_ALGO"); v = (e && e[0] == 'l') ? 1 : 0; }
    return v;
}
template <typename T, int CM, int NC>
static void kmg_lds_launch_nc(const KmWarpGmArgs<T>& a, hipStream_t s) {
    KmWarpGmArgs<T> b = a;
    b.tiles_x = (uint32_t)((a.g.w + KMF_T - 1) / KMF_T);
    b.tiles_y = (uint32_t)((a.g.h + KMF_T - 1) / KMF_T);
    b.nblocks = b.tiles_x * b.tiles_y * (uint32_t)a.g.B;
    if (a.g.align)
        hipLaunchKernelGGL((km_warp_gm_lds_kernel<T, CM, NC, 1>), dim3(b.nblocks), dim3(256), 0, s, b);
    else
        hipLaunchKernelGGL((km_warp_gm_lds_kernel<T, CM, NC, 0>), dim3(b.nblocks), dim3(256), 0, s, b);
}
template <typename T, int CM>
static int kmg_launch(const KmWarpGmArgs<T>& a, hipStream_t s) {
    const uint64_t nb = (uint64_t)((a.g.w + KMF_T - 1) / KMF_T) * (uint64_t)((a.g.h + KMF_T - 1) / KMF_T) * (uint64_t)a.g.B;
    if ((a.g.C == 3 || a.g.C == 1) && (a.g.W & 3) == 0 && ((uintptr_t)a.src % (4 * sizeof(T))) == 0 && nb < (1ull << 31) && kmg_algo() == 1) {
        if (a.g.C == 3) kmg_lds_launch_nc<T, CM, 3>(a, s);
        else kmg_lds_launch_nc<T, CM, 1>(a, s);
        return km_check_launch("km_warp2d_bwd(matrix gradient)");
    }
    if (a.g.C == 3) kmg_launch_nc<T, CM, 3>(a, s);
    else if (a.g.C == 1) kmg_launch_nc<T, CM, 1>(a, s);
    else kmg_launch_nc<T, CM, 0>(a, s);
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
    a.reverse = km_traversal_next();
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
