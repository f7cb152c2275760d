// kornia_amd - staging of the source box of an output tile in LDS, shared by the LDS-staged forward (km_warp.hip) and
// matrix-gradient (km_warp_gm.hip) kernels.
//
// A 256-thread block owns a KMF_T x KMF_T tile of the OUTPUT.  kmf_tile_setup tabulates the row halves of the coordinate
// numerators and finds the box of source pixels the tile's footprint covers: the four tile corners pushed through the
// forward's own coordinate arithmetic, + 1 px on every side (a projective map whose denominator keeps its sign sends the
// tile to a convex quad, so the corners bound it up to rounding).  kmf_stage_box copies the box - all channels - into LDS
// with 16-byte row loads, zeros outside the image (what zeros padding samples there).  Every pixel then checks that its
// 2 x 2 footprint lies in the box (kmf_in_box) and reads its taps from LDS (kmf_tap_ptr); callers fall back to gathers
// from global memory when the box does not fit or a footprint is not covered, so results never depend on the estimate.
#pragma once

#include "km_lean.h"
#include "km_regtile.h"

#ifndef KMF_T
#define KMF_T 32        // output tile (square: the source box stays small under rotation)
#endif
#ifndef KMF_PITCH
#define KMF_PITCH 56    // floats per staged source row (box width + alignment slack), multiple of 4
#endif
#ifndef KMF_ROWS
#define KMF_ROWS 56     // staged source rows
#endif
#define KMF_RPT (KMF_T * KMF_T / 256)   // output rows per thread
#define KMF_RSTEP (256 / KMF_T)          // tile rows between a thread's consecutive rows


struct KmfBox {
    bool fast;    // division operands of every row of the tile are in the safe range (kml_row_guard)
    bool staged;  // the box fits the LDS tile
    int xs, ys;   // first staged source column (multiple of 4) / row
    int nch;      // 4-element chunks per staged row
    int nrows;    // staged rows
    float bxlo, bxhi, bylo, byhi;  // floor(x), floor(y) of a footprint inside the box
};

// wave 0: row table (m1 v, m4 v, m7 v, keep_v ? v : 0) + division guard; wave 1: the source box.  Needs a barrier before kmf_read_box.
// MARGIN: extra source pixels around the 2 x 2 footprint that the consumer reads (bicubic: 1 -> columns floor(x) - 1 .. floor(x) + 2)
// TW x TH: the output tile; PITCH x ROWS: the capacity of the staged box (the defaults are the bilinear kernels' 32 x 32 / 56 x 56)
template <int CM, int ALIGN, int MARGIN = 0, int TW = KMF_T, int TH = KMF_T, int PITCH = KMF_PITCH, int ROWS = KMF_ROWS>
__device__ __forceinline__ void kmf_tile_setup(const KmWarpGeom<float>& g, const float (&m)[9], int j0, int i0, float4* s_rv, int* s_info, bool keep_v) {
    const int tid_ = km_tid_pinned(), lane = tid_ & 63, wave = tid_ >> 6;  // (pinned: callers walk several tiles in a loop - km_tid_pinned)
    const int W = g.W, H = g.H;
    const float Wm1 = (float)(W - 1), Hm1 = (float)(H - 1), hW = (float)W / 2, hH = (float)H / 2;
    static_assert(TH <= 64, "one wave fills the row table");
    const int j1 = min(j0 + TW - 1, g.w - 1), i1 = min(i0 + TH - 1, g.h - 1);
    if (wave == 0) {
        bool ok = true;
        if (lane < TH) {
            const float v = km_base_y<float, CM>(g, i0 + lane);
            const KmlHalf h = kml_row_half<CM>(m, v);
            s_rv[lane] = make_float4(h.a, h.b, h.c, keep_v ? v : 0.f);
            ok = kml_row_guard<CM>(g, m, v);
        }
        const bool all_ok = __all(ok);
        if (lane == 0) s_info[0] = all_ok ? 1 : 0;
    } else if (wave == 1) {
        // lanes 0..3 take one corner each (plain IEEE divisions: any operands)
        const int jc = (lane & 1) ? j1 : j0, ic = (lane & 2) ? i1 : i0;
        const KmlHalf cu = kml_col_half<CM>(m, km_base_x<float, CM>(g, jc));
        const KmlHalf rv = kml_row_half<CM>(m, km_base_y<float, CM>(g, ic));
        KmlPos p;
        kml_position<CM, false>(m, cu, rv, p);
        const float x = kml_unnormalize<ALIGN>(p.gx, Wm1, hW), y = kml_unnormalize<ALIGN>(p.gy, Hm1, hH);
        const float dsel = (CM == KM_COORD_AFFINE) ? 1.0f : p.den;  // the sign of the denominator (homography mode: of s = 1 / (Z + eps))
        float xmin = x, xmax = x, ymin = y, ymax = y, dmin = dsel, dmax = dsel;
#pragma unroll
        for (int off = 1; off <= 2; off <<= 1) {
            xmin = fminf(xmin, __shfl_down(xmin, off, 64)); xmax = fmaxf(xmax, __shfl_down(xmax, off, 64));
            ymin = fminf(ymin, __shfl_down(ymin, off, 64)); ymax = fmaxf(ymax, __shfl_down(ymax, off, 64));
            dmin = fminf(dmin, __shfl_down(dmin, off, 64)); dmax = fmaxf(dmax, __shfl_down(dmax, off, 64));
        }
        const bool finite4 = __all((lane > 3) || ((x == x) && (y == y) && (km_fabs(x) < 1.0e8f) && (km_fabs(y) < 1.0e8f) && (dsel == dsel)));
        if (lane == 0) {
            // taps of a pixel at (x, y): columns floor(x), floor(x) + 1; one more on each side for the rounding of positions
            // inside the quad relative to its corners (<< 1 px) - checked per pixel anyway
            const int bx0 = (int)km_floor(xmin) - 1 - MARGIN, bx1 = (int)km_floor(xmax) + 2 + MARGIN;
            const int by0 = (int)km_floor(ymin) - 1 - MARGIN, by1 = (int)km_floor(ymax) + 2 + MARGIN;
            const int xs = bx0 & ~3;  // 16-byte aligned start (two's complement: rounds toward -inf)
            const int wcols = bx1 - xs + 1, nrows = by1 - by0 + 1;
            const bool same_sign = (dmin > 0.f) || (dmax < 0.f);
            const bool staged = finite4 && same_sign && wcols <= PITCH && nrows <= ROWS;
            s_info[1] = staged ? 1 : 0;
            s_info[2] = xs;
            s_info[3] = by0;
            s_info[4] = (wcols + 3) >> 2;
            s_info[5] = nrows;
        }
        // tilt: source rows spanned by the tile's first OUTPUT row (lanes 0 and 1 hold its two ends) - what a gather kernel whose wave
        // instructions cover output rows pays for under rotation
        const float ytilt = km_fabs(__shfl_down(y, 1, 64) - y);  // (meaningful in lane 0)
        if (lane == 0) s_info[6] = (ytilt == ytilt && ytilt < 1.0e6f) ? (int)ytilt : 1000000;
    }
}

__device__ __forceinline__ KmfBox kmf_read_box(const int* s_info) {
    KmfBox b;
    b.fast = __builtin_amdgcn_readfirstlane(s_info[0]) != 0;
    b.staged = __builtin_amdgcn_readfirstlane(s_info[1]) != 0;
    b.xs = __builtin_amdgcn_readfirstlane(s_info[2]);
    b.ys = __builtin_amdgcn_readfirstlane(s_info[3]);
    b.nch = __builtin_amdgcn_readfirstlane(s_info[4]);
    b.nrows = __builtin_amdgcn_readfirstlane(s_info[5]);
    b.bxlo = (float)b.xs; b.bxhi = (float)(b.xs + 4 * b.nch - 2);
    b.bylo = (float)b.ys; b.byhi = (float)(b.ys + b.nrows - 2);
    return b;
}

// copy the box into s_src[row][channel][x] (KMF_PITCH floats per channel row): 16 lanes x 4 elements per source row,
// 16 rows per pass of the block; all loads of a thread are issued before the first LDS store.  Needs a barrier after.
// oob[c]: the value staged outside the image (0 for zeros padding; the fill value where the consumer subtracts it from every tap)
template <typename T, int NC>
__device__ __forceinline__ void kmf_stage_box(const T* __restrict__ src_b, size_t src_plane, int W, int H, const KmfBox& bx, float* s_src,
                                              const float (&oob)[NC]) {
    static_assert(KMF_T == 32 && (KMF_PITCH % 4) == 0 && KMF_PITCH / 4 <= 16, "16 lanes x 16 bytes cover a staged row");
    const int tid = threadIdx.x, ck = tid & 15, r0 = tid >> 4;
    const int x = bx.xs + 4 * ck;
    const bool col_in = (ck < bx.nch) && (x >= 0) && (x + 3 < W);  // W % 4 == 0 and xs % 4 == 0: a chunk is inside or outside as a whole
    constexpr int NPASS = (KMF_ROWS + 15) / 16;
    float v[NPASS][NC][4];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int r = r0 + 16 * ps, y = bx.ys + r;
        const bool inb = col_in && (r < bx.nrows) && (y >= 0) && (y < H);
        const uint32_t off = inb ? (uint32_t)y * (uint32_t)W + (uint32_t)x : 0u;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (inb) km_ld4(km_at(src_b + c * src_plane, off), v[ps][c]);
            else { v[ps][c][0] = oob[c]; v[ps][c][1] = oob[c]; v[ps][c][2] = oob[c]; v[ps][c][3] = oob[c]; }
        }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int r = r0 + 16 * ps;
        if ((ck < bx.nch) && (r < bx.nrows)) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float* q = s_src + (r * NC + c) * KMF_PITCH + 4 * ck;
                KM_CHECK_ALIGNED(q, 16);
                *reinterpret_cast<float4*>(q) = make_float4(v[ps][c][0], v[ps][c][1], v[ps][c][2], v[ps][c][3]);
            }
        }
    }
}

// The same copy for any box capacity (PITCH floats per staged row): 16 or 32 lanes per source row, chosen per block from the
// box width, four row loads of a thread in flight before their LDS stores.  Layout s_src[row][channel][PITCH].
template <typename T, int NC, int PITCH>
__device__ __forceinline__ void kmf_stage_box_dyn(const T* __restrict__ src_b, size_t src_plane, int W, int H, const KmfBox& bx, float* s_src,
                                                  const float (&oob)[NC]) {
    static_assert((PITCH % 4) == 0 && PITCH / 4 <= 32, "32 lanes x 16 bytes cover a staged row");
    const int shift = bx.nch <= 16 ? 4 : 5;  // block-uniform
    const int tid = threadIdx.x, ck = tid & ((1 << shift) - 1), r0 = tid >> shift, rpp = 256 >> shift;
    const int x = bx.xs + 4 * ck;
    const bool col_in = (ck < bx.nch) && (x >= 0) && (x + 3 < W);  // W % 4 == 0 and xs % 4 == 0: a chunk is inside or outside as a whole
    for (int base = 0; base < bx.nrows; base += 4 * rpp) {
        float v[4][NC][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = base + r0 + k * rpp, y = bx.ys + r;
            const bool inb = col_in && (r < bx.nrows) && (y >= 0) && (y < H);
            const uint32_t off = inb ? (uint32_t)y * (uint32_t)W + (uint32_t)x : 0u;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (inb) km_ld4(km_at(src_b + c * src_plane, off), v[k][c]);
                else { v[k][c][0] = oob[c]; v[k][c][1] = oob[c]; v[k][c][2] = oob[c]; v[k][c][3] = oob[c]; }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = base + r0 + k * rpp;
            if ((ck < bx.nch) && (r < bx.nrows)) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float* q = s_src + (r * NC + c) * PITCH + 4 * ck;
                    KM_CHECK_ALIGNED(q, 16);
                    *reinterpret_cast<float4*>(q) = make_float4(v[k][c][0], v[k][c][1], v[k][c][2], v[k][c][3]);
                }
            }
        }
    }
}

// The same copy by LDS-DMA (fp32 storage, zeros outside the image): thread t's 16-byte chunk of a pass lands at chunk t of the staged box -
// a pass is RCPP consecutive (row, channel) pairs of NCHK = PITCH / 4 chunks, the layout s_src[row][channel][PITCH] unchanged - straight
// from the memory pipe (KM_GLDS16, km_common.h): ALL passes are in flight together (the register copy above waits for four rows at a time)
// and no register holds the box.  Chunks outside the image are zeros written the ordinary way.  The caller makes the requests land:
// KM_VMCNT0() and a barrier.
template <int NC, int PITCH, int ROWS>
__device__ __forceinline__ void kmf_stage_box_dma(const float* __restrict__ src_b, size_t src_plane, int W, int H, const KmfBox& bx, float* s_src) {
    static_assert((PITCH % 4) == 0 && PITCH / 4 <= 64, "whole chunks; a staged row fits a wave instruction");
    constexpr int NCHK = PITCH / 4, RCPP = 256 / NCHK, NPASS = (ROWS * NC + RCPP - 1) / RCPP;
    const int tid = km_tid_pinned(), ck = tid % NCHK, rq = tid / NCHK;
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int xg = bx.xs + 4 * ck;
    const bool filler = (rq < RCPP) && (ck < bx.nch);
    const bool col_in = (xg >= 0) && (xg + 3 < W);  // W % 4 == 0 and xs % 4 == 0: a chunk is inside or outside as a whole
    const int nrc = bx.nrows * NC;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int rc = ps * RCPP + rq, r = rc / NC, c = rc - r * NC, y = bx.ys + r;
        const bool live = filler && (rc < nrc);
        const bool inb = live && col_in && (y >= 0) && (y < H);
        if (inb) KM_GLDS16(km_at(src_b + c * src_plane, (uint32_t)y * (uint32_t)W + (uint32_t)xg), s_src + ps * (RCPP * PITCH) + 4 * wbase);
        else if (live) *reinterpret_cast<float4*>(s_src + rc * PITCH + 4 * ck) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// the 2 x 2 footprint of a pixel lies in the staged box (NaN / inf positions do not)
__device__ __forceinline__ bool kmf_in_box(const KmlTaps& t, const KmfBox& bx) {
    return (t.xf >= bx.bxlo) & (t.xf <= bx.bxhi) & (t.yf >= bx.bylo) & (t.yf <= bx.byhi);
}
// the 4 x 4 footprint of a bicubic sample at floor (xf, yf) lies in the staged box
__device__ __forceinline__ bool kmf_in_box_cubic(float xf, float yf, const KmfBox& bx) {
    return (xf >= bx.bxlo + 1.0f) & (xf <= bx.bxhi - 1.0f) & (yf >= bx.bylo + 1.0f) & (yf <= bx.byhi - 1.0f);
}
// LDS address of the north-west tap of channel 0 (channel c: + c * KMF_PITCH ; next row: + NC * KMF_PITCH); !valid: the box origin
template <int NC>
__device__ __forceinline__ const float* kmf_tap_ptr(const float* s_src, const KmlTaps& t, const KmfBox& bx, bool valid) {
    const int xi = valid ? KM_F2I(t.xf) - bx.xs : 0, yi = valid ? KM_F2I(t.yf) - bx.ys : 0;
    return s_src + __mul24(yi, NC * KMF_PITCH) + xi;
}
