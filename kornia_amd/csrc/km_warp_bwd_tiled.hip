// kornia_amd - "owner-computes" backward of the bilinear warps (zeros / fill padding) for gfx950.
//
// The generic backward (km_warp.hip) scatters 4*C fp32 atomics per output pixel into HBM - what
// ATen's grid_sampler_2d_backward does - and needs grad_src zeroed first: ~5e bytes of HBM traffic
// per element against 3e algorithmic, serialised by the L2 atomic units (4.2 ms at 256x3x512^2).
// LDS float atomics are no better on gfx950 (ds_add_f32 measured at ~2.5 ms for the same work).
// This kernel therefore turns the scatter into an atomic-free, deterministic GATHER:
//
//   * a workgroup OWNS one 32x32 tile of grad_src (all channels); every thread owns 4 of its pixels
//     and accumulates them in registers;
//   * the output pixels q that can touch the tile are found by pushing the tile rectangle (grown by
//     the 1-pixel bilinear footprint) through the inverse map G (source pixel -> output index): a
//     projective map sends the rectangle to a convex quad, so the bounding box of the four mapped
//     corners (+1 px) contains them all;
//   * phase 1: for every q of the box the sampling position (x, y) is computed with EXACTLY the
//     instruction sequence of the forward kernel (km_gen_coord), and staged in LDS together with
//     grad_out[q, c].  The q's whose (clamped) north-west tap lies in this tile also contribute the
//     matrix gradient here (each q exactly once over all tiles);
//   * phase 2: each owned pixel p evaluates G(p) and inspects only the small window of q's around it
//     that can satisfy |x_q - px| < 1 and |y_q - py| < 1; the window half-size is the L1 norm of the
//     rows of the Jacobian of G over the tile (mean-value bound), 3x3 for near-unit scale.  Weights
//     are the forward's own expressions ((x0+1) - x, x - x0), so grad_src = W^T grad_out for the very
//     W the forward applied;
//   * the tile is written once with plain coalesced stores: no memset, no atomics on grad_src, and
//     a fixed summation order (bit-reproducible run to run).
//
// If the tile straddles the vanishing line of G (corner denominators of mixed sign or ~0) the
// pre-image is not a bounded convex quad: that tile falls back to scanning the whole output image
// with LDS atomics (correct, slower; only the tiles crossed by the line pay).
//
// HBM traffic: read grad_out ~1.4x (halo re-reads served by L2), read src once (matrix gradient
// only), write grad_src once  =>  ~3e bytes/element, the algorithmic figure.
#include <stdlib.h>

#include "km_sampler.h"

#define KMT_TW 64
#define KMT_TH 32
#define KMT_PX 4            // source pixels per thread (rows ty, ty+8, ty+16, ty+24)
#define KMT_CC 3            // channels per pass
#define KMT_LDS_BYTES (2 * 256 * 4 + KMT_CC * KMT_TH * KMT_TW * 4)

template <typename T>
struct KmWarpTiledArgs {
    const T* src;
    const T* gout;
    const float* mat;    // (B_M,9)
    float* gsrc;         // (B,C,H,W) fp32, written completely (no pre-zeroing needed)
    double* gmat;        // (B_M,9) fp64 accumulators, pre-zeroed, nullable
    const float* fill;   // (C) for pad == fill
    KmWarpGeom<float> g;
    uint32_t tiles_x, tiles_y, nblocks;
    int lds_bytes;       // dynamic LDS given to the block (staging capacity)
};

template <int CM>
__device__ __forceinline__ void kmt_index_affine(const KmWarpGeom<float>& g, int n, float lo, float step, float& scale, float& offs) {
    // base coordinate u -> output index:  idx = scale * u + offs   (inverse of km_base_x / km_base_y)
    if (CM == KM_COORD_AFFINE) {
        scale = step != 0.0f ? 1.0f / step : 0.0f;
        offs = -lo * scale;
    } else if (CM == KM_COORD_HOMOGRAPHY && !g.norm_coords) {
        scale = 1.0f;
        offs = 0.0f;
    } else {
        scale = 0.5f * (float)(n - 1);
        offs = scale;
    }
}

// ---- the box of output pixels (j0..j1) x (i0..i1) whose bilinear footprint can touch the source tile ----
struct KmtBox {
    int j0, j1, i0, i1;
    float mult;     // bound on the number of output pixels whose footprint covers one source pixel
    bool fixed_ok;  // bounded multiplicity: fixed-point accumulation is accurate enough
};

#ifndef KMT_TIGHT_BOX
#define KMT_TIGHT_BOX 1
#endif

// G = (output index <- source pixel) as a projective map: (Jn, In, D) = G (x, y, 1), (j, i) = (Jn, In) / D.
// A projective map sends the tile rectangle (grown by the 1-pixel footprint) to a convex quad when D keeps
// its sign, so the bounding box of the four mapped corners contains every output pixel that can touch the
// tile - up to rounding.  The margin added around the box is an explicit bound on that rounding:
//   * error of the fp32 inverse map itself (entries of G are differences of products: the bound follows
//     the sums of absolute values, so cancellation is accounted for),
//   * error of the forward fp32 position of a pixel (which is what decides whether it touches the tile),
//     pushed through the Jacobian of G,
// times a safety factor, plus 1/8 px.  (The first version used a flat 1 px + floor/ceil slack, i.e. ~1.5 px
// per side: 8 % more pixels to visit on a 64x32 tile.)
template <int CM>
__device__ __forceinline__ KmtBox kmt_tile_box(const KmWarpGeom<float>& g, const float (&m)[9], int X0, int X1, int Y0, int Y1) {
    typedef float R;
    KmtBox o;
    o.j0 = 0; o.j1 = g.w - 1; o.i0 = 0; o.i1 = g.h - 1;
    o.mult = (R)g.w * (R)g.h;  // whole-output scan: no multiplicity bound
    o.fixed_ok = false;
    R G[9], Ga[9];
    // adjugate of m (un-normalised inverse: the common scale cancels in the projective divide); *a = same with |.|
    const R A0 = m[4] * m[8] - m[5] * m[7], A1 = m[2] * m[7] - m[1] * m[8], A2 = m[1] * m[5] - m[2] * m[4];
    const R A3 = m[5] * m[6] - m[3] * m[8], A4 = m[0] * m[8] - m[2] * m[6], A5 = m[2] * m[3] - m[0] * m[5];
    const R A6 = m[3] * m[7] - m[4] * m[6], A7 = m[1] * m[6] - m[0] * m[7], A8 = m[0] * m[4] - m[1] * m[3];
    const R A0a = fabsf(m[4] * m[8]) + fabsf(m[5] * m[7]), A1a = fabsf(m[2] * m[7]) + fabsf(m[1] * m[8]), A2a = fabsf(m[1] * m[5]) + fabsf(m[2] * m[4]);
    const R A3a = fabsf(m[5] * m[6]) + fabsf(m[3] * m[8]), A4a = fabsf(m[0] * m[8]) + fabsf(m[2] * m[6]), A5a = fabsf(m[2] * m[3]) + fabsf(m[0] * m[5]);
    const R A6a = fabsf(m[3] * m[7]) + fabsf(m[4] * m[6]), A7a = fabsf(m[1] * m[6]) + fabsf(m[0] * m[7]), A8a = fabsf(m[0] * m[4]) + fabsf(m[1] * m[3]);
    // pixel -> normalised source coordinate (inverse of km_unnormalize): gn = ax * x + bx
    const R ax = g.align ? (g.W > 1 ? 2.0f / (R)(g.W - 1) : 0.0f) : 2.0f / (R)g.W;
    const R bx = g.align ? -1.0f : 1.0f / (R)g.W - 1.0f;
    const R ay = g.align ? (g.H > 1 ? 2.0f / (R)(g.H - 1) : 0.0f) : 2.0f / (R)g.H;
    const R by = g.align ? -1.0f : 1.0f / (R)g.H - 1.0f;
    const R P0 = A0 * ax, P1 = A1 * ay, P2 = A0 * bx + A1 * by + A2;
    const R P3 = A3 * ax, P4 = A4 * ay, P5 = A3 * bx + A4 * by + A5;
    const R P6 = A6 * ax, P7 = A7 * ay, P8 = A6 * bx + A7 * by + A8;
    const R abx = fabsf(bx), aby = fabsf(by);
    const R P0a = A0a * ax, P1a = A1a * ay, P2a = A0a * abx + A1a * aby + A2a;
    const R P3a = A3a * ax, P4a = A4a * ay, P5a = A3a * abx + A4a * aby + A5a;
    const R P6a = A6a * ax, P7a = A7a * ay, P8a = A6a * abx + A7a * aby + A8a;
    R sj, oj, si, oi;
    kmt_index_affine<CM>(g, g.w, g.lin_lo_x, g.lin_step_x, sj, oj);
    kmt_index_affine<CM>(g, g.h, g.lin_lo_y, g.lin_step_y, si, oi);
    G[0] = sj * P0 + oj * P6; G[1] = sj * P1 + oj * P7; G[2] = sj * P2 + oj * P8;
    G[3] = si * P3 + oi * P6; G[4] = si * P4 + oi * P7; G[5] = si * P5 + oi * P8;
    G[6] = P6; G[7] = P7; G[8] = P8;
    const R asj = fabsf(sj), aoj = fabsf(oj), asi = fabsf(si), aoi = fabsf(oi);
    Ga[0] = asj * P0a + aoj * P6a; Ga[1] = asj * P1a + aoj * P7a; Ga[2] = asj * P2a + aoj * P8a;
    Ga[3] = asi * P3a + aoi * P6a; Ga[4] = asi * P4a + aoi * P7a; Ga[5] = asi * P5a + aoi * P8a;
    Ga[6] = P6a; Ga[7] = P7a; Ga[8] = P8a;

    // corners of the tile grown by the bilinear footprint: floor(x) in [X0-1, X1-1]  <=>  x in [X0-1, X1)
    const R xs[2] = {(R)(X0 - 1), (R)X1}, ys[2] = {(R)(Y0 - 1), (R)Y1};
    R jmin = 3.0e38f, jmax = -3.0e38f, imin = 3.0e38f, imax = -3.0e38f, dmin = 3.0e38f, dmax = -3.0e38f, nmax = 0.f;
    R njx = 0.f, njy = 0.f, nix = 0.f, niy = 0.f;
    R egj = 0.f, egi = 0.f;     // rounding of the inverse map at the corners, in output pixels (before the factor gamma)
    R drel = 0.f;               // max |D| rounding relative to |D|
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
            const R Jn = G[0] * xs[cx] + G[1] * ys[cy] + G[2];
            const R In = G[3] * xs[cx] + G[4] * ys[cy] + G[5];
            const R D = G[6] * xs[cx] + G[7] * ys[cy] + G[8];
            dmin = fminf(dmin, D); dmax = fmaxf(dmax, D);
            nmax = fmaxf(nmax, fmaxf(fabsf(Jn), fabsf(In)));
            const R fj = Jn / D, fi = In / D;
            jmin = fminf(jmin, fj); jmax = fmaxf(jmax, fj);
            imin = fminf(imin, fi); imax = fmaxf(imax, fi);
            njx = fmaxf(njx, fabsf(G[0] * D - Jn * G[6])); njy = fmaxf(njy, fabsf(G[1] * D - Jn * G[7]));
            nix = fmaxf(nix, fabsf(G[3] * D - In * G[6])); niy = fmaxf(niy, fabsf(G[4] * D - In * G[7]));
            if (KMT_TIGHT_BOX) {
                const R axs = fabsf(xs[cx]), ays = fabsf(ys[cy]);
                const R Ja = Ga[0] * axs + Ga[1] * ays + Ga[2], Ia = Ga[3] * axs + Ga[4] * ays + Ga[5], Da = Ga[6] * axs + Ga[7] * ays + Ga[8];
                const R invd = 1.0f / fabsf(D);
                egj = fmaxf(egj, (Ja + fabsf(fj) * Da) * invd);
                egi = fmaxf(egi, (Ia + fabsf(fi) * Da) * invd);
                drel = fmaxf(drel, Da * invd);
            }
        }
    const bool same_sign = (dmin > 0.f) || (dmax < 0.f);
    const R dabs_min = fminf(fabsf(dmin), fabsf(dmax)), dabs_max = fmaxf(fabsf(dmin), fabsf(dmax));
    const bool ok = same_sign && (dabs_min > 1e-6f * fmaxf(nmax, dabs_max)) && (jmin == jmin) && (jmax == jmax) && (imin == imin) && (imax == imax);
    if (!ok) return o;  // tile crossed by the vanishing line: visit the whole output (correct, slower)

    const R big = 1.0e9f;
    jmin = fmaxf(jmin, -big); jmax = fminf(jmax, big); imin = fmaxf(imin, -big); imax = fminf(imax, big);
    const R inv_d2 = 1.0f / (dabs_min * dabs_min);
    const R jac_j = (njx + njy) * inv_d2, jac_i = (nix + niy) * inv_d2;  // |dj/dx| + |dj/dy|, |di/dx| + |di/dy| over the tile
    R mj = 1.0f, mi = 1.0f;     // margins in output pixels; with floor / ceil below this is the first version's box
    bool flat_box = true;
    if (KMT_TIGHT_BOX) {
        const R gamma = 64.0f * 5.9604645e-8f;  // ~10 roundings per quantity, x6 safety
        // forward rounding: the position of output pixel (j, i) is N / Dn with |u|, |v| <= U, V; Dn is affine in
        // (u, v), so its smallest magnitude over the (flat-margin) box is attained at a corner
        const int pj0 = max(0, (int)floorf(jmin) - 1), pj1 = min(g.w - 1, (int)ceilf(jmax) + 1);
        const int pi0 = max(0, (int)floorf(imin) - 1), pi1 = min(g.h - 1, (int)ceilf(imax) + 1);
        if (pj0 <= pj1 && pi0 <= pi1) {
            const R u0 = km_base_x<R, CM>(g, pj0), u1 = km_base_x<R, CM>(g, pj1), v0 = km_base_y<R, CM>(g, pi0), v1 = km_base_y<R, CM>(g, pi1);
            const R U = fmaxf(fabsf(u0), fabsf(u1)), V = fmaxf(fabsf(v0), fabsf(v1));
            R dn_min = 1.0f, dn_sgn_ok = 1.0f;
            if (CM != KM_COORD_AFFINE) {
                const R d00 = (m[6] * u0 + m[7] * v0) + m[8], d01 = (m[6] * u1 + m[7] * v0) + m[8];
                const R d10 = (m[6] * u0 + m[7] * v1) + m[8], d11 = (m[6] * u1 + m[7] * v1) + m[8];
                const R lo = fminf(fminf(d00, d01), fminf(d10, d11)), hi = fmaxf(fmaxf(d00, d01), fmaxf(d10, d11));
                dn_sgn_ok = ((lo > 0.f) || (hi < 0.f)) ? 1.0f : 0.0f;
                dn_min = fminf(fabsf(lo), fabsf(hi));
            }
            const R Sx = fabsf(m[0]) * U + fabsf(m[1]) * V + fabsf(m[2]), Sy = fabsf(m[3]) * U + fabsf(m[4]) * V + fabsf(m[5]);
            const R Sd = (CM == KM_COORD_AFFINE) ? 0.0f : fabsf(m[6]) * U + fabsf(m[7]) * V + fabsf(m[8]);
            const R gmax = 2.0f;  // |normalised coordinate| of a pixel that touches the image is < 1 + 2/size
            const R dgx = gamma * (Sx + gmax * Sd) / dn_min, dgy = gamma * (Sy + gmax * Sd) / dn_min;
            const R dx = 0.5f * (R)g.W * dgx + gamma * (R)g.W, dy = 0.5f * (R)g.H * dgy + gamma * (R)g.H;  // source pixels
            const R dfwd = fmaxf(dx, dy);
            const R tj = 0.125f + 2.0f * (gamma * egj + jac_j * dfwd), ti = 0.125f + 2.0f * (gamma * egi + jac_i * dfwd);
            // the first-order bounds need D and Dn well away from zero relative to their own rounding
            const bool trust = (dn_sgn_ok > 0.5f) && (gamma * drel < 0.125f) && (gamma * Sd < 0.125f * dn_min) && (tj == tj) && (ti == ti) &&
                               (tj < 1.0f) && (ti < 1.0f);
            if (trust) { mj = tj; mi = ti; flat_box = false; }
        }
    }
    if (flat_box) {
        o.j0 = max(0, (int)floorf(jmin) - 1); o.j1 = min(g.w - 1, (int)ceilf(jmax) + 1);
        o.i0 = max(0, (int)floorf(imin) - 1); o.i1 = min(g.h - 1, (int)ceilf(imax) + 1);
    } else {
        o.j0 = max(0, (int)floorf(jmin - mj)); o.j1 = min(g.w - 1, (int)ceilf(jmax + mj));
        o.i0 = max(0, (int)floorf(imin - mi)); o.i1 = min(g.h - 1, (int)ceilf(imax + mi));
    }
    // output pixels per source pixel: the 2x2 footprint box maps to at most (2 ex + 1)(2 ey + 1) lattice points
    const R ex = jac_j + 0.1f, ey = jac_i + 0.1f;
    o.mult = fminf((2.f * ex + 1.f) * (2.f * ey + 1.f), 1.0e6f);
    o.fixed_ok = o.mult <= 256.f;  // beyond ~7x magnification the head-room would eat the mantissa: float path
    return o;
}

#define KMT_TAB 256  // capacity of the per-band base-coordinate tables

// block-uniform values computed with VALU float math live in VGPRs unless moved to SGPRs explicitly
__device__ __forceinline__ float kmt_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ int kmt_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// fixed-point quantisation of one contribution: floor(v + 0.5) in ONE instruction (v_cvt_rpi_i32_f32) instead of
// v_rndne_f32 + v_cvt_i32_f32.  Ties round up instead of to even; both are exact integers of the same
// magnitude bound, and the result stays independent of the order of accumulation.
#ifndef KMT_CVT_RPI
#define KMT_CVT_RPI 1
#endif
__device__ __forceinline__ int kmt_quant(float v) {
#if KMT_CVT_RPI
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
#else
    return __float2int_rn(v);
#endif
}

// one output pixel of pass B.  Branch-free up to the (exec-masked) atomics: every load is unconditional
// (clamped addresses) so that the loads of two pixels processed back to back can be in flight together.
// per-thread walk over the box in steps of 256 elements: (qi, qj) of element e + 256 from those of e
__device__ __forceinline__ void kmt_advance(int& qi, int& qj, int di, int dj, int bw) {
    qj += dj;
    qi += di;
    const bool carry = qj >= bw;
    qj = carry ? qj - bw : qj;
    qi = carry ? qi + 1 : qi;
}

template <typename T, int CM, bool WANT_GM>
__device__ __forceinline__ void kmt_scatter_q(const KmWarpTiledArgs<T>& a, const float (&m)[9], int qi, int qj, bool valid, int j0, int ib,
                                             bool tab_x, const float* s_u, const float* s_v, int* s_acc, bool finite, float scale, int cbase,
                                             int cc, const T* src_b, const T* const (&gout_c)[KMT_CC], size_t src_plane, int X0, int TWc, int Y0,
                                             int THc, float (&gm)[9], uint32_t& seen_bits) {
    typedef float R;
    const KmWarpGeom<R>& g = a.g;
    const int jj = j0 + qj, ii = ib + qi;
    const uint32_t off = (uint32_t)ii * (uint32_t)g.w + (uint32_t)jj;  // host guarantees h * w < 2^31
    R go[KMT_CC];
#pragma unroll
    for (int c = 0; c < KMT_CC; ++c) go[c] = (R)km_ld(gout_c[c] + off);  // channels >= cc alias channel cc-1 (never used)
    // the fixed-point scale was chosen for |grad_out| <= bound: remember the largest magnitude seen, as an integer
    // (sign cleared, IEEE bit patterns order like unsigned integers and NaN / inf sort above every finite value)
    {
        uint32_t mb = __float_as_uint(go[0]) & 0x7fffffffu;
#pragma unroll
        for (int c = 1; c < KMT_CC; ++c) mb = max(mb, __float_as_uint(go[c]) & 0x7fffffffu);
        seen_bits = max(seen_bits, mb);
    }
    KmCoord<R> cd;
    km_gen_coord<R, CM>(m, tab_x ? s_u[qj] : km_base_x<R, CM>(g, jj), s_v[qi], cd);
    R mx, my;
    const R x = km_unnormalize(cd.gx, g.W, g.align, mx);
    const R y = km_unnormalize(cd.gy, g.H, g.align, my);
    // weights: the forward's own expressions ((x0 + 1) - x, x - x0), so grad_src = W^T grad_out for the very W it applied
    const R xf = km_floor(x), yf = km_floor(y);
    const R wx0 = x - xf, wx1 = (xf + 1) - x, wy0 = y - yf, wy1 = (yf + 1) - y;
    // tile-relative tap position.  A tap inside the tile is inside the image, so the in-tile test is the whole
    // predicate; clamping in float first sends NaN / huge coordinates (and the padding lanes) outside the tile
    const int ux = (int)fmaxf(fminf(xf, (R)g.W), (R)-2) - X0;
    const int uy = valid ? (int)fmaxf(fminf(yf, (R)g.H), (R)-2) - Y0 : (1 << 20);
    const bool in_x0 = (uint32_t)ux < (uint32_t)TWc, in_x1 = (uint32_t)(ux + 1) < (uint32_t)TWc;
    const bool in_y0 = (uint32_t)uy < (uint32_t)THc, in_y1 = (uint32_t)(uy + 1) < (uint32_t)THc;
    const bool t00 = in_x0 && in_y0, t01 = in_x1 && in_y0, t10 = in_x0 && in_y1, t11 = in_x1 && in_y1;
    const int l00 = uy * KMT_TW + ux;
    struct { R w00, w01, w10, w11; } t;
    t.w00 = wx1 * wy1; t.w01 = wx0 * wy1; t.w10 = wx1 * wy0; t.w11 = wx0 * wy0;
    // tap-outer order: one exec-mask region per tap (4 per pixel) instead of one per atomic (4 * C)
    if (finite) {
        const R w00 = t.w00 * scale, w01 = t.w01 * scale, w10 = t.w10 * scale, w11 = t.w11 * scale;
        int* accp = s_acc + l00;
        if (t00) {
#pragma unroll
            for (int c = 0; c < KMT_CC; ++c)
                if (c < cc) atomicAdd(accp + c * (KMT_TH * KMT_TW), kmt_quant(w00 * go[c]));
        }
        if (t01) {
#pragma unroll
            for (int c = 0; c < KMT_CC; ++c)
                if (c < cc) atomicAdd(accp + c * (KMT_TH * KMT_TW) + 1, kmt_quant(w01 * go[c]));
        }
        if (t10) {
#pragma unroll
            for (int c = 0; c < KMT_CC; ++c)
                if (c < cc) atomicAdd(accp + c * (KMT_TH * KMT_TW) + KMT_TW, kmt_quant(w10 * go[c]));
        }
        if (t11) {
#pragma unroll
            for (int c = 0; c < KMT_CC; ++c)
                if (c < cc) atomicAdd(accp + c * (KMT_TH * KMT_TW) + KMT_TW + 1, kmt_quant(w11 * go[c]));
        }
    } else {
        // inf / NaN in grad_out, vanishing-line tiles, extreme magnification: float LDS atomics
        float* accp = (float*)s_acc + l00;
#pragma unroll
        for (int c = 0; c < KMT_CC; ++c) {
            if (c < cc) {
                if (t00) atomicAdd(accp + c * (KMT_TH * KMT_TW), t.w00 * go[c]);
                if (t01) atomicAdd(accp + c * (KMT_TH * KMT_TW) + 1, t.w01 * go[c]);
                if (t10) atomicAdd(accp + c * (KMT_TH * KMT_TW) + KMT_TW, t.w10 * go[c]);
                if (t11) atomicAdd(accp + c * (KMT_TH * KMT_TW) + KMT_TW + 1, t.w11 * go[c]);
            }
        }
    }
    if (WANT_GM) {
        // the tile holding the clamped north-west tap owns q's matrix gradient
        const bool live = valid && (x >= (R)-1) && (x < (R)g.W) && (y >= (R)-1) && (y < (R)g.H);  // has an in-image tap
        KmBilin<R> tb;
        km_bilinear_setup(x, y, g.W, g.H, tb);
        const int x0 = (int)fmaxf(fminf(xf, (R)g.W), (R)-1), y0 = (int)fmaxf(fminf(yf, (R)g.H), (R)-1);
        const int ox = min(max(x0, 0), g.W - 1) - X0, oy = min(max(y0, 0), g.H - 1) - Y0;
        const bool own = live && ((uint32_t)ox < (uint32_t)TWc) && ((uint32_t)oy < (uint32_t)THc);
        R gix = 0, giy = 0;
        if (own) {  // measured: skipping the tap loads of non-owned pixels beats batching them (1.56 -> 1.50 ms)
            // d/dx = (ne - nw)(y1 - y) + (se - sw)(y - y0) ; d/dy = (sw - nw)(x1 - x) + (se - ne)(x - x0)
            if (__all(tb.b00 && tb.b01 && tb.b10 && tb.b11)) {
                // every owner lane of the wave samples inside the image: (x0, x0 + 1) come with one load per row
#pragma unroll
                for (int c = 0; c < KMT_CC; ++c) {
                    if (c < cc) {
                        const T* img = src_b + (size_t)(cbase + c) * src_plane;
                        R s00, s01, s10, s11;
                        km_ld2(img + tb.i00, s00, s01);
                        km_ld2(img + tb.i10, s10, s11);
                        if (g.pad == KM_PAD_FILL) {  // same rounding sequence as the oracle: (v - fill) first
                            const R f = a.fill[cbase + c];
                            s00 -= f; s01 -= f; s10 -= f; s11 -= f;
                        }
                        gix += go[c] * ((s01 - s00) * tb.wy1 + (s11 - s10) * tb.wy0);
                        giy += go[c] * ((s10 - s00) * tb.wx1 + (s11 - s01) * tb.wx0);
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < KMT_CC; ++c) {
                    if (c < cc) {
                        const T* img = src_b + (size_t)(cbase + c) * src_plane;
                        const R f = (g.pad == KM_PAD_FILL) ? a.fill[cbase + c] : (R)0;
                        // unconditional loads (clamped indices); out-of-bounds taps do not exist in the reference's sum
                        const R v00 = (R)km_ld(img + tb.i00), v01 = (R)km_ld(img + tb.i01), v10 = (R)km_ld(img + tb.i10), v11 = (R)km_ld(img + tb.i11);
                        const R s00 = tb.b00 ? v00 - f : (R)0, s01 = tb.b01 ? v01 - f : (R)0;
                        const R s10 = tb.b10 ? v10 - f : (R)0, s11 = tb.b11 ? v11 - f : (R)0;
                        gix += go[c] * ((s01 - s00) * tb.wy1 + (s11 - s10) * tb.wy0);
                        giy += go[c] * ((s10 - s00) * tb.wx1 + (s11 - s01) * tb.wx0);
                    }
                }
            }
        }
        if (own) km_accumulate_gm<CM>(gm, cd, gix * mx, giy * my);
    }
}

// Tile-owner SCATTER with fixed-point LDS accumulators.
//
// gfx950 measurements that shape this kernel (scratch micro-benchmark, 2048 blocks x 256 threads):
//   ds_add_f32 / ds_add_rtn_f32    193 cycles per wave-instruction per CU   (float LDS atomics are ~40x slower
//   ds_add_u32 / ds_add_rtn_u32      5 cycles per wave-instruction per CU    than integer ones)
// so the per-tap contributions w * grad_out are accumulated as int32 fixed point: the block first finds
// M = max |grad_out| over the output pixels it will visit, picks scale = 2^k with
// k = 30 - hb - ceil(log2 M) (hb = head-room bits for the number of taps that can land on one source pixel,
// from the Jacobian bound), and adds rint(w * g * scale) with ds_add_u32.  Per-term error <= 2^-(k+1),
// i.e. <= M * 2^-(27-hb): the same order as one fp32 ulp of M.  Integer addition is associative, so the
// result is independent of the order in which waves run: bit-reproducible, unlike float atomics.
// measured on MI355X (256x3x512^2), speculative-scale version: min-waves 4 -> 1.32 ms, 5 -> 1.38 ms (14 spills), 6 -> 2.00 ms
#ifndef KMT_MIN_WAVES
#define KMT_MIN_WAVES 4
#endif
#ifndef KMT_UNROLL
#define KMT_UNROLL 2
#endif
template <typename T, int CM, bool WANT_GM>
__global__ __launch_bounds__(256, KMT_MIN_WAVES) void km_warp_bwd_tiled_kernel(const KmWarpTiledArgs<T> a) {
    typedef float R;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ double red[4][9];
    __shared__ float red_max[4];
    __shared__ int s_box[8];

    const KmWarpGeom<R>& g = a.g;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int X0 = (int)tx * KMT_TW, Y0 = (int)ty * KMT_TH;
    const int X1 = min(X0 + KMT_TW, g.W), Y1 = min(Y0 + KMT_TH, g.H);  // tile = [X0,X1) x [Y0,Y1)

    R m[9];
    {
        const R* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = mp[k];
    }

    // ---- box of output pixels that can touch the tile: computed by wave 0 only (it is ~300 block-uniform
    //      VALU instructions, ~15 % of a block's work if all four waves repeat it), broadcast through LDS
    if (wave == 0) {
        const KmtBox bx = kmt_tile_box<CM>(g, m, X0, X1, Y0, Y1);
        if (lane == 0) {
            s_box[0] = bx.j0; s_box[1] = bx.j1; s_box[2] = bx.i0; s_box[3] = bx.i1;
            s_box[4] = (int)ceilf(log2f(fmaxf(bx.mult, 1.f))) + 1;  // head-room bits
            s_box[5] = bx.fixed_ok ? 1 : 0;
        }
    }
    __syncthreads();
    const int j0 = kmt_uniform(s_box[0]), j1 = kmt_uniform(s_box[1]), i0 = kmt_uniform(s_box[2]), i1 = kmt_uniform(s_box[3]);
    const int hb = kmt_uniform(s_box[4]);
    const bool fixed_ok = kmt_uniform(s_box[5]) != 0;  // fixed-point accumulation is accurate enough (bounded multiplicity)
    const int bw = j1 - j0 + 1, bh = i1 - i0 + 1;
    const bool empty = (bw <= 0 || bh <= 0);
    const float inv_bw = kmt_uniform(bw > 0 ? 1.0f / (float)bw : 0.f);

    const size_t src_plane = (size_t)g.H * g.W, dst_plane = (size_t)g.h * g.w;
    const T* src_b = a.src + (size_t)b * g.C * src_plane;
    const T* gout_b = a.gout + (size_t)b * g.C * dst_plane;
    R* gsrc_b = a.gsrc + (size_t)b * g.C * src_plane;

    // LDS carve: base-coordinate tables, then the int32 accumulators [cc][TH][TW]
    float* s_u = (float*)smem_raw;  // [KMT_TAB]
    float* s_v = s_u + KMT_TAB;      // [KMT_TAB]
    int* s_acc = (int*)(s_v + KMT_TAB);
    const bool tab_x = bw <= KMT_TAB;

    for (int cbase = 0; cbase < g.C; cbase += KMT_CC) {
        const int cc = min(KMT_CC, g.C - cbase);
        for (int e = tid; e < cc * KMT_TH * KMT_TW; e += 256) s_acc[e] = 0;
        if (tab_x && tid < bw) s_u[tid] = km_base_x<R, CM>(g, j0 + tid);

        // The scale needs an upper bound M on |grad_out| over the box.  Reading the whole box twice costs ~0.3 ms
        // at 256x3x512^2, so attempt 0 SPECULATES: M = 8 x (max over a 256-pixel sample of the box); pass B
        // checks every value it loads against M, and only a tile that sees a larger one (or a NaN/inf) is redone
        // with the exact maximum (attempt 1).  The guard factor costs 3 bits of the fixed-point resolution.
        R scale = 1.f, inv_scale = 1.f;
        bool finite = false;
        for (int attempt = (fixed_ok ? 0 : 1); attempt < 2; ++attempt) {
            R vmax = 0.f;
            bool bad = false;
            if (!empty) {
                const int nq = bw * bh;
                if (attempt == 0) {
                    const int e = (int)(((long long)tid * nq) >> 8);  // 256 pixels spread over the box
                    int qi = (int)(((float)e + 0.5f) * inv_bw);
                    int qj = e - qi * bw;
                    if (qj < 0) { qi -= 1; qj += bw; }
                    if (qj >= bw) { qi += 1; qj -= bw; }
                    const T* go_px = gout_b + (size_t)(i0 + qi) * g.w + (j0 + qj);
#pragma unroll
                    for (int c = 0; c < KMT_CC; ++c)
                        if (c < cc) vmax = fmaxf(vmax, km_fabs((R)km_ld(go_px + (size_t)(cbase + c) * dst_plane)));
                    vmax = vmax * 8.0f;
                    bad = !(vmax <= 3.0e38f);
                } else {
                    for (int base = 0; base < nq; base += 4 * 256) {
                        R vv[4][KMT_CC];
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            const int e = min(base + s4 * 256 + tid, nq - 1);  // clamped: duplicates do not change a max
                            int qi = (int)(((float)e + 0.5f) * inv_bw);
                            int qj = e - qi * bw;
                            if (qj < 0) { qi -= 1; qj += bw; }
                            if (qj >= bw) { qi += 1; qj -= bw; }
                            const T* go_px = gout_b + (size_t)(i0 + qi) * g.w + (j0 + qj);
#pragma unroll
                            for (int c = 0; c < KMT_CC; ++c) vv[s4][c] = (c < cc) ? km_fabs((R)km_ld(go_px + (size_t)(cbase + c) * dst_plane)) : (R)0;
                        }
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                            for (int c = 0; c < KMT_CC; ++c) {
                                bad = bad || !(vv[s4][c] <= 3.0e38f);
                                vmax = fmaxf(vmax, vv[s4][c]);
                            }
                    }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
            const unsigned long long badmask = __ballot(bad);
            __syncthreads();  // red_max may still be read by a previous attempt / channel chunk
            if (lane == 0) red_max[wave] = badmask ? __int_as_float(0x7f800000) : vmax;
            __syncthreads();
            const R M = fmaxf(fmaxf(red_max[0], red_max[1]), fmaxf(red_max[2], red_max[3]));
            // scale = 2^k with |w * g * scale| * (taps per pixel) < 2^30
            finite = (M <= 3.0e38f) && fixed_ok;  // else: IEEE float accumulation (slow ds_add_f32)
            int kexp = 0;
            if (finite && M > 0.f) {
                int ex2;
                (void)frexpf(M, &ex2);  // M = f * 2^ex2, f in [0.5, 1)  =>  M < 2^ex2
                kexp = 30 - hb - ex2;
                kexp = max(-126, min(126, kexp));
            }
            scale = kmt_uniform(ldexpf(1.0f, kexp));
            inv_scale = kmt_uniform(ldexpf(1.0f, -kexp));
            const R bound = finite ? M : __int_as_float(0x7f800000);  // the float path accepts anything

            // ---- pass B: scatter ----
            R gm_try[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) gm_try[k] = 0;
            uint32_t seen_bits = 0;  // largest |grad_out| bit pattern this thread has loaded
            if (!empty) {
                const T* gout_c[KMT_CC];
#pragma unroll
                for (int c = 0; c < KMT_CC; ++c) gout_c[c] = gout_b + (size_t)(cbase + min(c, cc - 1)) * dst_plane;
                const int di = kmt_uniform(256 / bw), dj = kmt_uniform(256 % bw);  // element e + 256 is di rows and dj columns on
                const int TWc = X1 - X0, THc = Y1 - Y0;
                for (int ib = i0; ib <= i1; ib += KMT_TAB) {
                    const int ie = min(i1, ib + KMT_TAB - 1);
                    __syncthreads();
                    if (tid <= ie - ib) s_v[tid] = km_base_y<R, CM>(g, ib + tid);
                    __syncthreads();
                    const int nq = bw * (ie - ib + 1);
                    int qi = tid / bw, qj = tid - qi * bw;  // element e = base + tid
                    int base = 0;
                    for (; base + KMT_UNROLL * 256 <= nq; base += KMT_UNROLL * 256) {
#pragma unroll
                        for (int s4 = 0; s4 < KMT_UNROLL; ++s4) {
                            kmt_scatter_q<T, CM, WANT_GM>(a, m, qi, qj, true, j0, ib, tab_x, s_u, s_v, s_acc, finite, scale, cbase, cc, src_b,
                                                          gout_c, src_plane, X0, TWc, Y0, THc, gm_try, seen_bits);
                            kmt_advance(qi, qj, di, dj, bw);
                        }
                    }
                    for (; base < nq; base += 256) {
                        const bool valid = base + tid < nq;
                        kmt_scatter_q<T, CM, WANT_GM>(a, m, valid ? qi : 0, valid ? qj : 0, valid, j0, ib, tab_x, s_u, s_v, s_acc, finite, scale,
                                                      cbase, cc, src_b, gout_c, src_plane, X0, TWc, Y0, THc, gm_try, seen_bits);
                        kmt_advance(qi, qj, di, dj, bw);
                    }
                }
            }
            // the fixed-point scale was chosen for |grad_out| <= bound; anything larger (or NaN) voids the attempt
            const bool exceeded = seen_bits > __float_as_uint(bound);
            const int redo = __syncthreads_or((int)exceeded);
            if (attempt == 0 && redo) {
                for (int e = tid; e < cc * KMT_TH * KMT_TW; e += 256) s_acc[e] = 0;  // discard the speculative attempt
                continue;  // the barrier at the top of attempt 1 orders these stores before the next atomics
            }
            if (WANT_GM) {
                // accepted attempt: fold this chunk's matrix gradient into the global fp64 accumulators
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const double sgm = km_wave_sum((double)gm_try[k]);
                    if (lane == 0) red[wave][k] = sgm;
                }
                __syncthreads();
                if (tid < 9) {
                    const double sgm = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
                    if (sgm != 0.0) km_atomic_add(a.gmat + (size_t)(g.B_M == 1 ? 0 : b) * 9 + tid, sgm);
                }
            }
            break;
        }

        // ---- convert and write the tile: rows of 64 floats, fully coalesced ----
        for (int e = tid; e < cc * KMT_TH * KMT_TW; e += 256) {
            const int c = e / (KMT_TH * KMT_TW), r = (e / KMT_TW) % KMT_TH, col = e % KMT_TW;
            const int yy = Y0 + r, xx = X0 + col;
            if (yy < g.H && xx < g.W) {
                const R val = finite ? (R)s_acc[e] * inv_scale : ((const float*)s_acc)[e];
                gsrc_b[(size_t)(cbase + c) * src_plane + (size_t)yy * g.W + xx] = val;
            }
        }
        __syncthreads();
    }

}

template <typename T, int CM>
static int kmt_launch(const KmWarpTiledArgs<T>& a, hipStream_t s) {
    if (a.gmat)
        hipLaunchKernelGGL((km_warp_bwd_tiled_kernel<T, CM, true>), dim3(a.nblocks), dim3(256), (size_t)a.lds_bytes, s, a);
    else
        hipLaunchKernelGGL((km_warp_bwd_tiled_kernel<T, CM, false>), dim3(a.nblocks), dim3(256), (size_t)a.lds_bytes, s, a);
    return km_check_launch("km_warp2d_bwd(tiled)");
}

template <typename T>
static int kmt_run(const void* gout, const void* src, const void* mat, void* gsrc, double* gmat, int B, int C, int H, int W, int h,
                   int w, int B_M, int coord_mode, int norm_coords, int pad, int align, const void* fill, hipStream_t s) {
    KmWarpTiledArgs<T> a;
    a.src = (const T*)src; a.gout = (const T*)gout; a.mat = (const float*)mat; a.gsrc = (float*)gsrc; a.gmat = gmat;
    a.fill = (const float*)fill;
    {
        a.lds_bytes = KMT_LDS_BYTES;
    }
    KmWarpGeom<float>& g = a.g;
    g.B = B; g.C = C; g.H = H; g.W = W; g.h = h; g.w = w; g.B_M = B_M;
    g.coord_mode = coord_mode; g.norm_coords = norm_coords; g.interp = KM_INTERP_BILINEAR; g.pad = pad; g.align = align;
    if (align) {
        g.lin_lo_x = -1.0f; g.lin_hi_x = 1.0f; g.lin_lo_y = -1.0f; g.lin_hi_y = 1.0f;
    } else {
        g.lin_lo_x = (float)(-1.0 + 1.0 / w); g.lin_hi_x = (float)(1.0 - 1.0 / w);
        g.lin_lo_y = (float)(-1.0 + 1.0 / h); g.lin_hi_y = (float)(1.0 - 1.0 / h);
    }
    g.lin_step_x = w > 1 ? (g.lin_hi_x - g.lin_lo_x) / (float)(w - 1) : 0.0f;
    g.lin_step_y = h > 1 ? (g.lin_hi_y - g.lin_lo_y) / (float)(h - 1) : 0.0f;
    a.tiles_x = (uint32_t)((W + KMT_TW - 1) / KMT_TW);
    a.tiles_y = (uint32_t)((H + KMT_TH - 1) / KMT_TH);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B;
    KM_REQUIRE(nb < (1ull << 31), "km_warp2d_bwd: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    switch (coord_mode) {
        case KM_COORD_PERSPECTIVE: return kmt_launch<T, KM_COORD_PERSPECTIVE>(a, s);
        case KM_COORD_AFFINE: return kmt_launch<T, KM_COORD_AFFINE>(a, s);
        default: return kmt_launch<T, KM_COORD_HOMOGRAPHY>(a, s);
    }
}

// 1 if the tiled backward applies (bilinear, zeros/fill padding, fp32 compute, grad_src wanted)
int km_warp_bwd_tiled_supported(int interp, int pad, int dtype, const void* gsrc) {
    static int disabled = -1;
    if (disabled < 0) {
        const char* e = getenv("KM_WARP_BWD_ALGO");  // "generic" forces the atomic scatter kernel
        disabled = (e && e[0] == 'g') ? 1 : 0;
    }
    if (disabled) return 0;
    return (interp == KM_INTERP_BILINEAR && (pad == KM_PAD_ZEROS || pad == KM_PAD_FILL) && dtype != KM_F64 && gsrc != nullptr) ? 1 : 0;
}

int km_warp_bwd_tiled_run(const void* gout, const void* src, const void* mat, void* gsrc, double* gmat, int B, int C, int H, int W,
                          int h, int w, int B_M, int coord_mode, int norm_coords, int pad, int align, const void* fill, int dtype,
                          hipStream_t s) {
    switch (dtype) {
        case KM_F32: return kmt_run<float>(gout, src, mat, gsrc, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, s);
        case KM_BF16: return kmt_run<km_bf16>(gout, src, mat, gsrc, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, s);
        default: return kmt_run<km_f16>(gout, src, mat, gsrc, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, s);
    }
}
