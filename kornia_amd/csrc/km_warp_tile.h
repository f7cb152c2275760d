// kornia_amd - pieces shared by the tile-owner backward kernels of the bilinear warps (km_warp_bwd_tiled.hip: gradient with
// respect to the image; km_warp_bwd_fused.hip: both gradients from one read of grad_out): the box of output pixels that can touch a
// tile of the source, the fixed-point quantisation, the walk over a box and the per-pixel position record.
#pragma once

#include "km_lean.h"

#ifndef KMT_TW
#define KMT_TW 64           // tile width  (the flush maps lane -> column: keep 64)
#endif
#ifndef KMT_TH
#define KMT_TH 64           // tile height
#endif
#define KMT_BAND_W 128      // output columns per band = capacity of the column table (float4 entries)
#define KMT_TAB 128         // output rows per band = capacity of the row table (float4 entries)
#define KMT_PLANE (KMT_TH * KMT_TW)

// [host-testable begin: tile_box]  (tests/test_tile_box_spec.py compiles this span for the host with g++)
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
    bool ok;        // the box is the tile's own (false: the whole output - vanishing line, degenerate map)
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
    o.ok = false;
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
    bool num = true;            // every corner maps to a number (fminf / fmaxf DROP a NaN operand: the minima above would keep their initial values)
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
            num = num && (D == D) && (fj == fj) && (fi == fi);
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
    // (a NaN anywhere in the matrix - a singular M through the closed-form inverse - makes every corner NaN: without `num` the box came
    // out as (INT_MAX - 1 .. INT_MIN + 1), whose width WRAPS to a small positive number - reads of grad_out gigabytes away from the
    // tensor on the device: found by the round-6 device run of tests/golden/nonfinite_coords.npz)
    const bool ok = num && same_sign && (dabs_min > 1e-6f * fmaxf(nmax, dabs_max)) && (jmin == jmin) && (jmax == jmax) && (imin == imin) && (imax == imax);
    if (!ok) return o;  // tile crossed by the vanishing line, or not a map at all: visit the whole output (correct, slower)

    const R big = 1.0e9f;  // (both sides: the conversions to int below must not saturate, nor their sums wrap)
    jmin = fminf(fmaxf(jmin, -big), big); jmax = fmaxf(fminf(jmax, big), -big); imin = fminf(fmaxf(imin, -big), big); imax = fmaxf(fminf(imax, big), -big);
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
    o.ok = true;
    return o;
}
// [host-testable end: tile_box]

// ---- padding modes that bring every sampling position INTO the image ---------------------------------------------------------------
// border: a position left of the image samples column 0, so the tiles along an edge of the image also own every output pixel that
// maps beyond that edge; reflection: a position is mirrored about the image's edges (period 2 (W - 1), or 2 W without align_corners)
// before it is clamped, so a tile also owns the pixels that map into its mirror images.  The box of a tile is the bounding box of the
// boxes of these "copies" of its rectangle - as far as the output image reaches: the raw positions of all output pixels lie in the hull
// of its four corners' positions (kmt_output_span; the denominator is affine in the base coordinates, so one sign at the corners is one
// sign everywhere).  Pixels in the holes of that bounding box are visited and found to touch nothing.
template <int CM>
__device__ __forceinline__ bool kmt_output_span(const KmWarpGeom<float>& g, const float (&m)[9], float& sx0, float& sx1, float& sy0, float& sy1) {
    const float Wm1 = (float)(g.W - 1), Hm1 = (float)(g.H - 1), hW = (float)g.W / 2, hH = (float)g.H / 2;
    sx0 = 3.0e38f; sx1 = -3.0e38f; sy0 = 3.0e38f; sy1 = -3.0e38f;
    float dmin = 3.0e38f, dmax = -3.0e38f;
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const KmlHalf cu = kml_col_half<CM>(m, km_base_x<float, CM>(g, (k & 1) ? g.w - 1 : 0));
        const KmlHalf rv = kml_row_half<CM>(m, km_base_y<float, CM>(g, (k & 2) ? g.h - 1 : 0));
        KmlPos p;
        kml_position<CM, false>(m, cu, rv, p);
        const float x = g.align ? kml_unnormalize<1>(p.gx, Wm1, hW) : kml_unnormalize<0>(p.gx, Wm1, hW);
        const float y = g.align ? kml_unnormalize<1>(p.gy, Hm1, hH) : kml_unnormalize<0>(p.gy, Hm1, hH);
        const float d = (CM == KM_COORD_AFFINE) ? 1.0f : p.den;
        finite = finite && (x == x) && (y == y) && (fabsf(x) < 1.0e7f) && (fabsf(y) < 1.0e7f) && (d == d) && (CM != KM_COORD_HOMOGRAPHY || p.live);
        sx0 = fminf(sx0, x); sx1 = fmaxf(sx1, x); sy0 = fminf(sy0, y); sy1 = fmaxf(sy1, y);
        dmin = fminf(dmin, d); dmax = fmaxf(dmax, d);
    }
    return finite && ((dmin > 0.f && dmin > 1e-4f * dmax) || (dmax < 0.f && dmax < 1e-4f * dmin));
}
// the intervals [a, b) of RAW positions (as cell indices) along one axis that end up in the cells [c0, c1) of an axis of n cells; lo / hi: the
// cells the output's raw positions span.  Returns the number of intervals (<= 3), or -1 when more would be needed.
__device__ __forceinline__ int kmt_axis_copies(int pad, int align, int c0, int c1, int n, int lo, int hi, int (&a)[3], int (&b)[3]) {
    if (pad == KM_PAD_BORDER) {
        a[0] = (c0 == 0) ? min(c0, lo) : c0;
        b[0] = (c1 == n) ? max(c1, hi + 1) : c1;
        return 1;
    }
    // reflection (ATen reflect_coordinates): align: about 0 and n - 1 (period 2 (n - 1)); otherwise about -0.5 and n - 0.5 (period 2 n);
    // one cell of slack on every interval for the clamp that follows the mirror
    const int P = align ? 2 * (n - 1) : 2 * n;
    if (P <= 0) return -1;
    int cnt = 0;
    const int kmin = (lo - n) / P - 2, kmax = (hi + n) / P + 2;
    for (int k = kmin; k <= kmax; ++k) {
        // the copy itself, then its mirror image: x -> -x (align) / -1 - x
        const int da = c0 + k * P - 1, db = c1 + k * P + 1;
        const int ma = (align ? -(c1 - 1) : -1 - (c1 - 1)) + k * P - 1, mb = (align ? -c0 : -1 - c0) + k * P + 2;
        if (db > lo && da <= hi) { if (cnt == 3) return -1; a[cnt] = da; b[cnt] = db; ++cnt; }
        if (mb > lo && ma <= hi) { if (cnt == 3) return -1; a[cnt] = ma; b[cnt] = mb; ++cnt; }
    }
    if (cnt == 0) { a[0] = c0; b[0] = c1; cnt = 1; }  // (nothing of the output reaches the tile: its own rectangle gives an empty or tiny box)
    return cnt;
}
template <int CM>
__device__ __forceinline__ KmtBox kmt_tile_box_padded(const KmWarpGeom<float>& g, const float (&m)[9], int X0, int X1, int Y0, int Y1) {
    if (g.pad != KM_PAD_BORDER && g.pad != KM_PAD_REFLECTION) return kmt_tile_box<CM>(g, m, X0, X1, Y0, Y1);
    KmtBox whole;
    whole.j0 = 0; whole.j1 = g.w - 1; whole.i0 = 0; whole.i1 = g.h - 1;
    whole.mult = (float)g.w * (float)g.h; whole.fixed_ok = false; whole.ok = false;
    float sx0, sx1, sy0, sy1;
    if (!kmt_output_span<CM>(g, m, sx0, sx1, sy0, sy1)) return whole;
    const int lox = (int)floorf(sx0) - 2, hix = (int)ceilf(sx1) + 2, loy = (int)floorf(sy0) - 2, hiy = (int)ceilf(sy1) + 2;
    int xa[3], xb[3], ya[3], yb[3];
    const int nx = kmt_axis_copies(g.pad, g.align, X0, X1, g.W, lox, hix, xa, xb);
    const int ny = kmt_axis_copies(g.pad, g.align, Y0, Y1, g.H, loy, hiy, ya, yb);
    if (nx < 0 || ny < 0) return whole;
    KmtBox o;
    o.j0 = g.w; o.j1 = -1; o.i0 = g.h; o.i1 = -1; o.mult = 0.f; o.fixed_ok = true; o.ok = true;
    for (int iy = 0; iy < ny; ++iy)
        for (int ix = 0; ix < nx; ++ix) {
            const KmtBox c = kmt_tile_box<CM>(g, m, xa[ix], xb[ix], ya[iy], yb[iy]);
            if (!c.ok) return whole;
            if (c.j0 <= c.j1 && c.i0 <= c.i1) {
                o.j0 = min(o.j0, c.j0); o.j1 = max(o.j1, c.j1); o.i0 = min(o.i0, c.i0); o.i1 = max(o.i1, c.i1);
            }
            o.mult += c.mult;
        }
    if (g.pad == KM_PAD_BORDER) {
        // every output pixel beyond an edge lands on that edge's cells: the multiplicity grows with the extent of the extension
        const KmtBox plain = kmt_tile_box<CM>(g, m, X0, X1, Y0, Y1);
        if (!plain.ok) return whole;
        const int wp = max(plain.j1 - plain.j0 + 1, 0), hp = max(plain.i1 - plain.i0 + 1, 0);
        const int we = max(o.j1 - o.j0 + 1, 0), he = max(o.i1 - o.i0 + 1, 0);
        o.mult = plain.mult * (float)(1 + max(we - wp, 0)) * (float)(1 + max(he - hp, 0));
    }
    o.fixed_ok = o.mult <= 4096.f;  // (the sums on the cells many pixels share are large: the quantisation step stays far below their rounding)
    return o;
}

// block-uniform values computed with VALU float math live in VGPRs unless moved to SGPRs explicitly
__device__ __forceinline__ float kmt_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ int kmt_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// fixed-point quantisation of one contribution: floor(v + 0.5) in ONE instruction (v_cvt_rpi_i32_f32) instead of
// v_rndne_f32 + v_cvt_i32_f32.  Ties round up instead of to even; both are exact integers of the same
// magnitude bound, and the result stays independent of the order of accumulation.  NaN -> 0.
__device__ __forceinline__ int kmt_quant(float v) {
    return KM_CVT_RPI(v);
}

// per-thread walk over a band in steps of KMT_NT elements: (qi, qj) of element e + KMT_NT from those of e
__device__ __forceinline__ void kmt_advance(int& qi, int& qj, int di, int dj, int bw) {
    qj += dj;
    qi += di;
    const bool carry = qj >= bw;
    qj = carry ? qj - bw : qj;
    qi = carry ? qi + 1 : qi;
}

// Column / row halves of the coordinate numerators, tabulated once per band so that a pixel needs additions only
// (kml_col_half / kml_row_half: the very products km_gen_coord forms, so the sums round identically).
__device__ __forceinline__ KmlHalf kmt_half(const float4 e) {
    KmlHalf h;
    h.a = e.x; h.b = e.y; h.c = e.z;
    return h;
}

// grad_out of one output pixel, all CC channels (issued for every pixel of an unrolled group before any is used)
template <typename T, int CC>
__device__ __forceinline__ void kmt_load_go(const T* const (&gout_c)[CC], uint32_t off, float (&go)[CC]) {
#pragma unroll
    for (int c = 0; c < CC; ++c) go[c] = (float)km_ld(km_at(gout_c[c], off));
}

// A visited output pixel: position (the forward's own instruction sequence, km_lean.h), footprint, tile-relative north-west tap.
struct KmtPix {
    KmlPos p;
    KmlTaps t;
    float x, y;       // sampling position in source pixels
    uint32_t ux, uy;  // (floor(x), floor(y)) relative to the tile, unsigned
};

// the same for the padding modes that transform the position first (PADX: KM_PAD_BORDER / KM_PAD_REFLECTION; 0: none): ATen's
// compute_coordinates on the unnormalised position, gdx / gdy its derivative (0 where clamped, -1 on a mirrored stretch)
template <int CM, int ALIGN, bool FAST, int PADX>
__device__ __forceinline__ void kmt_pix_position_pad(const float (&m)[9], const KmlHalf& cu, const KmlHalf& rv, bool valid, float Wm1, float hW, float Hm1, float hH,
                                                     uint32_t X0, uint32_t Y0, int W, int H, KmtPix& q, float& gdx, float& gdy) {
    kml_position<CM, FAST>(m, cu, rv, q.p);
    q.x = kml_unnormalize<ALIGN>(q.p.gx, Wm1, hW);
    q.y = kml_unnormalize<ALIGN>(q.p.gy, Hm1, hH);
    gdx = 1.f; gdy = 1.f;
    if (PADX != 0) {
        q.x = km_compute_coord(q.x, W, PADX, ALIGN, gdx);
        q.y = km_compute_coord(q.y, H, PADX, ALIGN, gdy);
    }
    kml_taps(q.x, q.y, q.t);
    q.ux = (uint32_t)KM_F2I(q.t.xf) - X0;
    q.uy = valid ? (uint32_t)KM_F2I(q.t.yf) - Y0 : 0x40000000u;
}

template <int CM, int ALIGN, bool FAST>
__device__ __forceinline__ void kmt_pix_position(const float (&m)[9], const KmlHalf& cu, const KmlHalf& rv, bool valid, float Wm1, float hW, float Hm1, float hH,
                                                 uint32_t X0, uint32_t Y0, KmtPix& q) {
    kml_position<CM, FAST>(m, cu, rv, q.p);
    q.x = kml_unnormalize<ALIGN>(q.p.gx, Wm1, hW);
    q.y = kml_unnormalize<ALIGN>(q.p.gy, Hm1, hH);
    // weights: the forward's own expressions ((x0 + 1) - x, x - x0), so grad_src = W^T grad_out for the very W it applied
    kml_taps(q.x, q.y, q.t);
    // Tile-relative tap position in unsigned arithmetic.  A tap inside the tile is inside the image, so the in-tile test is
    // the whole predicate: positions far outside saturate in the conversion and wrap to values >= 2^30, a NaN position
    // converts to 0 - its weights are NaN, which the quantisation turns into 0 (FIXED) or which is excluded below (float path).
    q.ux = (uint32_t)KM_F2I(q.t.xf) - X0;
    q.uy = valid ? (uint32_t)KM_F2I(q.t.yf) - Y0 : 0x40000000u;
}
