
#include <algorithm>
#include <cmath>
#define __device__
#define __forceinline__ inline
using std::max;
using std::min;
enum { KM_COORD_PERSPECTIVE = 0, KM_COORD_AFFINE = 1, KM_COORD_HOMOGRAPHY = 2, KM_COORD_GRID = 3 };
static inline float km_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline double km_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// Per-launch geometry shared by forward and backward.
template <typename R>
struct KmWarpGeom {
    int B, C, H, W, h, w, B_M;
    int coord_mode, norm_coords, interp, pad, align;
    // torch.linspace parameters for KM_COORD_AFFINE (imgwarp.py:271-276), evaluated on the host in R
    R lin_lo_x, lin_hi_x, lin_step_x, lin_lo_y, lin_hi_y, lin_step_y;
};

template <typename R>
struct KmCoord {
    R u, v;    // base coordinates
    R gx, gy;  // normalised sampling coordinates
    R den;     // perspective: denominator; homography: scale s
    R X, Y;    // homography: numerators
    bool live; // homography: |Z| > eps
};

// create_meshgrid(normalized_coordinates=True), grid.py:73-75: (i / (n-1) - 0.5) * 2
__device__ __forceinline__ float km_mesh_f32(int i, int n) { return (((float)i / (float)(n - 1)) - 0.5f) * 2.0f; }
template <typename R>
__device__ __forceinline__ R km_mesh(int i, int n) {
    return (((R)i / (R)(n - 1)) - (R)0.5) * (R)2;
}

// torch.linspace scalar formula (two-sided, fused multiply-add)
template <typename R>
__device__ __forceinline__ R km_linspace(R lo, R hi, R step, int n, int i) {
    if (n == 1) return lo;
    if (i < n / 2) return km_fma(step, (R)i, lo);
    return km_fma(-step, (R)(n - 1 - i), hi);
}

// base coordinate along x (column j) / y (row i)
template <typename R, int CM>
__device__ __forceinline__ R km_base_x(const KmWarpGeom<R>& g, int j) {
    if (CM == KM_COORD_GRID) return (R)0;  // coordinates come from memory
    if (CM == KM_COORD_PERSPECTIVE) return (R)km_mesh_f32(j, g.w);  // always computed in fp32, then cast
    if (CM == KM_COORD_AFFINE) return km_linspace<R>(g.lin_lo_x, g.lin_hi_x, g.lin_step_x, g.w, j);
    return g.norm_coords ? km_mesh<R>(j, g.w) : (R)j;
}
template <typename R, int CM>
__device__ __forceinline__ R km_base_y(const KmWarpGeom<R>& g, int i) {
    if (CM == KM_COORD_GRID) return (R)0;
    if (CM == KM_COORD_PERSPECTIVE) return (R)km_mesh_f32(i, g.h);
    if (CM == KM_COORD_AFFINE) return km_linspace<R>(g.lin_lo_y, g.lin_hi_y, g.lin_step_y, g.h, i);
    return g.norm_coords ? km_mesh<R>(i, g.h) : (R)i;
}

#define KMT_TIGHT_BOX 1
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
        o.j0 = max(0, (int)floorf(jmin - mj) + 3); o.j1 = min(g.w - 1, (int)ceilf(jmax + mj));
        o.i0 = max(0, (int)floorf(imin - mi)); o.i1 = min(g.h - 1, (int)ceilf(imax + mi));
    }
    // output pixels per source pixel: the 2x2 footprint box maps to at most (2 ex + 1)(2 ey + 1) lattice points
    const R ex = jac_j + 0.1f, ey = jac_i + 0.1f;
    o.mult = fminf((2.f * ex + 1.f) * (2.f * ey + 1.f), 1.0e6f);
    o.fixed_ok = o.mult <= 256.f;  // beyond ~7x magnification the head-room would eat the mantissa: float path
    return o;
}

template <int CM>
static void run(const float* m9, int H, int W, int h, int w, int align, int norm, int X0, int X1, int Y0, int Y1, int* out) {
    KmWarpGeom<float> g;
    g.B = 1; g.C = 1; g.H = H; g.W = W; g.h = h; g.w = w; g.B_M = 1;
    g.coord_mode = CM; g.norm_coords = norm; g.interp = 1; g.pad = 0; g.align = align;
    if (align) { g.lin_lo_x = -1.0f; g.lin_hi_x = 1.0f; g.lin_lo_y = -1.0f; g.lin_hi_y = 1.0f; }
    else {
        g.lin_lo_x = (float)(-1.0 + 1.0 / w); g.lin_hi_x = (float)(1.0 - 1.0 / w);
        g.lin_lo_y = (float)(-1.0 + 1.0 / h); g.lin_hi_y = (float)(1.0 - 1.0 / h);
    }
    g.lin_step_x = w > 1 ? (g.lin_hi_x - g.lin_lo_x) / (float)(w - 1) : 0.0f;
    g.lin_step_y = h > 1 ? (g.lin_hi_y - g.lin_lo_y) / (float)(h - 1) : 0.0f;
    float m[9];
    for (int k = 0; k < 9; ++k) m[k] = m9[k];
    const KmtBox b = kmt_tile_box<CM>(g, m, X0, X1, Y0, Y1);
    out[0] = b.j0; out[1] = b.j1; out[2] = b.i0; out[3] = b.i1; out[4] = b.fixed_ok ? 1 : 0; out[5] = (int)b.mult;
}
extern "C" void tile_box(int cm, const float* m9, int H, int W, int h, int w, int align, int norm, int X0, int X1, int Y0, int Y1, int* out) {
    if (cm == 0) run<KM_COORD_PERSPECTIVE>(m9, H, W, h, w, align, norm, X0, X1, Y0, Y1, out);
    else if (cm == 1) run<KM_COORD_AFFINE>(m9, H, W, h, w, align, norm, X0, X1, Y0, Y1, out);
    else run<KM_COORD_HOMOGRAPHY>(m9, H, W, h, w, align, norm, X0, X1, Y0, Y1, out);
}
