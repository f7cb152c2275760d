// kornia_amd - lean per-pixel front end of the specialised bilinear kernels (forward, matrix gradient, tile-owner scatter).
//
// The three hot kernels are VALU-issue bound on gfx950 (profiles/r02_*): what costs time is the NUMBER of vector
// instructions per pixel, and among them the two IEEE divisions of the projective map (12 instructions each, several of
// them at half or third rate: profiles/r02_valu_rates.txt).  This header holds the same arithmetic as km_sampler.h's
// km_gen_coord / km_unnormalize / km_bilinear_setup - same operations, same order, same roundings, bit-identical results -
// with everything removed that the hot case does not need:
//
//   * both quotients share ONE refined reciprocal of the common denominator (kml_div2): the compiler's expansion of
//     `a / b` is  scale - rcp - refine - multiply - two residual corrections - fmas - fixup;  for operands in a range
//     where the scaling is the identity and no special case applies (checked once per block: kml_div_guard) the middle
//     of that sequence IS the correctly rounded quotient, and the refinement of the reciprocal depends on the
//     denominator only;
//   * the column halves (m0 u, m3 u, m6 u) and row halves (m1 v, m4 v, m7 v) of the numerators are formed once per
//     column / row, so a pixel costs the additions only (same products, same sums as km_gen_coord);
//   * bounds are decided once for the 2 x 2 footprint (inside / not), indices with 24-bit multiplies and 32-bit
//     offsets from a wave-uniform base.
#pragma once

#include "km_sampler.h"

// ---- division -------------------------------------------------------------------------------------------------------
// v_rcp_f32 (1 ulp) refined by one Newton step: the `r` of the compiler's own fp32 division expansion
__device__ __forceinline__ float kml_rcp_refined(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = km_fma(-d, r0, 1.0f);
    return km_fma(e, r0, r0);
}
// n / d given r = kml_rcp_refined(d): multiply, two residual corrections (the second one is what v_div_fmas_f32 computes
// when no scaling is pending).  Equal to the IEEE quotient for operands accepted by kml_div_guard.
__device__ __forceinline__ float kml_div_by(float n, float d, float r) {
    float q = n * r;
    float t = km_fma(-d, q, n);
    q = km_fma(t, r, q);
    t = km_fma(-d, q, n);
    return km_fma(t, r, q);
}
// 2^-40 <= |d| <= 2^40 and |n| <= 2^40 (NaN fails): far inside the range where v_div_scale_f32 leaves both operands alone
// (|exponent difference| < 96, no denormal operand, reciprocal or quotient other than a vanishing numerator, which rounds
// away in the `+ 1` that follows) and v_div_fixup_f32 passes the quotient through
__device__ __forceinline__ bool kml_div_operands_ok(float nx, float ny, float d) {
    const float ad = km_fabs(d);
    return (ad >= 9.094947e-13f) && (ad <= 1.0995116e12f) && (km_fabs(nx) <= 1.0995116e12f) && (km_fabs(ny) <= 1.0995116e12f);
}

// ---- coordinates ----------------------------------------------------------------------------------------------------
// column / row halves of the numerators (perspective, affine) or the raw base coordinate (homography mode needs v for its fma)
struct KmlHalf {
    float a, b, c;
};
template <int CM>
__device__ __forceinline__ KmlHalf kml_col_half(const float (&m)[9], float u) {
    KmlHalf h;
    h.a = m[0] * u; h.b = m[3] * u; h.c = m[6] * u;  // homography mode: u * m0 - the product commutes bit for bit
    return h;
}
template <int CM>
__device__ __forceinline__ KmlHalf kml_row_half(const float (&m)[9], float v) {
    KmlHalf h;
    if (CM == KM_COORD_HOMOGRAPHY) { h.a = v; h.b = 0.f; h.c = 0.f; }
    else { h.a = m[1] * v; h.b = m[4] * v; h.c = m[7] * v; }
    return h;
}

struct KmlPos {
    float gx, gy;   // normalised sampling position
    float den, rinv;  // perspective: denominator and its refined reciprocal (FAST) ; homography: s in den
    float X, Y;     // homography: numerators
    bool live;      // homography: |Z| > eps
};

// FAST: operands were range-checked for the whole block (kml_div_guard); otherwise plain IEEE divisions
template <int CM, bool FAST>
__device__ __forceinline__ void kml_position(const float (&m)[9], const KmlHalf& cu, const KmlHalf& rv, KmlPos& p) {
    if (CM == KM_COORD_PERSPECTIVE) {
        const float den = (cu.c + rv.c) + m[8];
        const float nx = (cu.a + rv.a) + m[2];
        const float ny = (cu.b + rv.b) + m[5];
        p.den = den;
        if (FAST) {
            const float r = kml_rcp_refined(den);
            p.rinv = r;
            p.gx = kml_div_by(nx, den, r);
            p.gy = kml_div_by(ny, den, r);
        } else {
            p.rinv = 0.f;
            p.gx = nx / den;
            p.gy = ny / den;
        }
    } else if (CM == KM_COORD_AFFINE) {
        p.den = 1.f;
        p.rinv = 1.f;
        p.gx = (cu.a + rv.a) + m[2];
        p.gy = (cu.b + rv.b) + m[5];
    } else {
        const float v = rv.a;
        const float X = km_fma(v, m[1], cu.a) + m[2];
        const float Y = km_fma(v, m[4], cu.b) + m[5];
        const float Z = km_fma(v, m[7], cu.c) + m[8];
        const float eps = 1e-8f;
        p.live = km_fabs(Z) > eps;
        float s;
        if (FAST) {
            const float d = Z + eps;
            const float r = kml_rcp_refined(d);
            s = p.live ? kml_div_by(1.0f, d, r) : 1.0f;
        } else {
            s = p.live ? 1.0f / (Z + eps) : 1.0f;
        }
        p.X = X; p.Y = Y; p.den = s; p.rinv = 0.f;
        p.gx = s * X;
        p.gy = s * Y;
    }
}

// Block-level guard for FAST: every pixel of the output rectangle [j0, j1] x [i0, i1] has division operands inside
// kml_div_operands_ok's range.  Numerators and denominator are affine in (u, v) up to rounding, so their extreme
// magnitudes over the rectangle are at its corners (the ranges above leave 2^80 of slack for the rounding), and the
// denominator must not change sign (else it passes through zero inside).
template <int CM>
__device__ __forceinline__ bool kml_div_guard(const KmWarpGeom<float>& g, const float (&m)[9], int j0, int j1, int i0, int i1) {
    if (CM == KM_COORD_AFFINE) return true;
    bool ok = true;
    float dmin = 3.0e38f, dmax = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float u = km_base_x<float, CM>(g, (k & 1) ? j1 : j0), v = km_base_y<float, CM>(g, (k & 2) ? i1 : i0);
        float nx, ny, d;
        if (CM == KM_COORD_PERSPECTIVE) {
            d = (m[6] * u + m[7] * v) + m[8];
            nx = (m[0] * u + m[1] * v) + m[2];
            ny = (m[3] * u + m[4] * v) + m[5];
        } else {
            d = (km_fma(v, m[7], u * m[6]) + m[8]) + 1e-8f;
            nx = 1.0f; ny = 1.0f;
        }
        ok = ok && kml_div_operands_ok(nx, ny, d);
        dmin = fminf(dmin, d); dmax = fmaxf(dmax, d);
    }
    return ok && ((dmin > 0.f) || (dmax < 0.f));
}

// the same test for ONE output row, at the two ends of the row's full base-coordinate range: what the threads that fill a
// block's row table evaluate (conservative: a row fails if its denominator vanishes anywhere across the image)
template <int CM>
__device__ __forceinline__ bool kml_row_guard(const KmWarpGeom<float>& g, const float (&m)[9], float v) {
    // conservative per-row form of kml_div_guard: the two ends of the row's FULL base-coordinate range
    if (CM == KM_COORD_AFFINE) return true;
    const bool pix = (CM == KM_COORD_HOMOGRAPHY) && !g.norm_coords;
    const float ue[2] = {pix ? 0.0f : -1.0f, pix ? (float)(g.w - 1) : 1.0f};
    bool ok = true;
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float nx, ny, d;
        if (CM == KM_COORD_PERSPECTIVE) {
            d = (m[6] * ue[k] + m[7] * v) + m[8];
            nx = (m[0] * ue[k] + m[1] * v) + m[2];
            ny = (m[3] * ue[k] + m[4] * v) + m[5];
        } else {
            d = (km_fma(v, m[7], ue[k] * m[6]) + m[8]) + 1e-8f;
            nx = 1.0f; ny = 1.0f;
        }
        ok = ok && kml_div_operands_ok(nx, ny, d);
        if (k == 0) d0 = d; else d1 = d;
    }
    return ok && ((d0 > 0.f) == (d1 > 0.f));
}

// ---- sampling position -> bilinear footprint -----------------------------------------------------------------------------
template <int ALIGN>
__device__ __forceinline__ float kml_unnormalize(float g, float size_m1, float half_size) {
    // km_unnormalize: align ? ((g + 1) / 2) * (size - 1) : fma(g + 1, size / 2, -0.5)
    if (ALIGN) return ((g + 1.0f) * 0.5f) * size_m1;  // x / 2 == x * 0.5 bit for bit
    return km_fma(g + 1.0f, half_size, -0.5f);
}

struct KmlTaps {
    float xf, yf;            // floor(x), floor(y)
    float wx0, wx1, wy0, wy1;  // (x - x0), (x1 - x), (y - y0), (y1 - y): km_bilinear_setup's expressions
};
__device__ __forceinline__ void kml_taps(float x, float y, KmlTaps& t) {
    t.xf = km_floor(x); t.yf = km_floor(y);
    t.wx0 = x - t.xf; t.wx1 = (t.xf + 1.0f) - x;
    t.wy0 = y - t.yf; t.wy1 = (t.yf + 1.0f) - y;
}
// all four taps inside the image (NaN / inf positions are outside)
__device__ __forceinline__ bool kml_inside(const KmlTaps& t, float Wm2, float Hm2) {
    return (t.xf >= 0.f) & (t.xf <= Wm2) & (t.yf >= 0.f) & (t.yf <= Hm2);
}
