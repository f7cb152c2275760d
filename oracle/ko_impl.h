/*
 * TEST INFRASTRUCTURE ONLY - CPU oracle for the kornia warp + filter hot path.
 *
 * This header is included twice by kornia_oracle.c (REAL=float, SFX=f32 and REAL=double,
 * SFX=f64).  It is a plain-C restatement of the arithmetic that the reference performs
 * through PyTorch ATen ops; every function cites the reference file:line it follows
 * (paths relative to /root/reference) or the ATen header that specifies the op
 * (torch/include/ATen/native/GridSampler.h, UpSample.h - torch is the reference's pinned
 * third-party dependency, pyproject.toml:29, not vendored in the reference tree).
 *
 * Compile with -ffp-contract=off: each torch op in the reference rounds its result to the
 * working dtype, so the restatement must not fuse multiply-adds except where the reference's
 * own CPU kernel does (torch.linalg.cross, see ko_cross3).
 *
 * Nothing under kornia_amd/ may include, link or call this file.
 */

#define KO_CAT2(a, b) a##_##b
#define KO_CAT(a, b) KO_CAT2(a, b)
#define KO(name) KO_CAT(name, SFX)

/* ------------------------------------------------------------------------------------------
 * 3x3 chain:  kornia/geometry/conversions.py:1691-1763 (normalize_homography,
 * normal_transform_pixel), kornia/core/utils.py:137-166 (_inverse_3x3_closed_form, eager branch).
 * Bit pattern pinned against the reference on CPU (tests/golden/chain_*.npz):
 *   - torch.linalg.cross's CPU kernel contracts a1*b2 - a2*b1 into fma(a1, b2, -(a2*b1));
 *   - (col_a * row0).sum(-1) is a left-to-right sum of three rounded products;
 *   - the 3x3 matmuls are plain multiply/add chains in k order (ATen baddbmm small-matrix path).
 * ---------------------------------------------------------------------------------------- */

static inline REAL KO(ko_fma)(REAL a, REAL b, REAL c) {
#if KO_IS_DOUBLE
    return fma(a, b, c);
#else
    return fmaf(a, b, c);
#endif
}

static void KO(ko_cross3)(const REAL a[3], const REAL b[3], REAL out[3]) {
    REAL t;
    t = a[2] * b[1];
    out[0] = KO(ko_fma)(a[1], b[2], -t);
    t = a[0] * b[2];
    out[1] = KO(ko_fma)(a[2], b[0], -t);
    t = a[1] * b[0];
    out[2] = KO(ko_fma)(a[0], b[1], -t);
}

/* inverse rows = (b x c, c x a, a x b) / det with a,b,c the COLUMNS; det = a . (b x c).
 * kornia/core/utils.py:159-166 */
static void KO(ko_inv3)(const REAL m[9], REAL out[9]) {
    REAL a[3] = {m[0], m[3], m[6]};
    REAL b[3] = {m[1], m[4], m[7]};
    REAL c[3] = {m[2], m[5], m[8]};
    REAL r0[3], r1[3], r2[3];
    KO(ko_cross3)(b, c, r0);
    KO(ko_cross3)(c, a, r1);
    KO(ko_cross3)(a, b, r2);
    REAL p0 = a[0] * r0[0], p1 = a[1] * r0[1], p2 = a[2] * r0[2];
    REAL det = (p0 + p1) + p2;
    for (int k = 0; k < 3; ++k) {
        out[k] = r0[k] / det;
        out[3 + k] = r1[k] / det;
        out[6 + k] = r2[k] / det;
    }
}

static void KO(ko_mm3)(const REAL a[9], const REAL b[9], REAL out[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            REAL acc = (REAL)0;
            for (int k = 0; k < 3; ++k) acc = acc + a[3 * i + k] * b[3 * k + j];
            out[3 * i + j] = acc;
        }
}

/* kornia/geometry/conversions.py:1729-1763: the matrix is built from Python doubles as a
 * float32 tensor (torch.tensor default dtype) and THEN cast to the homography's dtype, so the
 * scale entries are float32-rounded even for a float64 homography. */
static void KO(ko_normal_transform_pixel)(int height, int width, REAL out[9]) {
    double wd = (width == 1) ? 1e-14 : (double)width - 1.0;
    double hd = (height == 1) ? 1e-14 : (double)height - 1.0;
    float sx = (float)(2.0 / wd);
    float sy = (float)(2.0 / hd);
    out[0] = (REAL)sx; out[1] = 0; out[2] = -1;
    out[3] = 0; out[4] = (REAL)sy; out[5] = -1;
    out[6] = 0; out[7] = 0; out[8] = 1;
}

/* normalize_homography (conversions.py:1713-1725) followed by the inverse taken by the warps
 * (imgwarp.py:153, :254).  M: (B,9) pixel src->dst.  A_out (nullable): normalised src->dst,
 * m_out: normalised dst->src. */
void KO(ko_homography_chain)(const REAL* M, int B, int Hs, int Ws, int hd, int wd, REAL* A_out, REAL* m_out) {
    REAL Ns[9], Nsi[9], Nd[9];
    KO(ko_normal_transform_pixel)(Hs, Ws, Ns);
    KO(ko_inv3)(Ns, Nsi);
    KO(ko_normal_transform_pixel)(hd, wd, Nd);
    for (int b = 0; b < B; ++b) {
        REAL t[9], A[9];
        KO(ko_mm3)(M + 9 * b, Nsi, t);
        KO(ko_mm3)(Nd, t, A);
        if (A_out) memcpy(A_out + 9 * b, A, sizeof(A));
        if (m_out) KO(ko_inv3)(A, m_out + 9 * b);
    }
}

/* Adjoint of ko_homography_chain wrt M, evaluated in double: m = inv(A), A = Nd M Nsi
 *   gA = -m^T gm m^T ;  gM = Nd^T gA Nsi^T.  (what autograd derives for conversions.py:1725 +
 * core/utils.py:159-166). */
void KO(ko_homography_chain_bwd)(const REAL* M, const REAL* gm, int B, int Hs, int Ws, int hd, int wd, REAL* gM) {
    REAL Ns[9], Nsi[9], Nd[9];
    KO(ko_normal_transform_pixel)(Hs, Ws, Ns);
    KO(ko_inv3)(Ns, Nsi);
    KO(ko_normal_transform_pixel)(hd, wd, Nd);
    for (int b = 0; b < B; ++b) {
        REAL t[9], A[9], mi[9];
        KO(ko_mm3)(M + 9 * b, Nsi, t);
        KO(ko_mm3)(Nd, t, A);
        KO(ko_inv3)(A, mi);
        double m_[9], g_[9], t1[9], gA[9], t2[9];
        for (int i = 0; i < 9; ++i) { m_[i] = mi[i]; g_[i] = gm[9 * b + i]; }
        /* t1 = m^T g */
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            double s = 0; for (int k = 0; k < 3; ++k) s += m_[3 * k + i] * g_[3 * k + j]; t1[3 * i + j] = s; }
        /* gA = -t1 m^T */
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            double s = 0; for (int k = 0; k < 3; ++k) s += t1[3 * i + k] * m_[3 * j + k]; gA[3 * i + j] = -s; }
        /* t2 = Nd^T gA */
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            double s = 0; for (int k = 0; k < 3; ++k) s += (double)Nd[3 * k + i] * gA[3 * k + j]; t2[3 * i + j] = s; }
        /* gM = t2 Nsi^T */
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            double s = 0; for (int k = 0; k < 3; ++k) s += t2[3 * i + k] * (double)Nsi[3 * j + k]; gM[9 * b + 3 * i + j] = (REAL)s; }
    }
}

/* ------------------------------------------------------------------------------------------
 * Base coordinates.
 * ---------------------------------------------------------------------------------------- */

/* create_meshgrid(normalized_coordinates=True): kornia/geometry/grid.py:65-77
 *   xs = linspace(0, n-1, n) (exact integers) ; xs = (xs / (n-1) - 0.5) * 2 */
static inline float KO(ko_mesh_f32)(int i, int n) { return (((float)i / (float)(n - 1)) - 0.5f) * 2.0f; }
static inline REAL KO(ko_mesh_real)(int i, int n) { return (((REAL)i / (REAL)(n - 1)) - (REAL)0.5) * (REAL)2; }

/* torch.linspace(lo, hi, n) scalar formula (ATen RangeFactories: step = (hi-lo)/(n-1);
 * i < n/2 ? lo + step*i : hi - step*(n-1-i)), evaluated with a fused multiply-add as both the
 * ATen CPU scalar tail and the GPU kernel do.  imgwarp.py:271-276. */
static inline REAL KO(ko_linspace)(REAL lo, REAL hi, int n, int i) {
    if (n == 1) return lo;
    REAL step = (hi - lo) / (REAL)(n - 1);
    if (i < n / 2) return KO(ko_fma)(step, (REAL)i, lo);
    return KO(ko_fma)(-step, (REAL)(n - 1 - i), hi);
}

/* ------------------------------------------------------------------------------------------
 * Sampler primitives: ATen GridSampler.h (grid_sampler_unnormalize, clip_coordinates,
 * reflect_coordinates, compute_coordinates, get_value_bounded) and UpSample.h:400-423.
 * ---------------------------------------------------------------------------------------- */
#if KO_IS_DOUBLE
#define KO_FLOOR floor
#define KO_FABS fabs
#define KO_FMOD fmod
#define KO_NEARBYINT nearbyint
#define KO_SQRT sqrt
#else
#define KO_FLOOR floorf
#define KO_FABS fabsf
#define KO_FMOD fmodf
#define KO_NEARBYINT nearbyintf
#define KO_SQRT sqrtf
#endif

static inline REAL KO(ko_unnormalize)(REAL g, int size, int align, REAL* mult) {
    if (align) { *mult = (REAL)(size - 1) / 2; return ((g + 1) / 2) * (REAL)(size - 1); }
    *mult = (REAL)size / 2;
    /* GridSampler.h writes ((g + 1) * size - 1) / 2; ATen's CPU kernel (and a contracting GPU
     * compiler) evaluate it as one fused multiply-add (g + 1) * (size / 2) - 0.5 - pinned bit-exactly
     * against the reference on CPU (tests/golden). */
    return KO(ko_fma)(g + 1, (REAL)size / 2, (REAL)-0.5);
}

static inline REAL KO(ko_clip)(REAL x, int size, REAL* grad) {
    if (x <= (REAL)0) { *grad = 0; return 0; }
    REAL mx = (REAL)(size - 1);
    if (x >= mx) { *grad = 0; return mx; }
    *grad = 1;
    return x;
}

static inline REAL KO(ko_reflect)(REAL x, int twice_low, int twice_high, REAL* grad) {
    if (twice_low == twice_high) { *grad = 0; return 0; }
    int sign;
    REAL mn = (REAL)twice_low / 2;
    REAL span = (REAL)(twice_high - twice_low) / 2;
    x = x - mn;
    if (x < (REAL)0) { sign = -1; x = -x; } else { sign = 1; }
    REAL extra = KO_FMOD(x, span);
    int flips = (int)KO_FLOOR(x / span);
    if (flips % 2 == 0) { *grad = (REAL)sign; return extra + mn; }
    *grad = (REAL)(-sign);
    return span - extra + mn;
}

/* pad: 0 zeros, 1 border, 2 reflection */
static inline REAL KO(ko_compute_coord)(REAL x, int size, int pad, int align, REAL* grad) {
    *grad = 1;
    if (pad == 1) {
        x = KO(ko_clip)(x, size, grad);
    } else if (pad == 2) {
        REAL gr, gc;
        if (align) x = KO(ko_reflect)(x, 0, 2 * (size - 1), &gr);
        else x = KO(ko_reflect)(x, -1, 2 * size - 1, &gr);
        x = KO(ko_clip)(x, size, &gc);
        *grad = gr * gc;
    }
    return x;
}

static inline void KO(ko_cubic_coeffs)(REAL t, REAL c[4]) {
    const REAL A = (REAL)-0.75;
    REAL x = t + (REAL)1.0;
    c[0] = ((A * x - 5 * A) * x + 8 * A) * x - 4 * A;
    x = t;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    x = (REAL)1.0 - t;
    c[2] = ((A + 2) * x - (A + 3)) * x * x + 1;
    x = x + (REAL)1.0;
    c[3] = ((A * x - 5 * A) * x + 8 * A) * x - 4 * A;
}

static inline REAL KO(ko_dot4)(const REAL v[4], const REAL c[4]) {
    return v[0] * c[0] + v[1] * c[1] + v[2] * c[2] + v[3] * c[3];
}

static inline void KO(ko_cubic_coeffs_grad)(REAL t, REAL c[4]) {
    const REAL A = (REAL)-0.75;
    REAL x = -1 - t;
    c[0] = (-3 * A * x - 10 * A) * x - 8 * A;
    x = -t;
    c[1] = (-3 * (A + 2) * x - 2 * (A + 3)) * x;
    x = 1 - t;
    c[2] = (3 * (A + 2) * x - 2 * (A + 3)) * x;
    x = 2 - t;
    c[3] = (3 * A * x - 10 * A) * x + 8 * A;
}

/* value at an integer tap that is itself passed through the padding transform (bicubic) */
static inline REAL KO(ko_tap_bounded)(const REAL* img, REAL x, REAL y, int W, int H, int pad, int align) {
    REAL g;
    x = KO(ko_compute_coord)(x, W, pad, align, &g);
    y = KO(ko_compute_coord)(y, H, pad, align, &g);
    long ix = (long)x, iy = (long)y;
    if (ix >= 0 && ix < W && iy >= 0 && iy < H) return img[iy * (long)W + ix];
    return 0;
}

static inline void KO(ko_add_bounded)(REAL* img, REAL x, REAL y, int W, int H, int pad, int align, REAL delta) {
    REAL g;
    x = KO(ko_compute_coord)(x, W, pad, align, &g);
    y = KO(ko_compute_coord)(y, H, pad, align, &g);
    long ix = (long)x, iy = (long)y;
    if (ix >= 0 && ix < W && iy >= 0 && iy < H) img[iy * (long)W + ix] += delta;
}

/* ------------------------------------------------------------------------------------------
 * Coordinate generation for one output pixel.
 *   coord_mode 0  warp_perspective   imgwarp.py:157-170  (base grid in fp32 then cast)
 *   coord_mode 1  warp_affine        imgwarp.py:271-281  (linspace base in REAL, no divide)
 *   coord_mode 2  homography_warp    imgwarp.py:1541-1546 -> warp_grid :323-353 ->
 *                 transform_points linalg.py:219-239 -> convert_points_from_homogeneous
 *                 conversions.py:303-307 (base grid in REAL; normalized_coordinates flag)
 * m: 9 values of the matrix that maps base coords -> sampling coords.
 * Outputs gx, gy and the auxiliaries needed for the matrix gradient.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    REAL u, v;     /* base coords */
    REAL gx, gy;   /* normalised sampling coords */
    REAL den;      /* mode 0: denominator ; mode 2: s (scale) */
    REAL X, Y;     /* mode 2: numerators */
    int live;      /* mode 2: |Z| > eps */
} KO(ko_coord_t);

static inline void KO(ko_gen_coord)(const REAL* m, int coord_mode, int align, int norm_coords, int i, int j, int h,
                                    int w, KO(ko_coord_t) * o) {
    if (coord_mode == 0) {
        REAL u = (REAL)KO(ko_mesh_f32)(j, w), v = (REAL)KO(ko_mesh_f32)(i, h);
        REAL den = (m[6] * u + m[7] * v) + m[8];
        o->u = u; o->v = v; o->den = den;
        o->gx = ((m[0] * u + m[1] * v) + m[2]) / den;
        o->gy = ((m[3] * u + m[4] * v) + m[5]) / den;
    } else if (coord_mode == 1) {
        REAL u, v;
        if (align) { u = KO(ko_linspace)((REAL)-1, (REAL)1, w, j); v = KO(ko_linspace)((REAL)-1, (REAL)1, h, i); }
        else {
            u = KO(ko_linspace)((REAL)(-1.0 + 1.0 / w), (REAL)(1.0 - 1.0 / w), w, j);
            v = KO(ko_linspace)((REAL)(-1.0 + 1.0 / h), (REAL)(1.0 - 1.0 / h), h, i);
        }
        o->u = u; o->v = v; o->den = 1;
        o->gx = (m[0] * u + m[1] * v) + m[2];
        o->gy = (m[3] * u + m[4] * v) + m[5];
    } else if (coord_mode == 3) {
        /* explicit sampling grid (remap, imgwarp.py:625-702 / F.grid_sample): m points at this image's (h,w,2) grid */
        o->u = 0; o->v = 0; o->den = 1;
        o->gx = m[((size_t)i * w + j) * 2 + 0];
        o->gy = m[((size_t)i * w + j) * 2 + 1];
    } else {
        REAL u, v;
        if (norm_coords) { u = KO(ko_mesh_real)(j, w); v = KO(ko_mesh_real)(i, h); }
        else { u = (REAL)j; v = (REAL)i; }
        /* bmm([u v 1], H^T): k-ordered fma chain acc = fma(p_k, H_rk, acc) starting from 0 - the
         * accumulation the CPU BLAS behind torch.bmm performs for K = 3 (pinned bit-exactly against
         * the reference on CPU, tests/golden). */
        REAL X = KO(ko_fma)(v, m[1], u * m[0]) + m[2];
        REAL Y = KO(ko_fma)(v, m[4], u * m[3]) + m[5];
        REAL Z = KO(ko_fma)(v, m[7], u * m[6]) + m[8];
        const REAL eps = (REAL)1e-8;
        int live = KO_FABS(Z) > eps;
        REAL s = live ? (REAL)1.0 / (Z + eps) : (REAL)1.0;
        o->u = u; o->v = v; o->X = X; o->Y = Y; o->den = s; o->live = live;
        o->gx = s * X;
        o->gy = s * Y;
    }
}

/* ------------------------------------------------------------------------------------------
 * Warp forward.  src (B,C,H,W) contiguous, mat (B_M,9), out (B,C,h,w).
 *   coord_mode 0 perspective 1 affine 2 homography 3 explicit grid (mat = (B_M,h,w,2) normalised grid, gmat must be NULL)
 *   interp 0 nearest 1 bilinear 2 bicubic ; pad 0 zeros 1 border 2 reflection 3 fill
 *   fill (C values, used when pad == 3): imgwarp.py:293-320 (_fill_and_warp)
 * ---------------------------------------------------------------------------------------- */
void KO(ko_warp2d_fwd)(const REAL* src, const REAL* mat, REAL* out, int B, int C, int H, int W, int h, int w, int B_M,
                       int coord_mode, int norm_coords, int interp, int pad, int align, const REAL* fill) {
    const int spad = (pad == 3) ? 0 : pad;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < h; ++i) {
            const size_t mstride = (coord_mode == 3) ? (size_t)2 * h * w : (size_t)9;  /* mode 3: mat is the (B_M,h,w,2) grid */
            const REAL* m = mat + mstride * (size_t)(B_M == 1 ? 0 : b);
            for (int j = 0; j < w; ++j) {
                KO(ko_coord_t) cd;
                KO(ko_gen_coord)(m, coord_mode, align, norm_coords, i, j, h, w, &cd);
                REAL mx, my, gdx, gdy;
                REAL x = KO(ko_unnormalize)(cd.gx, W, align, &mx);
                REAL y = KO(ko_unnormalize)(cd.gy, H, align, &my);
                if (interp == 1) {
                    x = KO(ko_compute_coord)(x, W, spad, align, &gdx);
                    y = KO(ko_compute_coord)(y, H, spad, align, &gdy);
                    REAL xf = KO_FLOOR(x), yf = KO_FLOOR(y);
                    long x0 = (long)xf, y0 = (long)yf, x1 = x0 + 1, y1 = y0 + 1;
                    REAL nw = ((REAL)x1 - x) * ((REAL)y1 - y), ne = (x - (REAL)x0) * ((REAL)y1 - y);
                    REAL sw = ((REAL)x1 - x) * (y - (REAL)y0), se = (x - (REAL)x0) * (y - (REAL)y0);
                    /* A tap outside the image is a ZERO THAT IS STILL MULTIPLIED by its weight: ATen's CPU sampler (GridSamplerKernel.cpp,
                     * ApplyGridSample<..., Bilinear, ...>::forward) gathers masked-out taps as 0 (mask_gather) and runs the same fma chain.  For
                     * finite weights that leaves the sum unchanged; a NaN / inf coordinate (singular matrix, zero projective denominator)
                     * converts to LONG_MIN here - all four taps masked - and its NaN weights make the pixel NaN, as in the reference
                     * (pinned by tests/golden/nonfinite_coords.npz; rounds 1-5 skipped the taps instead and returned 0 / the fill colour). */
                    int bnw = (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H), bne = (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H);
                    int bsw = (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H), bse = (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H);
                    REAL mask = 0;
                    if (pad == 3) {  /* grid_sample(ones): the same chain on an image of ones (imgwarp.py:316) */
                        mask = KO(ko_fma)(bnw ? (REAL)1 : (REAL)0, nw, (REAL)0);
                        mask = KO(ko_fma)(bne ? (REAL)1 : (REAL)0, ne, mask);
                        mask = KO(ko_fma)(bsw ? (REAL)1 : (REAL)0, sw, mask);
                        mask = KO(ko_fma)(bse ? (REAL)1 : (REAL)0, se, mask);
                    }
                    for (int c = 0; c < C; ++c) {
                        const REAL* img = src + ((size_t)b * C + c) * H * W;
                        /* fma chain nw, ne, sw, se from 0: bit-exact with ATen's CPU kernel (tests/golden) */
                        REAL acc = KO(ko_fma)(bnw ? img[y0 * (long)W + x0] : (REAL)0, nw, (REAL)0);
                        acc = KO(ko_fma)(bne ? img[y0 * (long)W + x1] : (REAL)0, ne, acc);
                        acc = KO(ko_fma)(bsw ? img[y1 * (long)W + x0] : (REAL)0, sw, acc);
                        acc = KO(ko_fma)(bse ? img[y1 * (long)W + x1] : (REAL)0, se, acc);
                        if (pad == 3) acc = acc + ((REAL)1 - mask) * fill[c];
                        out[(((size_t)b * C + c) * h + i) * w + j] = acc;
                    }
                } else if (interp == 0) {
                    x = KO(ko_compute_coord)(x, W, spad, align, &gdx);
                    y = KO(ko_compute_coord)(y, H, spad, align, &gdy);
                    long xn = (long)KO_NEARBYINT(x), yn = (long)KO_NEARBYINT(y);
                    int inb = (xn >= 0 && xn < W && yn >= 0 && yn < H);
                    for (int c = 0; c < C; ++c) {
                        const REAL* img = src + ((size_t)b * C + c) * H * W;
                        REAL acc = inb ? img[yn * (long)W + xn] : (REAL)0;
                        if (pad == 3) acc = acc + ((REAL)1 - (inb ? (REAL)1 : (REAL)0)) * fill[c];
                        out[(((size_t)b * C + c) * h + i) * w + j] = acc;
                    }
                } else {
                    REAL xf = KO_FLOOR(x), yf = KO_FLOOR(y);
                    REAL tx = x - xf, ty = y - yf;
                    REAL cx[4], cy[4];
                    KO(ko_cubic_coeffs)(tx, cx);
                    KO(ko_cubic_coeffs)(ty, cy);
                    REAL mask = 0;
                    if (pad == 3) {
                        /* bicubic sample of an all-ones image with zeros padding */
                        REAL rows[4];
                        for (int r = 0; r < 4; ++r) {
                            REAL t[4];
                            for (int q = 0; q < 4; ++q) {
                                long xx = (long)(xf - 1 + q), yy = (long)(yf - 1 + r);
                                t[q] = (xx >= 0 && xx < W && yy >= 0 && yy < H) ? (REAL)1 : (REAL)0;
                            }
                            rows[r] = KO(ko_dot4)(t, cx);
                        }
                        mask = KO(ko_dot4)(rows, cy);
                    }
                    for (int c = 0; c < C; ++c) {
                        const REAL* img = src + ((size_t)b * C + c) * H * W;
                        REAL rows[4];
                        for (int r = 0; r < 4; ++r) {
                            REAL t[4];
                            for (int q = 0; q < 4; ++q)
                                t[q] = KO(ko_tap_bounded)(img, xf - 1 + q, yf - 1 + r, W, H, spad, align);
                            rows[r] = KO(ko_dot4)(t, cx);
                        }
                        REAL acc = KO(ko_dot4)(rows, cy);
                        if (pad == 3) acc = acc + ((REAL)1 - mask) * fill[c];
                        out[(((size_t)b * C + c) * h + i) * w + j] = acc;
                    }
                }
            }
        }
}

/* ------------------------------------------------------------------------------------------
 * Warp backward (ATen grid_sampler_2d_backward + autograd of the coordinate chain).
 *   gsrc (B,C,H,W) is overwritten (nullable); gmat (B_M,9) nullable, accumulated in double;
 *   ggrid (B,h,w,2) nullable debugging output (gradient wrt the normalised sampling grid).
 * Sequential per image => deterministic.  Nearest has zero grid gradient.
 * ---------------------------------------------------------------------------------------- */
void KO(ko_warp2d_bwd)(const REAL* gout, const REAL* src, const REAL* mat, REAL* gsrc, REAL* gmat, REAL* ggrid, int B,
                       int C, int H, int W, int h, int w, int B_M, int coord_mode, int norm_coords, int interp, int pad,
                       int align, const REAL* fill) {
    const int spad = (pad == 3) ? 0 : pad;
    double* gm_all = (double*)calloc((size_t)B * 9, sizeof(double));
    if (gsrc) memset(gsrc, 0, (size_t)B * C * H * W * sizeof(REAL));
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const size_t mstride = (coord_mode == 3) ? (size_t)2 * h * w : (size_t)9;  /* mode 3: mat is the (B_M,h,w,2) grid */
        const REAL* m = mat + mstride * (size_t)(B_M == 1 ? 0 : b);
        double* gm = gm_all + 9 * (size_t)b;
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j) {
                KO(ko_coord_t) cd;
                KO(ko_gen_coord)(m, coord_mode, align, norm_coords, i, j, h, w, &cd);
                REAL mx, my, gdx = 1, gdy = 1;
                REAL x = KO(ko_unnormalize)(cd.gx, W, align, &mx);
                REAL y = KO(ko_unnormalize)(cd.gy, H, align, &my);
                REAL gix = 0, giy = 0;
                if (interp == 1) {
                    x = KO(ko_compute_coord)(x, W, spad, align, &gdx);
                    y = KO(ko_compute_coord)(y, H, spad, align, &gdy);
                    REAL xf = KO_FLOOR(x), yf = KO_FLOOR(y);
                    long x0 = (long)xf, y0 = (long)yf, x1 = x0 + 1, y1 = y0 + 1;
                    REAL wx1 = (REAL)x1 - x, wx0 = x - (REAL)x0, wy1 = (REAL)y1 - y, wy0 = y - (REAL)y0;
                    REAL nw = wx1 * wy1, ne = wx0 * wy1, sw = wx1 * wy0, se = wx0 * wy0;
                    int bnw = (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H), bne = (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H);
                    int bsw = (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H), bse = (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H);
                    for (int c = 0; c < C; ++c) {
                        const REAL* img = src + ((size_t)b * C + c) * H * W;
                        REAL g = gout[(((size_t)b * C + c) * h + i) * w + j];
                        REAL* gi = gsrc ? gsrc + ((size_t)b * C + c) * H * W : NULL;
                        REAL f = (pad == 3) ? fill[c] : (REAL)0;
                        if (bnw) { if (gi) gi[y0 * (long)W + x0] += nw * g; REAL v = img[y0 * (long)W + x0] - f; gix -= v * wy1 * g; giy -= v * wx1 * g; }
                        if (bne) { if (gi) gi[y0 * (long)W + x1] += ne * g; REAL v = img[y0 * (long)W + x1] - f; gix += v * wy1 * g; giy -= v * wx0 * g; }
                        if (bsw) { if (gi) gi[y1 * (long)W + x0] += sw * g; REAL v = img[y1 * (long)W + x0] - f; gix -= v * wy0 * g; giy += v * wx1 * g; }
                        if (bse) { if (gi) gi[y1 * (long)W + x1] += se * g; REAL v = img[y1 * (long)W + x1] - f; gix += v * wy0 * g; giy += v * wx0 * g; }
                        /* An out-of-bounds tap is a ZERO that is still multiplied: ATen's CPU kernel (GridSamplerKernel.cpp, ApplyGridSample<..., Bilinear,
                         * ...>::backward) gathers masked-out taps as 0 and forms ((ne - nw) * s + (se - sw) * n) * gOut, so an inf / NaN grad_out at a
                         * pixel that samples outside the image makes the grid - hence the matrix - gradient NaN there (checked against
                         * torch.nn.functional.grid_sample on CPU in the build container).  For finite values the terms below are +-0. */
                        { const REAL z = (REAL)0;
                          if (!bnw) { gix -= z * wy1 * g; giy -= z * wx1 * g; }
                          if (!bne) { gix += z * wy1 * g; giy -= z * wx0 * g; }
                          if (!bsw) { gix -= z * wy0 * g; giy += z * wx1 * g; }
                          if (!bse) { gix += z * wy0 * g; giy += z * wx0 * g; } }
                    }
                    gix = gix * (mx * gdx);
                    giy = giy * (my * gdy);
                } else if (interp == 0) {
                    x = KO(ko_compute_coord)(x, W, spad, align, &gdx);
                    y = KO(ko_compute_coord)(y, H, spad, align, &gdy);
                    long xn = (long)KO_NEARBYINT(x), yn = (long)KO_NEARBYINT(y);
                    if (gsrc && xn >= 0 && xn < W && yn >= 0 && yn < H)
                        for (int c = 0; c < C; ++c)
                            gsrc[((size_t)b * C + c) * H * W + yn * (long)W + xn] += gout[(((size_t)b * C + c) * h + i) * w + j];
                } else {
                    REAL xf = KO_FLOOR(x), yf = KO_FLOOR(y);
                    REAL tx = x - xf, ty = y - yf;
                    REAL cx[4], cy[4], dx[4], dy[4];
                    KO(ko_cubic_coeffs)(tx, cx);
                    KO(ko_cubic_coeffs)(ty, cy);
                    KO(ko_cubic_coeffs_grad)(tx, dx);
                    KO(ko_cubic_coeffs_grad)(ty, dy);
                    for (int c = 0; c < C; ++c) {
                        const REAL* img = src + ((size_t)b * C + c) * H * W;
                        REAL g = gout[(((size_t)b * C + c) * h + i) * w + j];
                        REAL* gi = gsrc ? gsrc + ((size_t)b * C + c) * H * W : NULL;
                        REAL f = (pad == 3) ? fill[c] : (REAL)0;
                        for (int r = 0; r < 4; ++r)
                            for (int q = 0; q < 4; ++q) {
                                if (gi) KO(ko_add_bounded)(gi, xf - 1 + q, yf - 1 + r, W, H, spad, align, g * cx[q] * cy[r]);
                                REAL v = KO(ko_tap_bounded)(img, xf - 1 + q, yf - 1 + r, W, H, spad, align);
                                if (pad == 3) {
                                    long xx = (long)(xf - 1 + q), yy = (long)(yf - 1 + r);
                                    if (xx >= 0 && xx < W && yy >= 0 && yy < H) v -= f;
                                }
                                gix -= v * dx[q] * cy[r] * g;
                                giy -= v * dy[r] * cx[q] * g;
                            }
                    }
                    gix = gix * mx;
                    giy = giy * my;
                }
                if (ggrid) {
                    ggrid[(((size_t)b * h + i) * w + j) * 2 + 0] = gix;
                    ggrid[(((size_t)b * h + i) * w + j) * 2 + 1] = giy;
                }
                if (gmat) {
                    double ggx = gix, ggy = giy, r[3] = {cd.u, cd.v, 1.0};
                    if (coord_mode == 0) {
                        double inv = 1.0 / (double)cd.den;
                        double tz = -(ggx * (double)cd.gx + ggy * (double)cd.gy) * inv;
                        for (int k = 0; k < 3; ++k) { gm[k] += ggx * r[k] * inv; gm[3 + k] += ggy * r[k] * inv; gm[6 + k] += tz * r[k]; }
                    } else if (coord_mode == 1) {
                        for (int k = 0; k < 3; ++k) { gm[k] += ggx * r[k]; gm[3 + k] += ggy * r[k]; }
                    } else {
                        double s = cd.den;
                        double tz = cd.live ? -(ggx * (double)cd.X + ggy * (double)cd.Y) * s * s : 0.0;
                        for (int k = 0; k < 3; ++k) { gm[k] += ggx * s * r[k]; gm[3 + k] += ggy * s * r[k]; gm[6 + k] += tz * r[k]; }
                    }
                }
            }
    }
    if (gmat) {
        if (B_M == 1) {
            for (int k = 0; k < 9; ++k) { double s = 0; for (int b = 0; b < B; ++b) s += gm_all[9 * (size_t)b + k]; gmat[k] = (REAL)s; }
        } else {
            for (size_t k = 0; k < (size_t)B * 9; ++k) gmat[k] = (REAL)gm_all[k];
        }
    }
    free(gm_all);
}

/* ------------------------------------------------------------------------------------------
 * transform_points: kornia/geometry/linalg.py:183-239, conversions.py:247-339.
 * T (B_T,(D+1)^2) row-major, pts (B,N,D), D in {2,3}.
 * ---------------------------------------------------------------------------------------- */
void KO(ko_transform_points)(const REAL* T, const REAL* pts, REAL* out, int B, int N, int D, int B_T) {
    const int E = D + 1;
    const REAL eps = (REAL)1e-8;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        const REAL* t = T + (size_t)(B_T == 1 ? 0 : b) * E * E;
        for (int n = 0; n < N; ++n) {
            const REAL* p = pts + ((size_t)b * N + n) * D;
            REAL hp[4];
            for (int r = 0; r < E; ++r) {
                REAL acc = p[0] * t[r * E];
                for (int k = 1; k < D; ++k) acc = KO(ko_fma)(p[k], t[r * E + k], acc);
                acc = acc + t[r * E + D];
                hp[r] = acc;
            }
            REAL z = hp[D];
            REAL s = (KO_FABS(z) > eps) ? (REAL)1.0 / (z + eps) : (REAL)1.0;
            for (int k = 0; k < D; ++k) out[((size_t)b * N + n) * D + k] = s * hp[k];
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * filter2d: kornia/filters/filter.py:122-150 (+ _compute_padding :31-51).
 * The kernel passed here is the *prepared* one (already flipped for 'conv', L1-normalised,
 * cast to the input dtype): y[b,c,i,j] = sum_{p,q} k[b % Bk][p][q] * xpad[b,c,i+p,j+q],
 * accumulated as an fma chain in (p, q) order from 0 (the reference's conv backend - mkldnn on
 * CPU, MIOpen on GPU - does not define an order; measured |oracle - reference| <= 2e-7).
 *   border 0 constant 1 reflect 2 replicate 3 circular ; same=1 -> output HxW, else valid.
 * ---------------------------------------------------------------------------------------- */
static inline long KO(ko_border_index)(long s, long n, int border) {
    /* maps an unpadded coordinate s (may be <0 or >=n) to a source index or -1 (zero) */
    if (s >= 0 && s < n) return s;
    switch (border) {
        case 1: /* reflect (no edge repeat): torch reflection_pad requires pad < n */
            if (s < 0) s = -s;
            if (s >= n) s = 2 * (n - 1) - s;
            return (s >= 0 && s < n) ? s : -1;
        case 2: return s < 0 ? 0 : n - 1;
        case 3: { long r = s % n; if (r < 0) r += n; return r; }
        default: return -1;
    }
}

void KO(ko_filter2d_fwd)(const REAL* x, const REAL* k, REAL* y, int B, int C, int H, int W, int Bk, int kH, int kW,
                         int border, int same) {
    const int pt = (kH - 1) / 2, pl = (kW - 1) / 2;
    const int Ho = same ? H : H - kH + 1, Wo = same ? W : W - kW + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const REAL* img = x + ((size_t)b * C + c) * H * W;
            const REAL* kk = k + (size_t)(b % Bk) * kH * kW;
            REAL* o = y + ((size_t)b * C + c) * Ho * Wo;
            for (int i = 0; i < Ho; ++i)
                for (int j = 0; j < Wo; ++j) {
                    REAL acc = 0;
                    for (int p = 0; p < kH; ++p) {
                        long yy = same ? KO(ko_border_index)((long)i + p - pt, H, border) : (long)i + p;
                        for (int q = 0; q < kW; ++q) {
                            long xx = same ? KO(ko_border_index)((long)j + q - pl, W, border) : (long)j + q;
                            REAL v = (yy >= 0 && xx >= 0) ? img[yy * (long)W + xx] : (REAL)0;
                            acc = KO(ko_fma)(kk[p * kW + q], v, acc);
                        }
                    }
                    o[(size_t)i * Wo + j] = acc;
                }
        }
}

/* ------------------------------------------------------------------------------------------
 * Bilinear resize = F.interpolate(mode='bilinear') as called by pyrdown / pyrup
 * (kornia/geometry/transform/pyramid.py:447-452, 494-496).  The arithmetic is ATen's:
 *   scale = align ? (in-1)/(out-1) (0 when out <= 1) : in/out          (UpSample.h area_pixel_compute_scale)
 *   src   = align ? scale*d : max(scale*(d+0.5) - 0.5, 0)               (area_pixel_compute_source_index)
 *   i0 = (int)src ; i1 = i0 + (i0 < in-1) ; l1 = src - i0 ; l0 = 1 - l1
 *   out = h0*(w0*x00 + w1*x01) + h1*(w0*x10 + w1*x11)                   (device kernel upsample_bilinear2d_out_frame)
 * ATen's *CPU* kernel groups the same four products differently (vectorised); measured
 * |this - F.interpolate on CPU| <= 1.2e-7 for inputs in [0,1] - the fixtures are compared with a tolerance.
 * ---------------------------------------------------------------------------------------- */
static inline void KO(ko_resize_axis)(int d, int n_in, int n_out, int align, int* i0, int* i1, REAL* l0, REAL* l1) {
    REAL scale, src;
    if (align) {
        scale = n_out > 1 ? (REAL)(n_in - 1) / (REAL)(n_out - 1) : (REAL)0;
        src = scale * (REAL)d;
    } else {
        scale = (REAL)n_in / (REAL)n_out;
        src = scale * ((REAL)d + (REAL)0.5) - (REAL)0.5;
        if (src < (REAL)0) src = (REAL)0;
    }
    *i0 = (int)src;
    if (*i0 > n_in - 1) *i0 = n_in - 1;
    *i1 = *i0 + (*i0 < n_in - 1 ? 1 : 0);
    *l1 = src - (REAL)*i0;
    *l0 = (REAL)1 - *l1;
}

void KO(ko_resize_bilinear_fwd)(const REAL* x, REAL* y, int BC, int H, int W, int oh, int ow, int align) {
#pragma omp parallel for schedule(static)
    for (int bc = 0; bc < BC; ++bc) {
        const REAL* img = x + (size_t)bc * H * W;
        REAL* o = y + (size_t)bc * oh * ow;
        for (int i = 0; i < oh; ++i) {
            int y0, y1;
            REAL h0, h1;
            KO(ko_resize_axis)(i, H, oh, align, &y0, &y1, &h0, &h1);
            for (int j = 0; j < ow; ++j) {
                int x0, x1;
                REAL w0, w1;
                KO(ko_resize_axis)(j, W, ow, align, &x0, &x1, &w0, &w1);
                const REAL top = w0 * img[(size_t)y0 * W + x0] + w1 * img[(size_t)y0 * W + x1];
                const REAL bot = w0 * img[(size_t)y1 * W + x0] + w1 * img[(size_t)y1 * W + x1];
                o[(size_t)i * ow + j] = h0 * top + h1 * bot;
            }
        }
    }
}

/* gradient wrt input: scatter form of the adjoint (pad-fold included by construction). */
void KO(ko_filter2d_bwd_input)(const REAL* gy, const REAL* k, REAL* gx, int B, int C, int H, int W, int Bk, int kH,
                               int kW, int border, int same) {
    const int pt = (kH - 1) / 2, pl = (kW - 1) / 2;
    const int Ho = same ? H : H - kH + 1, Wo = same ? W : W - kW + 1;
    memset(gx, 0, (size_t)B * C * H * W * sizeof(REAL));
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            REAL* gi = gx + ((size_t)b * C + c) * H * W;
            const REAL* kk = k + (size_t)(b % Bk) * kH * kW;
            const REAL* go = gy + ((size_t)b * C + c) * Ho * Wo;
            for (int i = 0; i < Ho; ++i)
                for (int j = 0; j < Wo; ++j) {
                    REAL g = go[(size_t)i * Wo + j];
                    for (int p = 0; p < kH; ++p) {
                        long yy = same ? KO(ko_border_index)((long)i + p - pt, H, border) : (long)i + p;
                        if (yy < 0) continue;
                        for (int q = 0; q < kW; ++q) {
                            long xx = same ? KO(ko_border_index)((long)j + q - pl, W, border) : (long)j + q;
                            if (xx < 0) continue;
                            gi[yy * (long)W + xx] += kk[p * kW + q] * g;
                        }
                    }
                }
        }
}

/* gradient wrt the prepared kernel (Bk,kH,kW), accumulated in double. */
void KO(ko_filter2d_bwd_kernel)(const REAL* gy, const REAL* x, REAL* gk, int B, int C, int H, int W, int Bk, int kH,
                                int kW, int border, int same) {
    const int pt = (kH - 1) / 2, pl = (kW - 1) / 2;
    const int Ho = same ? H : H - kH + 1, Wo = same ? W : W - kW + 1;
    double* acc = (double*)calloc((size_t)Bk * kH * kW, sizeof(double));
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const REAL* img = x + ((size_t)b * C + c) * H * W;
            const REAL* go = gy + ((size_t)b * C + c) * Ho * Wo;
            double* a = acc + (size_t)(b % Bk) * kH * kW;
            for (int p = 0; p < kH; ++p)
                for (int q = 0; q < kW; ++q) {
                    double s = 0;
                    for (int i = 0; i < Ho; ++i) {
                        long yy = same ? KO(ko_border_index)((long)i + p - pt, H, border) : (long)i + p;
                        if (yy < 0) continue;
                        for (int j = 0; j < Wo; ++j) {
                            long xx = same ? KO(ko_border_index)((long)j + q - pl, W, border) : (long)j + q;
                            if (xx < 0) continue;
                            s += (double)go[(size_t)i * Wo + j] * (double)img[yy * (long)W + xx];
                        }
                    }
                    a[p * kW + q] += s;
                }
        }
    for (size_t t = 0; t < (size_t)Bk * kH * kW; ++t) gk[t] = (REAL)acc[t];
    free(acc);
}

/* ------------------------------------------------------------------------------------------
 * spatial_gradient / sobel: kornia/filters/sobel.py:59-72, :164-171.
 * kernels (n_out,kS,kS) prepared by the caller (kernels.py:504-529, normalised :68-74);
 * replicate border; out (B,C,n_out,H,W).  magnitude (nullable, n_out == 2): sqrt(gx*gx+gy*gy+eps).
 * ---------------------------------------------------------------------------------------- */
void KO(ko_spatial_gradient_fwd)(const REAL* x, const REAL* kern, REAL* out, REAL* magnitude, int B, int C, int H,
                                 int W, int n_out, int kS, REAL eps) {
    const int pd = kS / 2;
#pragma omp parallel for schedule(static)
    for (int bc = 0; bc < B * C; ++bc) {
        const REAL* img = x + (size_t)bc * H * W;
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                REAL vals[3] = {0, 0, 0};
                for (int o = 0; o < n_out; ++o) {
                    REAL acc = 0;
                    for (int p = 0; p < kS; ++p) {
                        long yy = KO(ko_border_index)((long)i + p - pd, H, 2);
                        for (int q = 0; q < kS; ++q) {
                            long xx = KO(ko_border_index)((long)j + q - pd, W, 2);
                            acc = KO(ko_fma)(kern[(o * kS + p) * kS + q], img[yy * (long)W + xx], acc);
                        }
                    }
                    vals[o] = acc;
                    if (out) out[(((size_t)bc * n_out + o) * H + i) * W + j] = acc;
                }
                if (magnitude) magnitude[((size_t)bc * H + i) * W + j] = KO_SQRT((vals[0] * vals[0] + vals[1] * vals[1]) + eps);
            }
    }
}

/* adjoint of spatial_gradient wrt input: gout (B,C,n_out,H,W) -> gx (B,C,H,W) */
void KO(ko_spatial_gradient_bwd)(const REAL* gout, const REAL* kern, REAL* gx, int B, int C, int H, int W, int n_out,
                                 int kS) {
    const int pd = kS / 2;
    memset(gx, 0, (size_t)B * C * H * W * sizeof(REAL));
#pragma omp parallel for schedule(static)
    for (int bc = 0; bc < B * C; ++bc) {
        REAL* gi = gx + (size_t)bc * H * W;
        for (int o = 0; o < n_out; ++o)
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < W; ++j) {
                    REAL g = gout[(((size_t)bc * n_out + o) * H + i) * W + j];
                    for (int p = 0; p < kS; ++p) {
                        long yy = KO(ko_border_index)((long)i + p - pd, H, 2);
                        for (int q = 0; q < kS; ++q) {
                            long xx = KO(ko_border_index)((long)j + q - pd, W, 2);
                            gi[yy * (long)W + xx] += kern[(o * kS + p) * kS + q] * g;
                        }
                    }
                }
    }
}

#undef KO_FLOOR
#undef KO_FABS
#undef KO_FMOD
#undef KO_NEARBYINT
#undef KO_SQRT
#undef KO
#undef KO_CAT
#undef KO_CAT2
