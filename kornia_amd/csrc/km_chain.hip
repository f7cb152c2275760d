// kornia_amd - the batched 3x3 homography chain (normalise + invert) as one tiny launch.
//
// Replaces ~25 PyTorch launches per warp call:
//   normalize_homography      kornia/geometry/conversions.py:1691-1726
//   normal_transform_pixel    kornia/geometry/conversions.py:1729-1763
//   _inverse_3x3_closed_form  kornia/core/utils.py:137-166 (eager branch; called at imgwarp.py:153,254)
//   convert_affinematrix_to_homography  kornia/geometry/conversions.py:342-378 (rows == 2)
// B x ~150 flop: far below anything MFMA could help with - one thread per
// matrix, rounding sequence identical to the reference's CPU ops (oracle/ko_impl.h ko_inv3 /
// ko_mm3, pinned bit-exactly by tests/golden/chain_*.npz).
#include "km_common.h"

template <typename R>
__device__ __forceinline__ void km_cross3(const R (&a)[3], const R (&b)[3], R (&o)[3]) {
    // torch.linalg.cross's kernel contracts a1*b2 - a2*b1 into fma(a1, b2, -(a2*b1))
    R t;
    t = a[2] * b[1];
    o[0] = km_fma(a[1], b[2], -t);
    t = a[0] * b[2];
    o[1] = km_fma(a[2], b[0], -t);
    t = a[1] * b[0];
    o[2] = km_fma(a[0], b[1], -t);
}

template <typename R>
__device__ __forceinline__ void km_inv3(const R (&m)[9], R (&o)[9]) {
    const R a[3] = {m[0], m[3], m[6]}, b[3] = {m[1], m[4], m[7]}, c[3] = {m[2], m[5], m[8]};
    R r0[3], r1[3], r2[3];
    km_cross3(b, c, r0);
    km_cross3(c, a, r1);
    km_cross3(a, b, r2);
    const R p0 = a[0] * r0[0], p1 = a[1] * r0[1], p2 = a[2] * r0[2];
    const R det = (p0 + p1) + p2;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = r0[k] / det;
        o[3 + k] = r1[k] / det;
        o[6 + k] = r2[k] / det;
    }
}

template <typename R>
__device__ __forceinline__ void km_mm3(const R (&a)[9], const R (&b)[9], R (&o)[9]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            R acc = (R)0;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc = acc + a[3 * i + k] * b[3 * k + j];
            o[3 * i + j] = acc;
        }
}

template <typename R>
struct KmChainArgs {
    const R* M;    // (B, rows, 3)
    R* A;          // (B,9) normalised src->dst, nullable
    R* m;          // (B,9) normalised dst->src, nullable
    const double* gm;  // bwd: (B,9) fp64
    R* gM;         // bwd: (B, rows, 3)
    int B, rows;
    float sx_s, sy_s, sx_d, sy_d;  // float32-rounded 2/(W-1), 2/(H-1) of source / destination
};

template <typename R>
__device__ __forceinline__ void km_chain_setup(const KmChainArgs<R>& a, int b, R (&M)[9], R (&Nsi)[9], R (&Nd)[9]) {
    const R* mp = a.M + (size_t)b * a.rows * 3;
#pragma unroll
    for (int k = 0; k < 6; ++k) M[k] = mp[k];
    if (a.rows == 3) {
        M[6] = mp[6]; M[7] = mp[7]; M[8] = mp[8];
    } else {
        M[6] = 0; M[7] = 0; M[8] = 1;
    }
    const R Ns[9] = {(R)a.sx_s, 0, -1, 0, (R)a.sy_s, -1, 0, 0, 1};
    km_inv3(Ns, Nsi);
    const R nd[9] = {(R)a.sx_d, 0, -1, 0, (R)a.sy_d, -1, 0, 0, 1};
#pragma unroll
    for (int k = 0; k < 9; ++k) Nd[k] = nd[k];
}

template <typename R>
__global__ void km_chain_fwd_kernel(const KmChainArgs<R> a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    R M[9], Nsi[9], Nd[9], t[9], A[9], mi[9];
    km_chain_setup(a, b, M, Nsi, Nd);
    km_mm3(M, Nsi, t);
    km_mm3(Nd, t, A);
    if (a.A) {
#pragma unroll
        for (int k = 0; k < 9; ++k) a.A[(size_t)b * 9 + k] = A[k];
    }
    if (a.m) {
        km_inv3(A, mi);
#pragma unroll
        for (int k = 0; k < 9; ++k) a.m[(size_t)b * 9 + k] = mi[k];
    }
}

// adjoint wrt M in fp64: m = inv(A), A = Nd M Nsi ; gA = -m^T gm m^T ; gM = Nd^T gA Nsi^T
template <typename R>
__global__ void km_chain_bwd_kernel(const KmChainArgs<R> a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    R M[9], Nsi[9], Nd[9], t[9], A[9], mi[9];
    km_chain_setup(a, b, M, Nsi, Nd);
    km_mm3(M, Nsi, t);
    km_mm3(Nd, t, A);
    km_inv3(A, mi);
    double g[9], t1[9], gA[9], t2[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = a.gm[(size_t)b * 9 + k];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += (double)mi[3 * k + i] * g[3 * k + j];
            t1[3 * i + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += t1[3 * i + k] * (double)mi[3 * j + k];
            gA[3 * i + j] = -s;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += (double)Nd[3 * k + i] * gA[3 * k + j];
            t2[3 * i + j] = s;
        }
    R* out = a.gM + (size_t)b * a.rows * 3;
    for (int i = 0; i < a.rows; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += t2[3 * i + k] * (double)Nsi[3 * j + k];
            out[3 * i + j] = (R)s;
        }
}

static inline float km_norm_scale(int size) {
    // conversions.py:1750-1755: eps=1e-14 when size == 1; value rounded to float32 by torch.tensor()
    const double d = (size == 1) ? 1e-14 : (double)size - 1.0;
    return (float)(2.0 / d);
}

template <typename R>
static int km_chain_run(bool bwd, const void* M, int rows, void* A, void* m, const double* gm, void* gM, int B, int Hs,
                        int Ws, int hd, int wd, hipStream_t s) {
    KmChainArgs<R> a;
    a.M = (const R*)M; a.A = (R*)A; a.m = (R*)m; a.gm = gm; a.gM = (R*)gM; a.B = B; a.rows = rows;
    a.sx_s = km_norm_scale(Ws); a.sy_s = km_norm_scale(Hs); a.sx_d = km_norm_scale(wd); a.sy_d = km_norm_scale(hd);
    if (B == 0) return 0;
    const int threads = 64, blocks = (B + threads - 1) / threads;
    if (bwd)
        hipLaunchKernelGGL(km_chain_bwd_kernel<R>, dim3(blocks), dim3(threads), 0, s, a);
    else
        hipLaunchKernelGGL(km_chain_fwd_kernel<R>, dim3(blocks), dim3(threads), 0, s, a);
    return km_check_launch(bwd ? "km_homography_chain_bwd" : "km_homography_chain_fwd");
}


// ------------------------------------------------------------------------------------------------
// get_affine_matrix2d (kornia/geometry/transform/imgwarp.py:746-787) in one launch: the matrix RandomAffine builds
// from (translations, center, scale, angle, shear) with ~45 tiny PyTorch launches
// (kornia/augmentation/_2d/geometric/affine.py:125-141 -> get_rotation_matrix2d :529-622 -> angle_to_rotation_matrix
// conversions.py:1652-1688, get_shear_matrix2d :815-869).  One thread per matrix, the reference's own sequence:
//   R = (T(c) @ Rot(-angle)) @ S(scale) @ T(-c), R[:, 2] += t, optionally @ Shear(c, sx, sy); 3x3 products as the
//   k-ordered mul/add chain of km_mm3; degrees -> radians with the float32 pi the reference uses.
template <typename R>
struct KmAffineArgs {
    const R* trans;   // (B,2)
    const R* center;  // (B,2)
    const R* scale;   // (B,2)
    const R* angle;   // (B) degrees, clockwise-positive (negated before the rotation, imgwarp.py:778)
    const R* sx;      // (B) radians, nullable
    const R* sy;      // (B) radians, nullable
    R* out;           // (B,9)
    int B;
};

__device__ __forceinline__ float km_sin(float x) { return sinf(x); }
__device__ __forceinline__ double km_sin(double x) { return sin(x); }
__device__ __forceinline__ float km_cos(float x) { return cosf(x); }
__device__ __forceinline__ double km_cos(double x) { return cos(x); }
__device__ __forceinline__ float km_tan(float x) { return tanf(x); }
__device__ __forceinline__ double km_tan(double x) { return tan(x); }

// the matrix of sample b; SHEAR_DEG: sx / sy are in degrees and converted as RandomAffine.compute_transformation does
// (x * float(pi / 180), kornia/augmentation/_2d/geometric/affine.py:131-133), else radians
template <typename R, bool SHEAR_DEG>
__device__ __forceinline__ void km_affine_matrix_of(const KmAffineArgs<R>& a, int b, R (&m)[9]) {
    const R cx = a.center[2 * b], cy = a.center[2 * b + 1];
    // deg2rad(-angle): tensor * pi.type(dtype) / 180 with pi a float32 constant (conversions.py:148, constants.py:25)
    const R pi32 = (R)3.14159265358979323846f;
    const R rad = ((-a.angle[b]) * pi32) / (R)180.0;
    const R c = km_cos(rad), s = km_sin(rad);
    const R shift[9] = {1, 0, cx, 0, 1, cy, 0, 0, 1};
    const R shift_inv[9] = {1, 0, -cx, 0, 1, -cy, 0, 0, 1};
    const R rot[9] = {c, s, 0, -s, c, 0, 0, 0, 1};
    const R scl[9] = {a.scale[2 * b], 0, 0, 0, a.scale[2 * b + 1], 0, 0, 0, 1};
    R t0[9], t1[9];
    km_mm3(shift, rot, t0);
    km_mm3(t0, scl, t1);
    km_mm3(t1, shift_inv, m);
    m[2] = m[2] + a.trans[2 * b];
    m[5] = m[5] + a.trans[2 * b + 1];
    m[6] = 0; m[7] = 0; m[8] = 1;  // convert_affinematrix_to_homography of the (2,3) block
    if (a.sx || a.sy) {
        const R d2r = (R)(3.14159265358979323846 / 180.0);
        const R ax = a.sx ? (SHEAR_DEG ? a.sx[b] * d2r : a.sx[b]) : (R)0, ay = a.sy ? (SHEAR_DEG ? a.sy[b] * d2r : a.sy[b]) : (R)0;
        const R tx = km_tan(ax), ty = km_tan(ay);
        const R sh[9] = {1, -tx, tx * cy, -ty, (R)1 + tx * ty, ty * (cx - tx * cy), 0, 0, 1};
        R o[9];
        km_mm3(m, sh, o);
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = o[k];
    }
}

template <typename R>
__global__ __launch_bounds__(64) void km_affine_matrix2d_kernel(const KmAffineArgs<R> a) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    R m[9];
    km_affine_matrix_of<R, false>(a, b, m);
#pragma unroll
    for (int k = 0; k < 9; ++k) a.out[(size_t)b * 9 + k] = m[k];
}

// RandomAffine's sampled parameters -> what the warp kernel reads, in ONE launch: the pixel matrix as above (shears in degrees),
// then normalise and invert exactly as km_chain_fwd_kernel does for warp_affine (imgwarp.py:271-284):  m = inv(Nd M inv(Ns)).
__global__ __launch_bounds__(64) void km_affine_params_chain_kernel(const KmAffineArgs<float> a, const KmChainArgs<float> ch, float* M_out,
                                                                    const float* prob, uint8_t* apply) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    if (apply) apply[b] = (prob[b] > 0.5f) ? 1 : 0;  // the per-sample switch of the augmentation layer (base.py:380)
    float M[9];
    km_affine_matrix_of<float, true>(a, b, M);
    if (M_out) {
#pragma unroll
        for (int k = 0; k < 9; ++k) M_out[(size_t)b * 9 + k] = M[k];
    }
    const float Ns[9] = {ch.sx_s, 0, -1, 0, ch.sy_s, -1, 0, 0, 1};
    const float Nd[9] = {ch.sx_d, 0, -1, 0, ch.sy_d, -1, 0, 0, 1};
    float Nsi[9], t[9], A[9], mi[9];
    km_inv3(Ns, Nsi);
    km_mm3(M, Nsi, t);
    km_mm3(Nd, t, A);
    km_inv3(A, mi);
#pragma unroll
    for (int k = 0; k < 9; ++k) ch.m[(size_t)b * 9 + k] = mi[k];
}

template <typename R>
static int km_affine_run(const void* trans, const void* center, const void* scale, const void* angle, const void* sx, const void* sy,
                         void* out, int B, hipStream_t s) {
    KmAffineArgs<R> a;
    a.trans = (const R*)trans; a.center = (const R*)center; a.scale = (const R*)scale; a.angle = (const R*)angle;
    a.sx = (const R*)sx; a.sy = (const R*)sy; a.out = (R*)out; a.B = B;
    hipLaunchKernelGGL(km_affine_matrix2d_kernel<R>, dim3((B + 63) / 64), dim3(64), 0, s, a);
    return km_check_launch("km_affine_matrix2d_fwd");
}

// ------------------------------------------------------------------------------------------------
// get_perspective_transform (kornia/geometry/transform/imgwarp.py:397-525) in one launch: the homography taking four source
// points onto four destination points, H = Q_dst * adj(Q_src) normalised by H[2][2], with Q the unit-square -> quad map of
// Heckbert's closed form ("Fundamentals of Texture Mapping and Image Warping", 1989, ch. 2).  One thread per matrix;
// same formulas as kornia_amd/geometry/transform/builders.py (the differentiable tensor expression).
template <typename R>
__device__ __forceinline__ void km_square_to_quad(const R* q, R (&o)[8]) {
    const R x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3], x2 = q[4], y2 = q[5], x3 = q[6], y3 = q[7];
    const R ex1 = x1 - x2, ex2 = x3 - x2, ey1 = y1 - y2, ey2 = y3 - y2;
    const R sx = (x0 - x1) + (x2 - x3), sy = (y0 - y1) + (y2 - y3);
    const R det = ex1 * ey2 - ex2 * ey1;
    const R g = (sx * ey2 - ex2 * sy) / det, h = (ex1 * sy - sx * ey1) / det;
    o[0] = (x1 - x0) + g * x1; o[1] = (x3 - x0) + h * x3; o[2] = x0;
    o[3] = (y1 - y0) + g * y1; o[4] = (y3 - y0) + h * y3; o[5] = y0;
    o[6] = g; o[7] = h;
}

template <typename R>
__global__ __launch_bounds__(64) void km_perspective_transform_kernel(const R* src, const R* dst, R* out, int B) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    R s[8], d[8];
    km_square_to_quad<R>(src + (size_t)b * 8, s);
    km_square_to_quad<R>(dst + (size_t)b * 8, d);
    const R a = s[0], bb = s[1], c = s[2], dd = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
    // adjugate of [[a,b,c],[d,e,f],[g,h,1]]
    const R j00 = e - f * h, j01 = c * h - bb, j02 = bb * f - c * e;
    const R j10 = f * g - dd, j11 = a - c * g, j12 = c * dd - a * f;
    const R j20 = dd * h - e * g, j21 = bb * g - a * h, j22 = a * e - bb * dd;
    R m[9];
    m[0] = d[0] * j00 + d[1] * j10 + d[2] * j20; m[1] = d[0] * j01 + d[1] * j11 + d[2] * j21; m[2] = d[0] * j02 + d[1] * j12 + d[2] * j22;
    m[3] = d[3] * j00 + d[4] * j10 + d[5] * j20; m[4] = d[3] * j01 + d[4] * j11 + d[5] * j21; m[5] = d[3] * j02 + d[4] * j12 + d[5] * j22;
    m[6] = d[6] * j00 + d[7] * j10 + j20; m[7] = d[6] * j01 + d[7] * j11 + j21; m[8] = d[6] * j02 + d[7] * j12 + j22;
#pragma unroll
    for (int k = 0; k < 9; ++k) out[(size_t)b * 9 + k] = m[k] / m[8];
}

extern "C" {

// M: (B,rows,3) pixel src->dst matrix, rows in {2,3}; A_out/m_out: (B,9), either may be null.
// dtype: KM_F32 or KM_F64 (matrices of bf16/f16 images are handled in fp32 by the caller).
int km_homography_chain_fwd(const void* M, int rows, void* A_out, void* m_out, int B, int Hs, int Ws, int hd, int wd,
                            int dtype, void* stream) {
    if (B == 0) return 0;  // empty batch: nothing to do (data pointers of empty tensors are null)
    KM_REQUIRE(M && (A_out || m_out), "km_homography_chain_fwd: null pointer");
    KM_REQUIRE(rows == 2 || rows == 3, "km_homography_chain_fwd: rows must be 2 or 3, got %d", rows);
    KM_REQUIRE(B >= 0 && Hs > 0 && Ws > 0 && hd > 0 && wd > 0, "km_homography_chain_fwd: bad sizes");
    KM_REQUIRE(dtype == KM_F32 || dtype == KM_F64, "km_homography_chain_fwd: dtype must be f32/f64");
    if (dtype == KM_F32) return km_chain_run<float>(false, M, rows, A_out, m_out, nullptr, nullptr, B, Hs, Ws, hd, wd, (hipStream_t)stream);
    return km_chain_run<double>(false, M, rows, A_out, m_out, nullptr, nullptr, B, Hs, Ws, hd, wd, (hipStream_t)stream);
}

// gm: (B,9) fp64 gradient wrt m_out; gM: (B,rows,3) in dtype.
int km_homography_chain_bwd(const void* M, int rows, const double* gm, void* gM, int B, int Hs, int Ws, int hd, int wd,
                            int dtype, void* stream) {
    if (B == 0) return 0;
    KM_REQUIRE(M && gm && gM, "km_homography_chain_bwd: null pointer");
    KM_REQUIRE(rows == 2 || rows == 3, "km_homography_chain_bwd: rows must be 2 or 3, got %d", rows);
    KM_REQUIRE(dtype == KM_F32 || dtype == KM_F64, "km_homography_chain_bwd: dtype must be f32/f64");
    if (dtype == KM_F32) return km_chain_run<float>(true, M, rows, nullptr, nullptr, gm, gM, B, Hs, Ws, hd, wd, (hipStream_t)stream);
    return km_chain_run<double>(true, M, rows, nullptr, nullptr, gm, gM, B, Hs, Ws, hd, wd, (hipStream_t)stream);
}

// translations / center / scale (B,2), angle (B) degrees, sx / sy (B) radians or null -> out (B,3,3); dtype f32 / f64.
int km_affine_matrix2d_fwd(const void* translations, const void* center, const void* scale, const void* angle, const void* sx,
                           const void* sy, void* out, int B, int dtype, void* stream) {
    if (B == 0) return 0;
    KM_REQUIRE(translations && center && scale && angle && out, "km_affine_matrix2d_fwd: null pointer");
    KM_REQUIRE(B > 0, "km_affine_matrix2d_fwd: bad batch size %d", B);
    KM_REQUIRE(dtype == KM_F32 || dtype == KM_F64, "km_affine_matrix2d_fwd: dtype must be f32/f64");
    if (dtype == KM_F32) return km_affine_run<float>(translations, center, scale, angle, sx, sy, out, B, (hipStream_t)stream);
    return km_affine_run<double>(translations, center, scale, angle, sx, sy, out, B, (hipStream_t)stream);
}

// RandomAffine.compute_transformation + the normalise / invert chain of warp_affine in one launch (float32):
// translations / center / scale (B,2), angle (B) degrees, shear_x / shear_y (B) DEGREES or null ->
// M_out (B,9) pixel src->dst matrix (nullable) and m_out (B,9), the normalised dst->src matrix km_warp2d_fwd reads
// (coord_mode = affine) for a (Hs,Ws) source and a (hd,wd) destination; batch_prob (B) fp32 or null -> apply (B) uint8 = batch_prob > 0.5.
// Reference: kornia/augmentation/_2d/geometric/affine.py:125-141 -> imgwarp.py:746-787, then imgwarp.py:271-284; base.py:380.
int km_affine_params_chain_fwd(const void* translations, const void* center, const void* scale, const void* angle, const void* shear_x,
                               const void* shear_y, const void* batch_prob, void* M_out, void* m_out, void* apply, int B, int Hs, int Ws, int hd, int wd,
                               void* stream) {
    if (B == 0) return 0;
    KM_REQUIRE(translations && center && scale && angle && m_out, "km_affine_params_chain_fwd: null pointer");
    KM_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && hd > 0 && wd > 0, "km_affine_params_chain_fwd: bad sizes");
    KM_REQUIRE((batch_prob == nullptr) == (apply == nullptr), "km_affine_params_chain_fwd: batch_prob and apply go together");
    KmAffineArgs<float> a;
    a.trans = (const float*)translations; a.center = (const float*)center; a.scale = (const float*)scale; a.angle = (const float*)angle;
    a.sx = (const float*)shear_x; a.sy = (const float*)shear_y; a.out = nullptr; a.B = B;
    KmChainArgs<float> ch;
    ch.M = nullptr; ch.A = nullptr; ch.m = (float*)m_out; ch.gm = nullptr; ch.gM = nullptr; ch.B = B; ch.rows = 3;
    ch.sx_s = km_norm_scale(Ws); ch.sy_s = km_norm_scale(Hs); ch.sx_d = km_norm_scale(wd); ch.sy_d = km_norm_scale(hd);
    hipLaunchKernelGGL(km_affine_params_chain_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, a, ch, (float*)M_out, (const float*)batch_prob,
                       (uint8_t*)apply);
    return km_check_launch("km_affine_params_chain_fwd");
}

// points_src / points_dst (B,4,2) -> out (B,3,3), H[2][2] == 1; dtype f32 / f64.
int km_perspective_transform_fwd(const void* points_src, const void* points_dst, void* out, int B, int dtype, void* stream) {
    if (B == 0) return 0;
    KM_REQUIRE(points_src && points_dst && out, "km_perspective_transform_fwd: null pointer");
    KM_REQUIRE(B > 0, "km_perspective_transform_fwd: bad batch size %d", B);
    KM_REQUIRE(dtype == KM_F32 || dtype == KM_F64, "km_perspective_transform_fwd: dtype must be f32/f64");
    if (dtype == KM_F32)
        hipLaunchKernelGGL(km_perspective_transform_kernel<float>, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const float*)points_src,
                           (const float*)points_dst, (float*)out, B);
    else
        hipLaunchKernelGGL(km_perspective_transform_kernel<double>, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const double*)points_src,
                           (const double*)points_dst, (double*)out, B);
    return km_check_launch("km_perspective_transform_fwd");
}

}  // extern "C"
