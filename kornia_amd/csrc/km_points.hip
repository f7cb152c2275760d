// kornia_amd - transform_points for gfx950.
//
// Reference: kornia/geometry/linalg.py:183-239 (repeat_interleave + homogeneous pad + bmm +
// convert_points_from_homogeneous, kornia/geometry/conversions.py:303-307).  The contraction is
// K = D+1 in {3,4}: 2-3 fma per output against 8-12 bytes of traffic, i.e. purely HBM-bound, so this
// is a streaming kernel (one point per lane) and not an MFMA tile; the (B,N,D+1) homogeneous copy
// and the repeated matrices of the reference are never materialised.
// Accumulation = k-ordered fma chain (the BLAS behind torch.bmm; oracle/ko_impl.h ko_transform_points).
#include "km_common.h"

template <typename R, int D>
struct KmPointsArgs {
    const R* T;     // (B_T, D+1, D+1)
    const R* pts;   // (B, N, D)
    R* out;         // fwd: (B,N,D)
    const R* gout;  // bwd: (B,N,D)
    R* gpts;        // bwd: (B,N,D) nullable
    double* gT;     // bwd: (B_T,(D+1)^2) fp64 accumulators, pre-zeroed, nullable
    int B, N, B_T;
    uint32_t blocks_per_batch;
};

template <typename R, int D>
__device__ __forceinline__ void km_point_fwd(const R (&t)[(D + 1) * (D + 1)], const R (&p)[D], R (&hp)[D + 1], R& s, bool& live) {
    constexpr int E = D + 1;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        R acc = p[0] * t[r * E];
#pragma unroll
        for (int k = 1; k < D; ++k) acc = km_fma(p[k], t[r * E + k], acc);
        hp[r] = acc + t[r * E + D];
    }
    const R eps = (R)1e-8;
    live = km_fabs(hp[D]) > eps;
    s = live ? (R)1 / (hp[D] + eps) : (R)1;
}

template <typename R, int D>
__global__ __launch_bounds__(256) void km_transform_points_fwd_kernel(const KmPointsArgs<R, D> a) {
    constexpr int E = D + 1;
    const int b = blockIdx.x / a.blocks_per_batch;
    const int n = (blockIdx.x % a.blocks_per_batch) * 256 + threadIdx.x;
    if (n >= a.N) return;
    R t[E * E];
    const R* tp = a.T + (size_t)(a.B_T == 1 ? 0 : b) * E * E;
#pragma unroll
    for (int k = 0; k < E * E; ++k) t[k] = tp[k];
    R p[D], hp[E], s;
    bool live;
    const size_t off = ((size_t)b * a.N + n) * D;
#pragma unroll
    for (int k = 0; k < D; ++k) p[k] = a.pts[off + k];
    km_point_fwd<R, D>(t, p, hp, s, live);
#pragma unroll
    for (int k = 0; k < D; ++k) a.out[off + k] = s * hp[k];
}

// Vector form of the forward for fp32: a lane owns PPL consecutive points = a whole number of 16-byte words (D = 2: 2 points in
// one float4; D = 3: 4 points in three float4), so a wave moves 1 KB per memory instruction instead of 256 B with a stride -
// the memory pipeline issues about one wave-wide instruction per ~22 cycles per CU whatever its width (profiles/README.md),
// which is what bounded the scalar form at a quarter of the streaming rate.  Same per-point arithmetic (km_point_fwd).
template <int D>
__global__ __launch_bounds__(256) void km_transform_points_fwd_vec_kernel(const KmPointsArgs<float, D> a) {
    constexpr int E = D + 1, PPL = (D == 2) ? 2 : 4, NV = PPL * D / 4;
    const int b = blockIdx.x / a.blocks_per_batch;
    const int n = ((blockIdx.x % a.blocks_per_batch) * 256 + threadIdx.x) * PPL;
    if (n >= a.N) return;
    float t[E * E];
    const float* tp = a.T + (size_t)(a.B_T == 1 ? 0 : b) * E * E;
#pragma unroll
    for (int k = 0; k < E * E; ++k) t[k] = tp[k];
    const size_t off = ((size_t)b * a.N + n) * D;
    float v[NV * 4];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        KM_CHECK_ALIGNED(a.pts + off + 4 * q, 16);
        const float4 w = *reinterpret_cast<const float4*>(a.pts + off + 4 * q);
        v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
    }
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        float p[D], hp[E], s;
        bool live;
#pragma unroll
        for (int k = 0; k < D; ++k) p[k] = v[i * D + k];
        km_point_fwd<float, D>(t, p, hp, s, live);
#pragma unroll
        for (int k = 0; k < D; ++k) v[i * D + k] = s * hp[k];
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        KM_CHECK_ALIGNED(a.out + off + 4 * q, 16);
#ifdef KM_NT_ST
        typedef float km_f4v __attribute__((ext_vector_type(4)));
        km_f4v vv; vv.x = v[4 * q]; vv.y = v[4 * q + 1]; vv.z = v[4 * q + 2]; vv.w = v[4 * q + 3];
        __builtin_nontemporal_store(vv, reinterpret_cast<km_f4v*>(a.out + off + 4 * q));
#else
        *reinterpret_cast<float4*>(a.out + off + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
#endif
    }
}

// out_k = s * h_k, s = 1/(h_D + eps) (live) ; dL/dh_k = g_k s ; dL/dh_D = -s^2 sum_k g_k h_k (live)
// dL/dp_j = sum_r dL/dh_r T[r][j] ; dL/dT[r][j] = sum_n dL/dh_r p_j (p_D = 1)
template <typename R, int D>
__global__ __launch_bounds__(256) void km_transform_points_bwd_kernel(const KmPointsArgs<R, D> a) {
    constexpr int E = D + 1;
    __shared__ double red[4][E * E];
    const int b = blockIdx.x / a.blocks_per_batch;
    const int n = (blockIdx.x % a.blocks_per_batch) * 256 + threadIdx.x;
    R t[E * E];
    const R* tp = a.T + (size_t)(a.B_T == 1 ? 0 : b) * E * E;
#pragma unroll
    for (int k = 0; k < E * E; ++k) t[k] = tp[k];
    R gT[E * E];
#pragma unroll
    for (int k = 0; k < E * E; ++k) gT[k] = 0;
    if (n < a.N) {
        R p[E], hp[E], s, g[D], gh[E];
        bool live;
        const size_t off = ((size_t)b * a.N + n) * D;
#pragma unroll
        for (int k = 0; k < D; ++k) { p[k] = a.pts[off + k]; g[k] = a.gout[off + k]; }
        p[D] = (R)1;
        R pd[D];
#pragma unroll
        for (int k = 0; k < D; ++k) pd[k] = p[k];
        km_point_fwd<R, D>(t, pd, hp, s, live);
        R dot = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) { gh[k] = g[k] * s; dot += g[k] * hp[k]; }
        gh[D] = live ? -(s * s) * dot : (R)0;
        if (a.gpts) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                R acc = 0;
#pragma unroll
                for (int r = 0; r < E; ++r) acc += gh[r] * t[r * E + j];
                a.gpts[off + j] = acc;
            }
        }
#pragma unroll
        for (int r = 0; r < E; ++r)
#pragma unroll
            for (int j = 0; j < E; ++j) gT[r * E + j] = gh[r] * p[j];
    }
    if (a.gT) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < E * E; ++k) {
            const double v = km_wave_sum((double)gT[k]);
            if (lane == 0) red[wave][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < E * E)
            km_atomic_add(a.gT + (size_t)(a.B_T == 1 ? 0 : b) * E * E + threadIdx.x,
                          (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
    }
}

template <typename R, int D>
static int km_points_run(bool bwd, const void* T, const void* pts, void* out, const void* gout, void* gpts, double* gT, int B,
                         int N, int B_T, hipStream_t s) {
    KmPointsArgs<R, D> a;
    a.T = (const R*)T; a.pts = (const R*)pts; a.out = (R*)out; a.gout = (const R*)gout; a.gpts = (R*)gpts; a.gT = gT;
    a.B = B; a.N = N; a.B_T = B_T;
    a.blocks_per_batch = (uint32_t)((N + 255) / 256);
    const uint64_t nb = (uint64_t)a.blocks_per_batch * (uint64_t)B;
    KM_REQUIRE(nb < (1ull << 31), "km_transform_points: grid too large");
    if (nb == 0) return 0;
    if (bwd)
        hipLaunchKernelGGL((km_transform_points_bwd_kernel<R, D>), dim3((uint32_t)nb), dim3(256), 0, s, a);
    else {
        if constexpr (sizeof(R) == sizeof(float)) {
            constexpr int PPL = (D == 2) ? 2 : 4;
            if ((N % PPL) == 0 && ((uintptr_t)pts % 16) == 0 && ((uintptr_t)out % 16) == 0) {
                KmPointsArgs<float, D> v = a;
                v.blocks_per_batch = (uint32_t)((N / PPL + 255) / 256);
                hipLaunchKernelGGL((km_transform_points_fwd_vec_kernel<D>), dim3(v.blocks_per_batch * (uint32_t)B), dim3(256), 0, s, v);
                return km_check_launch("km_transform_points_fwd");
            }
        }
        hipLaunchKernelGGL((km_transform_points_fwd_kernel<R, D>), dim3((uint32_t)nb), dim3(256), 0, s, a);
    }
    return km_check_launch(bwd ? "km_transform_points_bwd" : "km_transform_points_fwd");
}

static int km_points_validate(const char* fn, int B, int N, int D, int B_T, int dtype) {
    KM_REQUIRE(B >= 0 && N >= 0, "%s: bad shape", fn);
    KM_REQUIRE(D == 2 || D == 3, "%s: D must be 2 or 3, got %d", fn, D);
    KM_REQUIRE(B_T == 1 || B_T == B, "%s: transform batch %d must be 1 or %d", fn, B_T, B);
    KM_REQUIRE(dtype == KM_F32 || dtype == KM_F64, "%s: dtype must be f32/f64", fn);
    return 0;
}

extern "C" {

// T (B_T,D+1,D+1), pts (B,N,D) -> out (B,N,D); dtype KM_F32 | KM_F64.
int km_transform_points_fwd(const void* T, const void* pts, void* out, int B, int N, int D, int B_T, int dtype, void* stream) {
    if (B == 0 || N == 0) return 0;
    if (km_points_validate("km_transform_points_fwd", B, N, D, B_T, dtype)) return -1;
    KM_REQUIRE(T && pts && out, "km_transform_points_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KM_F32) return D == 2 ? km_points_run<float, 2>(false, T, pts, out, nullptr, nullptr, nullptr, B, N, B_T, s)
                                       : km_points_run<float, 3>(false, T, pts, out, nullptr, nullptr, nullptr, B, N, B_T, s);
    return D == 2 ? km_points_run<double, 2>(false, T, pts, out, nullptr, nullptr, nullptr, B, N, B_T, s)
                  : km_points_run<double, 3>(false, T, pts, out, nullptr, nullptr, nullptr, B, N, B_T, s);
}

// gpts (B,N,D) nullable; gT (B_T,(D+1)^2) fp64 accumulators, pre-zeroed, nullable.
int km_transform_points_bwd(const void* gout, const void* T, const void* pts, void* gpts, void* gT, int B, int N, int D,
                            int B_T, int dtype, void* stream) {
    if (B == 0 || N == 0) return 0;
    if (km_points_validate("km_transform_points_bwd", B, N, D, B_T, dtype)) return -1;
    KM_REQUIRE(gout && T && pts, "km_transform_points_bwd: null pointer");
    if (!gpts && !gT) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KM_F32) return D == 2 ? km_points_run<float, 2>(true, T, pts, nullptr, gout, gpts, (double*)gT, B, N, B_T, s)
                                       : km_points_run<float, 3>(true, T, pts, nullptr, gout, gpts, (double*)gT, B, N, B_T, s);
    return D == 2 ? km_points_run<double, 2>(true, T, pts, nullptr, gout, gpts, (double*)gT, B, N, B_T, s)
                  : km_points_run<double, 3>(true, T, pts, nullptr, gout, gpts, (double*)gT, B, N, B_T, s);
}

}  // extern "C"
