// kornia_amd - image pyramid kernels (reference: kornia/geometry/transform/pyramid.py).
//
//   km_pyrdown_fwd          pyrdown (pyramid.py:409-453) in ONE pass:  blur with the fixed 5x5 binomial kernel / 256
//                           (filter2d, any border mode) followed by the bilinear resize to (oh, ow).  The blurred image is
//                           never written: HBM traffic = read x once + write y once = (1 + 1/f^2) e bytes per input
//                           element instead of (3 + 1/f^2) e for blur + resize as two kernels (f = 2: 1.25 e vs 3.25 e).
//   km_resize_bilinear_fwd  F.interpolate(mode='bilinear') with ATen's arithmetic (the resize leg of pyrup, :494-496).
//
// Arithmetic (bit-identical to oracle/ko_impl.h ko_filter2d_fwd + ko_resize_bilinear_fwd):
//   * blur: fma chain over the 5x5 taps in (p, q) order from 0, taps r[p]*r[q]/256 (exact in binary floating point);
//     the blurred value is rounded to the storage dtype before it is interpolated, as the unfused pipeline does;
//   * resize: scale = align ? (in-1)/(out-1) : in/out ; src = align ? scale*d : max(scale*(d+0.5)-0.5, 0) ;
//     i0 = (int)src ; i1 = i0 + (i0 < in-1) ; l1 = src - i0 ; l0 = 1 - l1 ;
//     out = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)   (no contraction: the library is built with -ffp-contract=off).
//
// Mapping: one lane per OUTPUT pixel, a wave covers 64 adjacent output columns (for f = 2 its 6 loads per window row
// cover one contiguous 520-byte span of the input row; the 6x6 window overlaps between neighbours are served by L1/L2),
// a 256-thread workgroup covers a 64 x 4 output tile, tiles of one image stay on one XCD (km_xcd_remap).
#include "km_common.h"

enum { KMP_CONSTANT = 0, KMP_REFLECT = 1, KMP_REPLICATE = 2, KMP_CIRCULAR = 3 };

__device__ __forceinline__ int kmp_map(int s, int n, int border) {
    if (s >= 0 && s < n) return s;
    switch (border) {
        case KMP_REFLECT:
            if (s < 0) s = -s;
            if (s >= n) s = 2 * (n - 1) - s;
            return (s >= 0 && s < n) ? s : -1;
        case KMP_REPLICATE: return s < 0 ? 0 : n - 1;
        case KMP_CIRCULAR: { int r = s % n; return r < 0 ? r + n : r; }
        default: return -1;
    }
}

// value as the storage dtype would hold it
__device__ __forceinline__ float kmp_round(float v, const float*) { return v; }
__device__ __forceinline__ double kmp_round(double v, const double*) { return v; }
__device__ __forceinline__ float kmp_round(float v, const km_bf16*) { return __uint_as_float(((uint32_t)km_f32_to_bf16_bits(v)) << 16); }
__device__ __forceinline__ float kmp_round(float v, const km_f16*) { return (float)(km_f16)v; }

// ATen area_pixel_compute_scale / area_pixel_compute_source_index (bilinear: negative sources clamp to 0)
template <typename R>
__device__ __forceinline__ void kmp_axis(int d, int n_in, int n_out, int align, int& i0, int& i1, R& l0, R& l1) {
    R src;
    if (align) {
        const R scale = n_out > 1 ? (R)(n_in - 1) / (R)(n_out - 1) : (R)0;
        src = scale * (R)d;
    } else {
        const R scale = (R)n_in / (R)n_out;
        src = scale * ((R)d + (R)0.5) - (R)0.5;
        if (src < (R)0) src = (R)0;
    }
    i0 = (int)src;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = src - (R)i0;
    l0 = (R)1 - l1;
}

template <typename T>
struct KmPyrArgs {
    const T* x;
    T* y;
    int H, W, oh, ow, border, align;
    uint32_t tiles_x, tiles_y, nblocks;
};

template <typename T>
__device__ __forceinline__ bool kmp_tile(const KmPyrArgs<T>& a, int& ox, int& oy, uint32_t& bc) {
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tbx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t tby = bid % a.tiles_y;
    bc = bid / a.tiles_y;
    ox = (int)tbx * 64 + (int)(threadIdx.x & 63);
    oy = (int)tby * 4 + (int)(threadIdx.x >> 6);
    return ox < a.ow && oy < a.oh;
}

template <typename T>
__global__ __launch_bounds__(256) void km_pyrdown_kernel(const KmPyrArgs<T> a) {
    typedef typename KmTraits<T>::R R;
    int ox, oy;
    uint32_t bc;
    if (!kmp_tile(a, ox, oy, bc)) return;
    const int H = a.H, W = a.W;
    const T* img = a.x + (size_t)bc * H * W;

    int y0, y1, x0, x1;
    R h0, h1, w0, w1;
    kmp_axis<R>(oy, H, a.oh, a.align, y0, y1, h0, h1);
    kmp_axis<R>(ox, W, a.ow, a.align, x0, x1, w0, w1);
    const int dy = y1 - y0, dx = x1 - x0;  // 0 or 1

    // 6 x 6 input window: rows y0-2 .. y0+3, columns x0-2 .. x0+3, through the border map (-1: zero padding)
    int cm[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) cm[c] = kmp_map(x0 - 2 + c, W, a.border);
    R v[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int rm = kmp_map(y0 - 2 + r, H, a.border);
        const T* rowp = img + (size_t)(rm < 0 ? 0 : rm) * W;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const R val = (R)km_ld(rowp + (cm[c] < 0 ? 0 : cm[c]));
            v[r][c] = (rm >= 0 && cm[c] >= 0) ? val : (R)0;
        }
    }

    // the second blurred column / row sits 0 or 1 pixel further (0 only on the last column / row): shift the window
    // once with selects instead of indexing registers dynamically
    R vc[6][5];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int q = 0; q < 5; ++q) vc[r][q] = dx ? v[r][q + 1] : v[r][q];
    const R taps[5] = {(R)1, (R)4, (R)6, (R)4, (R)1};
    R s00 = (R)0, s01 = (R)0, s10 = (R)0, s11 = (R)0;
#pragma unroll
    for (int p = 0; p < 5; ++p)
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const R k = (taps[p] * taps[q]) / (R)256;
            s00 = km_fma(k, v[p][q], s00);
            s01 = km_fma(k, vc[p][q], s01);
            s10 = km_fma(k, dy ? v[p + 1][q] : v[p][q], s10);
            s11 = km_fma(k, dy ? vc[p + 1][q] : vc[p][q], s11);
        }
    R blur[2][2];
    blur[0][0] = kmp_round(s00, (const T*)nullptr); blur[0][1] = kmp_round(s01, (const T*)nullptr);
    blur[1][0] = kmp_round(s10, (const T*)nullptr); blur[1][1] = kmp_round(s11, (const T*)nullptr);
    const R top = w0 * blur[0][0] + w1 * blur[0][1];
    const R bot = w0 * blur[1][0] + w1 * blur[1][1];
    km_st(a.y + ((size_t)bc * a.oh + oy) * a.ow + ox, h0 * top + h1 * bot);
}

template <typename T>
__global__ __launch_bounds__(256) void km_resize_bilinear_kernel(const KmPyrArgs<T> a) {
    typedef typename KmTraits<T>::R R;
    int ox, oy;
    uint32_t bc;
    if (!kmp_tile(a, ox, oy, bc)) return;
    const int H = a.H, W = a.W;
    const T* img = a.x + (size_t)bc * H * W;
    int y0, y1, x0, x1;
    R h0, h1, w0, w1;
    kmp_axis<R>(oy, H, a.oh, a.align, y0, y1, h0, h1);
    kmp_axis<R>(ox, W, a.ow, a.align, x0, x1, w0, w1);
    const T* r0 = img + (size_t)y0 * W;
    const T* r1 = img + (size_t)y1 * W;
    const R v00 = (R)km_ld(r0 + x0), v01 = (R)km_ld(r0 + x1), v10 = (R)km_ld(r1 + x0), v11 = (R)km_ld(r1 + x1);
    const R top = w0 * v00 + w1 * v01;
    const R bot = w0 * v10 + w1 * v11;
    km_st(a.y + ((size_t)bc * a.oh + oy) * a.ow + ox, h0 * top + h1 * bot);
}

template <typename T>
static int kmp_run(bool blur, const void* x, void* y, int B, int C, int H, int W, int oh, int ow, int border, int align, hipStream_t s) {
    KmPyrArgs<T> a;
    a.x = (const T*)x; a.y = (T*)y;
    a.H = H; a.W = W; a.oh = oh; a.ow = ow; a.border = border; a.align = align;
    a.tiles_x = (uint32_t)((ow + 63) / 64);
    a.tiles_y = (uint32_t)((oh + 3) / 4);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B * (uint64_t)C;
    KM_REQUIRE(nb < (1ull << 31), "km_pyramid: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    if (blur) {
        hipLaunchKernelGGL((km_pyrdown_kernel<T>), dim3(a.nblocks), dim3(256), 0, s, a);
        return km_check_launch("km_pyrdown_fwd");
    }
    hipLaunchKernelGGL((km_resize_bilinear_kernel<T>), dim3(a.nblocks), dim3(256), 0, s, a);
    return km_check_launch("km_resize_bilinear_fwd");
}

static int kmp_validate(const char* what, const void* x, const void* y, int B, int C, int H, int W, int oh, int ow, int dtype) {
    KM_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && oh >= 0 && ow >= 0, "%s: bad shape B=%d C=%d H=%d W=%d -> %dx%d", what, B, C, H, W, oh, ow);
    KM_REQUIRE(dtype >= KM_F32 && dtype <= KM_F16, "%s: unknown dtype code %d", what, dtype);
    if ((uint64_t)B * C * oh * ow == 0) return 0;
    KM_REQUIRE(x && y, "%s: null pointer", what);
    return 0;
}

static int kmp_dispatch(bool blur, const void* x, void* y, int B, int C, int H, int W, int oh, int ow, int border, int align, int dtype, hipStream_t s) {
    switch (dtype) {
        case KM_F32: return kmp_run<float>(blur, x, y, B, C, H, W, oh, ow, border, align, s);
        case KM_F64: return kmp_run<double>(blur, x, y, B, C, H, W, oh, ow, border, align, s);
        case KM_BF16: return kmp_run<km_bf16>(blur, x, y, B, C, H, W, oh, ow, border, align, s);
        default: return kmp_run<km_f16>(blur, x, y, B, C, H, W, oh, ow, border, align, s);
    }
}

extern "C" {

int km_pyrdown_fwd(const void* x, void* y, int B, int C, int H, int W, int oh, int ow, int border, int align, int dtype, void* stream) {
    if (kmp_validate("km_pyrdown_fwd", x, y, B, C, H, W, oh, ow, dtype)) return -1;
    KM_REQUIRE(border >= KMP_CONSTANT && border <= KMP_CIRCULAR, "km_pyrdown_fwd: unknown border code %d", border);
    // torch's reflection padding needs pad < size (filter.py:139 F.pad): the 5x5 blur pads by 2
    KM_REQUIRE(border != KMP_REFLECT || (H > 2 && W > 2), "km_pyrdown_fwd: reflect padding needs H, W > 2 (got %dx%d)", H, W);
    if ((uint64_t)B * C * oh * ow == 0) return 0;
    return kmp_dispatch(true, x, y, B, C, H, W, oh, ow, border, align ? 1 : 0, dtype, (hipStream_t)stream);
}

int km_resize_bilinear_fwd(const void* x, void* y, int B, int C, int H, int W, int oh, int ow, int align, int dtype, void* stream) {
    if (kmp_validate("km_resize_bilinear_fwd", x, y, B, C, H, W, oh, ow, dtype)) return -1;
    if ((uint64_t)B * C * oh * ow == 0) return 0;
    return kmp_dispatch(false, x, y, B, C, H, W, oh, ow, 0, align ? 1 : 0, dtype, (hipStream_t)stream);
}

}  // extern "C"
