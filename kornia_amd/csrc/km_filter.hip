// kornia_amd - filter2d / filter2d_separable kernels for gfx950.
//
// Reference semantics (kornia/filters/filter.py:54-207, _compute_padding :31-51):
//   y[b,c,i,j] = sum_{p,q} k[b % Bk][p][q] * xpad[b,c,i+p,j+q]        (cross-correlation)
//   xpad = F.pad(x, [l, r, t, b], mode), l=(kW-1)//2, r=kW-1-l, t=(kH-1)//2, b=kH-1-t ('same'),
//   no padding and a shrunken output for 'valid'.  Separable = (1 x kW) pass then (kH x 1) pass,
//   the intermediate stored in the input dtype.
// The padded copy is never materialised: border handling is an index map applied while staging.
// Accumulation is an fma chain in (p, q) order from 0, identical to the CPU oracle
// (oracle/ko_impl.h ko_filter2d_fwd), so fp32 results are bit-identical to it.
//
// Kernels:
//   km_filter_sep_fwd_kernel      fused row+column pass through LDS (the GaussianBlur2d hot path):
//                                 HBM traffic = read x once (+halo out of L2), write y once.
//   km_filter_sep_bwd_kernel      adjoint of the above (column-adjoint then row-adjoint, with the
//                                 reflect/replicate pad fold), same traffic.
//   km_filter2d_fwd_kernel        generic kH x kW (any size / border / Bk), direct loads.
//   km_filter2d_bwd_input_kernel  generic adjoint, gather form (no atomics, deterministic).
//   km_filter2d_bwd_kernel_kernel gradient wrt the taps (fp64 accumulation).
#include <stdlib.h>

#include "km_regtile.h"


struct KmFilterGeom {
    int B, C, H, W, Bk, kH, kW, border, same;
    int Ho, Wo;   // output size
    int pt, pl;   // top / left padding ('same'), 0 for 'valid'
};

static KmFilterGeom km_filter_geom(int B, int C, int H, int W, int Bk, int kH, int kW, int border, int same) {
    KmFilterGeom g;
    g.B = B; g.C = C; g.H = H; g.W = W; g.Bk = Bk; g.kH = kH; g.kW = kW; g.border = border; g.same = same;
    g.Ho = same ? H : H - kH + 1;
    g.Wo = same ? W : W - kW + 1;
    g.pt = same ? (kH - 1) / 2 : 0;
    g.pl = same ? (kW - 1) / 2 : 0;
    return g;
}

// =================================================================================================
// fused separable forward
// =================================================================================================
#define KM_FS_TW 64
#define KM_FS_TH 32

template <typename T>
struct KmSepArgs {
    typedef typename KmTraits<T>::R R;
    const T* x;   // fwd: input (B,C,H,W) ; bwd: grad_out (B,C,Ho,Wo)
    T* y;         // fwd: output (B,C,Ho,Wo) ; bwd: grad_in (B,C,H,W)
    const R* kx;  // (Bk,kW)
    const R* ky;  // (Bk,kH)
    KmFilterGeom g;
    uint32_t tiles_x, tiles_y, nblocks;
};

// round an intermediate to the storage dtype (filter2d_separable stores out_x in input.dtype)
__device__ __forceinline__ float km_round_to(float v, const float*) { return v; }
__device__ __forceinline__ double km_round_to(double v, const double*) { return v; }
__device__ __forceinline__ float km_round_to(float v, const km_bf16*) { return __uint_as_float(((uint32_t)km_f32_to_bf16_bits(v)) << 16); }
__device__ __forceinline__ float km_round_to(float v, const km_f16*) { KM_OPAQUE(v); return (float)(km_f16)v; }

template <typename T>
__global__ __launch_bounds__(256) void km_filter_sep_fwd_kernel(const KmSepArgs<T> a) {
    typedef typename KmTraits<T>::R R;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const KmFilterGeom& g = a.g;
    const int IW = KM_FS_TW + g.kW - 1;  // staged input width
    const int IH = KM_FS_TH + g.kH - 1;  // staged input height
    R* s_in = (R*)smem_raw;              // [IH][IW]
    R* s_tmp = s_in + IH * IW;           // [IH][KM_FS_TW]
    R* s_kx = s_tmp + IH * KM_FS_TW;     // [kW]
    R* s_ky = s_kx + g.kW;               // [kH]

    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int b = bc / g.C;
    const int x0 = tx * KM_FS_TW, y0 = ty * KM_FS_TH;
    const T* img = a.x + (size_t)bc * g.H * g.W;
    T* out = a.y + (size_t)bc * g.Ho * g.Wo;
    const int tid = threadIdx.x;

    const R* kxp = a.kx + (size_t)(b % g.Bk) * g.kW;
    const R* kyp = a.ky + (size_t)(b % g.Bk) * g.kH;
    for (int t = tid; t < g.kW; t += 256) s_kx[t] = kxp[t];
    for (int t = tid; t < g.kH; t += 256) s_ky[t] = kyp[t];

    // stage the input tile (border applied as an index map; columns fastest => coalesced)
    for (int e = tid; e < IH * IW; e += 256) {
        const int r = e / IW, c = e - r * IW;
        int sy = y0 + r - g.pt, sx = x0 + c - g.pl;
        if (g.same) {
            sy = km_border_map(sy, g.H, g.border);
            sx = km_border_map(sx, g.W, g.border);
        } else {
            if (sy >= g.H) sy = -1;
            if (sx >= g.W) sx = -1;
        }
        s_in[e] = (sy >= 0 && sx >= 0) ? (R)km_ld(img + (size_t)sy * g.W + sx) : (R)0;
    }
    __syncthreads();

    // row pass: tmp[r][c] = sum_q kx[q] * in[r][c+q]
    for (int e = tid; e < IH * KM_FS_TW; e += 256) {
        const int r = e / KM_FS_TW, c = e - r * KM_FS_TW;
        const R* row = s_in + r * IW + c;
        R acc = 0;
        for (int q = 0; q < g.kW; ++q) acc = km_fma(s_kx[q], row[q], acc);
        s_tmp[e] = km_round_to(acc, (const T*)nullptr);
    }
    __syncthreads();

    // column pass: out[r][c] = sum_p ky[p] * tmp[r+p][c]
    for (int e = tid; e < KM_FS_TH * KM_FS_TW; e += 256) {
        const int r = e / KM_FS_TW, c = e - r * KM_FS_TW;
        const int oy = y0 + r, ox = x0 + c;
        if (oy < g.Ho && ox < g.Wo) {
            R acc = 0;
            for (int p = 0; p < g.kH; ++p) acc = km_fma(s_ky[p], s_tmp[(r + p) * KM_FS_TW + c], acc);
            km_st(out + (size_t)oy * g.Wo + ox, acc);
        }
    }
}

// =================================================================================================
// fused separable backward (gradient wrt input)
//
// forward (per axis, 'same'):  y[i] = sum_t k[t] * x[map(i + t - l)]
// adjoint:                     gx[p] = sum_{s in map^-1(p)} G[s],  G[s] = sum_t k[t] * gy0[s + l - t]
// with gy0 = gy zero-extended, s ranging over the padded axis [-l, n-1+r].  map^-1(p) is {p} plus,
// near an image border, the pad positions that the border mode folds onto p (reflect: s = -p and
// s = 2(n-1)-p; replicate: every s < 0 onto 0, every s >= n onto n-1; constant: nothing).  Circular
// is periodic, so its adjoint is the periodic correlation and needs no fold.
// The 2-D adjoint = column adjoint (on gy) followed by row adjoint.
// The block computes G on its tile extended by the pad on every side, so that a border tile finds
// its fold sources in LDS (host guarantees tile >= pad, else the generic kernel is used).
// =================================================================================================
__device__ __forceinline__ int km_zero_index(int s, int n, int border) {
    // index into gy (zero-extended, or periodic for circular)
    if (border == KM_BORDER_CIRCULAR) {
        int r = s % n;
        return r < 0 ? r + n : r;
    }
    return (s >= 0 && s < n) ? s : -1;
}

// Sum G over the fold pre-images of p (excluding p itself); G is addressed as Gline[(s - base) * stride]
template <typename R>
__device__ __forceinline__ R km_fold_sum(const R* Gline, int stride, int base, int p, int n, int l, int r, int border) {
    R acc = 0;
    if (border == KM_BORDER_REFLECT) {
        if (p >= 1 && p <= l) acc += Gline[(-p - base) * stride];
        if (p <= n - 2 && p >= n - 1 - r) acc += Gline[(2 * (n - 1) - p - base) * stride];
    } else if (border == KM_BORDER_REPLICATE) {
        if (p == 0)
            for (int s = -l; s < 0; ++s) acc += Gline[(s - base) * stride];
        if (p == n - 1)
            for (int s = n; s <= n - 1 + r; ++s) acc += Gline[(s - base) * stride];
    }
    return acc;
}

template <typename T>
__global__ __launch_bounds__(256) void km_filter_sep_bwd_kernel(const KmSepArgs<T> a) {
    typedef typename KmTraits<T>::R R;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const KmFilterGeom& g = a.g;
    const int kH = g.kH, kW = g.kW;
    const int lt = g.pt, rb = g.same ? kH - 1 - lt : 0;  // vertical pad before / after
    const int ll = g.pl, rr = g.same ? kW - 1 - ll : 0;  // horizontal pad
    // Tile of grad_in: rows [y0, y0+TH), cols [x0, x0+TW).
    // Step A (column adjoint) is needed on rows [y0-lt, y0+TH+rb) (for the vertical fold) and on the
    // intermediate columns the row adjoint will read: u in [x0-ll-(kW-1)+ll, ...] - see below.
    // Intermediate axis sizes: the row pass output has width Wi = Wo and height H ('same': H, 'valid': H).
    const int Wi = g.Wo;                 // width of the intermediate (row-pass output)
    const int EH = KM_FS_TH + lt + 2 * rb;  // extended rows of the column-adjoint result (2*rb: a reflect
                                          // fold source of a non-final tile lies up to 2*rb below it)
    // row adjoint at column s (s in [x0-ll, x0+TW+rr)) reads intermediate-gradient columns
    // u = s + ll - t, t in [0,kW)  =>  u in [x0 - (kW-1), x0 + TW + rr + ll)
    const int UW = KM_FS_TW + (kW - 1) + ll + 2 * rr;  // staged intermediate-gradient columns
    const int EW = KM_FS_TW + ll + 2 * rr;              // extended columns of the row-adjoint result
    // column adjoint at row s reads gy rows v = s + lt - t => v in [y0 - (kH-1), y0 + TH + rb + lt)
    const int VH = KM_FS_TH + (kH - 1) + lt + 2 * rb;
    R* s_gy = (R*)smem_raw;          // [VH][UW]  staged grad_out
    R* s_g1 = s_gy + VH * UW;        // [EH][UW]  column adjoint G1 (extended rows)
    R* s_gi = s_g1 + EH * UW;        // [TH][UW]  folded gradient wrt the intermediate
    R* s_g2 = s_gi + KM_FS_TH * UW;  // [TH][EW]  row adjoint G2 (extended cols)
    R* s_kx = s_g2 + KM_FS_TH * EW;
    R* s_ky = s_kx + kW;

    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int b = bc / g.C;
    const int x0 = tx * KM_FS_TW, y0 = ty * KM_FS_TH;
    const T* gy = a.x + (size_t)bc * g.Ho * g.Wo;
    T* gx = a.y + (size_t)bc * g.H * g.W;
    const int tid = threadIdx.x;

    const R* kxp = a.kx + (size_t)(b % g.Bk) * kW;
    const R* kyp = a.ky + (size_t)(b % g.Bk) * kH;
    for (int t = tid; t < kW; t += 256) s_kx[t] = kxp[t];
    for (int t = tid; t < kH; t += 256) s_ky[t] = kyp[t];

    // stage grad_out rows [y0-(kH-1), ...) x cols [x0-(kW-1), ...), zero-extended (periodic if circular)
    const int v_base = y0 - (kH - 1), u_base = x0 - (kW - 1);
    for (int e = tid; e < VH * UW; e += 256) {
        const int r = e / UW, c = e - r * UW;
        const int sy = g.same ? km_zero_index(v_base + r, g.Ho, g.border) : ((v_base + r >= 0 && v_base + r < g.Ho) ? v_base + r : -1);
        const int sx = g.same ? km_zero_index(u_base + c, g.Wo, g.border) : ((u_base + c >= 0 && u_base + c < g.Wo) ? u_base + c : -1);
        s_gy[e] = (sy >= 0 && sx >= 0) ? (R)km_ld(gy + (size_t)sy * g.Wo + sx) : (R)0;
    }
    __syncthreads();

    // column adjoint on extended rows s in [y0-lt, y0+TH+rb): G1[s][u] = sum_t ky[t] * gy0[s+lt-t][u]
    // staged row index of (s + lt - t) = (s + lt - t) - v_base = (s - (y0 - lt)) + (kH-1) - t
    for (int e = tid; e < EH * UW; e += 256) {
        const int r = e / UW, c = e - r * UW;
        R acc = 0;
        for (int t = 0; t < kH; ++t) acc = km_fma(s_ky[t], s_gy[(r + (kH - 1) - t) * UW + c], acc);
        s_g1[e] = acc;
    }
    __syncthreads();

    // vertical fold -> gradient wrt the intermediate at rows [y0, y0+TH) (intermediate height = H)
    for (int e = tid; e < KM_FS_TH * UW; e += 256) {
        const int r = e / UW, c = e - r * UW;
        const int p = y0 + r;
        R acc = s_g1[(r + lt) * UW + c];
        if (g.same && p < g.H) acc += km_fold_sum(s_g1 + c, UW, y0 - lt, p, g.H, lt, rb, g.border);
        // 16-bit storage: the reference materialises this gradient in the input dtype
        s_gi[e] = km_round_to(acc, (const T*)nullptr);
    }
    __syncthreads();

    // row adjoint on extended cols s in [x0-ll, x0+TW+rr): G2[r][s] = sum_t kx[t] * gi0[r][s+ll-t]
    // the intermediate has width Wi: columns outside [0,Wi) are zero (or periodic for circular) -
    // already encoded in the staging of s_gy (its columns were mapped with Wo == Wi).
    for (int e = tid; e < KM_FS_TH * EW; e += 256) {
        const int r = e / EW, c = e - r * EW;
        R acc = 0;
        for (int t = 0; t < kW; ++t) acc = km_fma(s_kx[t], s_gi[r * UW + c + (kW - 1) - t], acc);
        s_g2[e] = acc;
    }
    __syncthreads();

    // horizontal fold and store
    for (int e = tid; e < KM_FS_TH * KM_FS_TW; e += 256) {
        const int r = e / KM_FS_TW, c = e - r * KM_FS_TW;
        const int py = y0 + r, px = x0 + c;
        if (py < g.H && px < g.W) {
            R acc = s_g2[r * EW + c + ll];
            if (g.same) acc += km_fold_sum(s_g2 + r * EW, 1, x0 - ll, px, g.W, ll, rr, g.border);
            km_st(gx + (size_t)py * g.W + px, acc);
        }
    }
    (void)Wi;
}

// =================================================================================================
// generic full kH x kW kernels (direct global loads; L1/L2 provide the tap reuse)
// =================================================================================================
template <typename T>
struct KmFullArgs {
    typedef typename KmTraits<T>::R R;
    const T* x;
    const T* gy;
    T* y;
    const R* k;   // (Bk,kH,kW)
    double* gk;   // (Bk,kH,kW) fp64 accumulators
    KmFilterGeom g;
    uint32_t tiles_x, tiles_y, nblocks;
};

template <typename T>
__global__ __launch_bounds__(256) void km_filter2d_fwd_kernel(const KmFullArgs<T> a) {
    typedef typename KmTraits<T>::R R;
    const KmFilterGeom& g = a.g;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int b = bc / g.C;
    const int ox = tx * 64 + (threadIdx.x & 63);
    const int oy0 = ty * 16 + (threadIdx.x >> 6) * 4;
    if (ox >= g.Wo) return;
    const T* img = a.x + (size_t)bc * g.H * g.W;
    T* out = a.y + (size_t)bc * g.Ho * g.Wo;
    const R* kk = a.k + (size_t)(b % g.Bk) * g.kH * g.kW;
    for (int r = 0; r < 4; ++r) {
        const int oy = oy0 + r;
        if (oy >= g.Ho) break;
        R acc = 0;
        for (int p = 0; p < g.kH; ++p) {
            const int sy = g.same ? km_border_map(oy + p - g.pt, g.H, g.border) : oy + p;
            for (int q = 0; q < g.kW; ++q) {
                const int sx = g.same ? km_border_map(ox + q - g.pl, g.W, g.border) : ox + q;
                const R v = (sy >= 0 && sx >= 0) ? (R)km_ld(img + (size_t)sy * g.W + sx) : (R)0;
                acc = km_fma(kk[p * g.kW + q], v, acc);
            }
        }
        km_st(out + (size_t)oy * g.Wo + ox, acc);
    }
}

// pre-images of p under the border map along one axis: writes up to `cap` padded coordinates
// (unpadded coordinate system, s in [-l, n-1+r]) and returns the count.  p itself is always first.
__device__ __forceinline__ int km_preimages(int p, int n, int l, int r, int border, int same, int* out, int cap) {
    int cnt = 0;
    out[cnt++] = p;
    if (!same) return cnt;
    if (border == KM_BORDER_REFLECT) {
        if (p >= 1 && p <= l && cnt < cap) out[cnt++] = -p;
        if (p <= n - 2 && p >= n - 1 - r && cnt < cap) out[cnt++] = 2 * (n - 1) - p;
    } else if (border == KM_BORDER_REPLICATE) {
        if (p == 0)
            for (int s = -l; s < 0 && cnt < cap; ++s) out[cnt++] = s;
        if (p == n - 1)
            for (int s = n; s <= n - 1 + r && cnt < cap; ++s) out[cnt++] = s;
    } else if (border == KM_BORDER_CIRCULAR) {
        if (p - n >= -l && cnt < cap) out[cnt++] = p - n;
        if (p + n <= n - 1 + r && cnt < cap) out[cnt++] = p + n;
    }
    return cnt;
}

#define KM_MAX_PRE 64

// adjoint of the padded correlation at one input pixel (py, px) of plane bc: every padded coordinate that the border
// map sends to (py, px) (the pixel itself + its pre-images) collects  sum_{p,q} k[p][q] * gy[s + pad - (p,q)]
template <typename T>
__device__ __forceinline__ typename KmTraits<T>::R km_f2d_adjoint_pixel(const KmFullArgs<T>& a, uint32_t bc, int py, int px) {
    typedef typename KmTraits<T>::R R;
    const KmFilterGeom& g = a.g;
    const int b = bc / g.C;
    const T* gy = a.gy + (size_t)bc * g.Ho * g.Wo;
    const R* kk = a.k + (size_t)(b % g.Bk) * g.kH * g.kW;
    const int rb = g.same ? g.kH - 1 - g.pt : 0, rr = g.same ? g.kW - 1 - g.pl : 0;
    int sxs[KM_MAX_PRE], sys[KM_MAX_PRE];
    const int nsx = km_preimages(px, g.W, g.pl, rr, g.border, g.same, sxs, KM_MAX_PRE);
    const int nsy = km_preimages(py, g.H, g.pt, rb, g.border, g.same, sys, KM_MAX_PRE);
    R acc = 0;
    for (int iy = 0; iy < nsy; ++iy)
        for (int ix = 0; ix < nsx; ++ix) {
            // G[sy][sx] = sum_{p,q} k[p][q] * gy0[sy + pt - p][sx + pl - q]
            const int sy = sys[iy], sx = sxs[ix];
            for (int p = 0; p < g.kH; ++p) {
                const int oy = sy + g.pt - p;
                if (oy < 0 || oy >= g.Ho) continue;
                for (int q = 0; q < g.kW; ++q) {
                    const int ox = sx + g.pl - q;
                    if (ox < 0 || ox >= g.Wo) continue;
                    acc = km_fma(kk[p * g.kW + q], (R)km_ld(gy + (size_t)oy * g.Wo + ox), acc);
                }
            }
        }
    return acc;
}

template <typename T>
__global__ __launch_bounds__(256) void km_filter2d_bwd_input_kernel(const KmFullArgs<T> a) {
    const KmFilterGeom& g = a.g;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int px = tx * 64 + (threadIdx.x & 63);
    const int py0 = ty * 16 + (threadIdx.x >> 6) * 4;
    if (px >= g.W) return;
    T* gx = a.y + (size_t)bc * g.H * g.W;
    for (int r = 0; r < 4; ++r) {
        const int py = py0 + r;
        if (py >= g.H) break;
        km_st(gx + (size_t)py * g.W + px, km_f2d_adjoint_pixel<T>(a, bc, py, px));
    }
}

// The adjoint differs from "correlate the zero-extended gradient with the rotated taps" only within pad pixels of an
// image edge (where padded coordinates fold back).  This kernel recomputes exactly that frame (pd = pad + 1 pixels wide:
// 2*pd*(W + H - 2*pd) pixels per plane) - after the register-tiled kernel (km_filter2d_fast.hip) has produced the interior.
template <typename T>
__global__ __launch_bounds__(256) void km_filter2d_bwd_frame_kernel(const KmFullArgs<T> a, int pd, uint32_t frame_px, uint32_t blocks_per_plane) {
    const KmFilterGeom& g = a.g;
    const uint32_t bc = blockIdx.x / blocks_per_plane;
    const uint32_t idx = (blockIdx.x % blocks_per_plane) * 256u + threadIdx.x;
    if (idx >= frame_px) return;
    const uint32_t band = 2u * (uint32_t)pd * (uint32_t)g.W;  // top + bottom rows, full width
    int py, px;
    if (idx < band) {
        const int rr = (int)(idx / (uint32_t)g.W);
        px = (int)(idx % (uint32_t)g.W);
        py = rr < pd ? rr : g.H - 2 * pd + rr;
    } else {
        const uint32_t i2 = idx - band;
        const int cc = (int)(i2 % (uint32_t)(2 * pd));
        py = pd + (int)(i2 / (uint32_t)(2 * pd));
        px = cc < pd ? cc : g.W - 2 * pd + cc;
    }
    T* gx = a.y + (size_t)bc * g.H * g.W;
    km_st(gx + (size_t)py * g.W + px, km_f2d_adjoint_pixel<T>(a, bc, py, px));
}

// gradient wrt the taps: gk[b % Bk][p][q] += sum_{i,j} gy[b,c,i,j] * xpad[b,c,i+p,j+q]
// grid: (B*C, kH*kW); one block reduces one tap of one image plane.
template <typename T>
__global__ __launch_bounds__(256) void km_filter2d_bwd_kernel_kernel(const KmFullArgs<T> a) {
    const KmFilterGeom& g = a.g;
    __shared__ double red[4];
    const int bc = blockIdx.x, b = bc / g.C;
    const int p = blockIdx.y / g.kW, q = blockIdx.y % g.kW;
    const T* img = a.x + (size_t)bc * g.H * g.W;
    const T* gy = a.gy + (size_t)bc * g.Ho * g.Wo;
    double acc = 0;
    const int n = g.Ho * g.Wo;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int i = e / g.Wo, j = e - i * g.Wo;
        const int sy = g.same ? km_border_map(i + p - g.pt, g.H, g.border) : i + p;
        const int sx = g.same ? km_border_map(j + q - g.pl, g.W, g.border) : j + q;
        if (sy >= 0 && sx >= 0) acc += (double)km_ld(gy + e) * (double)km_ld(img + (size_t)sy * g.W + sx);
    }
    acc = km_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        km_atomic_add(a.gk + ((size_t)(b % g.Bk) * g.kH + p) * g.kW + q, (red[0] + red[1]) + (red[2] + red[3]));
}

// =================================================================================================
// host side
// =================================================================================================
static int km_filter_validate(const char* fn, int B, int C, int H, int W, int Bk, int kH, int kW, int border, int same, int dtype) {
    KM_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0, "%s: bad shape B=%d C=%d H=%d W=%d", fn, B, C, H, W);
    KM_REQUIRE((int64_t)H * W < (1ll << 31), "%s: image plane exceeds 2^31 elements", fn);
    KM_REQUIRE(kH > 0 && kW > 0, "%s: bad kernel size %dx%d", fn, kH, kW);
    KM_REQUIRE(Bk >= 1 && (B == 0 || B % Bk == 0), "%s: kernel batch %d must divide the input batch %d", fn, Bk, B);
    KM_REQUIRE(border >= 0 && border <= 3, "%s: bad border %d", fn, border);
    KM_REQUIRE(dtype >= 0 && dtype <= 3, "%s: bad dtype %d", fn, dtype);
    if (same) {
        if (border == KM_BORDER_REFLECT)
            KM_REQUIRE(kH - 1 - (kH - 1) / 2 < H && kW - 1 - (kW - 1) / 2 < W, "%s: reflect padding must be smaller than the image", fn);
        if (border == KM_BORDER_CIRCULAR)
            KM_REQUIRE(kH - 1 - (kH - 1) / 2 <= H && kW - 1 - (kW - 1) / 2 <= W, "%s: circular padding must not exceed the image", fn);
    } else {
        KM_REQUIRE(kH <= H && kW <= W, "%s: 'valid' needs kernel <= image", fn);
    }
    return 0;
}

template <typename T>
static size_t km_sep_fwd_lds(int kH, int kW) {
    typedef typename KmTraits<T>::R R;
    const size_t IW = KM_FS_TW + kW - 1, IH = KM_FS_TH + kH - 1;
    return (IH * IW + IH * KM_FS_TW + kW + kH) * sizeof(R);
}
template <typename T>
static size_t km_sep_bwd_lds(int kH, int kW, int same) {
    typedef typename KmTraits<T>::R R;
    const int lt = same ? (kH - 1) / 2 : 0, rb = same ? kH - 1 - lt : 0, ll = same ? (kW - 1) / 2 : 0, rr = same ? kW - 1 - ll : 0;
    const size_t EH = KM_FS_TH + lt + 2 * rb, UW = KM_FS_TW + (kW - 1) + ll + 2 * rr, EW = KM_FS_TW + ll + 2 * rr, VH = KM_FS_TH + (kH - 1) + lt + 2 * rb;
    return (VH * UW + EH * UW + KM_FS_TH * UW + KM_FS_TH * EW + kW + kH) * sizeof(R);
}

#define KM_LDS_LIMIT (64 * 1024)

template <typename T>
static int km_sep_run(bool bwd, const void* x, const void* kx, const void* ky, void* y, const KmFilterGeom& g, hipStream_t s) {
    typedef typename KmTraits<T>::R R;
    KmSepArgs<T> a;
    a.x = (const T*)x; a.y = (T*)y; a.kx = (const R*)kx; a.ky = (const R*)ky; a.g = g;
    const int ow = bwd ? g.W : g.Wo, oh = bwd ? g.H : g.Ho;
    a.tiles_x = (ow + KM_FS_TW - 1) / KM_FS_TW;
    a.tiles_y = (oh + KM_FS_TH - 1) / KM_FS_TH;
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)g.B * g.C;
    KM_REQUIRE(nb < (1ull << 31), "km_filter2d_sep: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    const size_t lds = bwd ? km_sep_bwd_lds<T>(g.kH, g.kW, g.same) : km_sep_fwd_lds<T>(g.kH, g.kW);
    KM_REQUIRE(lds <= KM_LDS_LIMIT, "km_filter2d_sep: kernel %dx%d needs %zu B of LDS (> %d); use the two-pass generic path", g.kH, g.kW, lds, KM_LDS_LIMIT);
    if (bwd)
        hipLaunchKernelGGL(km_filter_sep_bwd_kernel<T>, dim3(a.nblocks), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL(km_filter_sep_fwd_kernel<T>, dim3(a.nblocks), dim3(256), lds, s, a);
    return km_check_launch(bwd ? "km_filter2d_sep_bwd_input" : "km_filter2d_sep_fwd");
}

template <typename T>
static int km_full_run(int which, const void* x, const void* gy, const void* k, void* y, double* gk, const KmFilterGeom& g, hipStream_t s) {
    typedef typename KmTraits<T>::R R;
    KmFullArgs<T> a;
    a.x = (const T*)x; a.gy = (const T*)gy; a.y = (T*)y; a.k = (const R*)k; a.gk = gk; a.g = g;
    if (which == 2) {
        if ((uint64_t)g.B * g.C == 0) return 0;
        KM_REQUIRE((uint64_t)g.kH * g.kW < 65536, "km_filter2d_bwd_kernel: kernel too large");
        hipLaunchKernelGGL(km_filter2d_bwd_kernel_kernel<T>, dim3(g.B * g.C, g.kH * g.kW), dim3(256), 0, s, a);
        return km_check_launch("km_filter2d_bwd_kernel");
    }
    const int ow = which == 1 ? g.W : g.Wo, oh = which == 1 ? g.H : g.Ho;
    a.tiles_x = (ow + 63) / 64;
    a.tiles_y = (oh + 15) / 16;
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)g.B * g.C;
    KM_REQUIRE(nb < (1ull << 31), "km_filter2d: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    if (which == 0)
        hipLaunchKernelGGL(km_filter2d_fwd_kernel<T>, dim3(a.nblocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(km_filter2d_bwd_input_kernel<T>, dim3(a.nblocks), dim3(256), 0, s, a);
    return km_check_launch(which == 0 ? "km_filter2d_fwd" : "km_filter2d_bwd_input");
}

template <typename T>
static int km_full_frame_run(const void* gy, const void* k, void* gx, const KmFilterGeom& g, int pd, hipStream_t s) {
    typedef typename KmTraits<T>::R R;
    KmFullArgs<T> a;
    a.x = nullptr; a.gy = (const T*)gy; a.y = (T*)gx; a.k = (const R*)k; a.gk = nullptr; a.g = g;
    a.tiles_x = a.tiles_y = a.nblocks = 0;
    const uint64_t frame = 2ull * pd * g.W + 2ull * pd * (uint64_t)(g.H - 2 * pd);
    const uint32_t bpp = (uint32_t)((frame + 255) / 256);
    const uint64_t nb = (uint64_t)bpp * g.B * g.C;
    KM_REQUIRE(nb < (1ull << 31), "km_filter2d_bwd_input: grid too large");
    if (nb == 0) return 0;
    hipLaunchKernelGGL(km_filter2d_bwd_frame_kernel<T>, dim3((uint32_t)nb), dim3(256), 0, s, a, pd, (uint32_t)frame, bpp);
    return km_check_launch("km_filter2d_bwd_input(frame)");
}

// large separable kernels, forward (km_filter_sep_big.hip)
int km_filter_sep_big_supported(int kH, int kW, int dtype);
int km_filter_sep_big_run(const void* x, const void* kx, const void* ky, void* y, int B, int C, int H, int W, int Bk, int kH, int kW,
                          int border, int same, int dtype, hipStream_t s);

// register-tiled full kH x kW forward for square odd 3/5/7 kernels (km_filter2d_fast.hip)
int km_filter2d_fast_tapgrad_run(const void* gy, const void* x, double* gk, int B, int C, int H, int W, int Bk, int K, int border, int dtype,
                                 hipStream_t s);
int km_filter2d_fast_supported(const void* x, const void* y, int H, int W, int kH, int kW, int border, int same, int dtype);
int km_filter2d_fast_run(const void* x, const void* k, void* y, int B, int C, int H, int W, int Bk, int K, int border, int flip, int dtype,
                         hipStream_t s);

// register-tiled fast path for small square odd kernels (km_blur_fast.hip)
int km_blur_fast_supported(const void* x, const void* y, int H, int W, int kH, int kW, int border, int same, int dtype);
int km_blur_fast_run(bool bwd, const void* x, const void* kx, const void* ky, void* y, int B, int C, int H, int W, int Bk, int K,
                     int border, int dtype, hipStream_t s);
static int km_sep_algo() {
    // KM_SEP_ALGO=lds forces the generic LDS-tiled kernels (debugging / A-B timing)
    return km_config().sep_lds;
}

#define KM_DISPATCH_DTYPE(dtype, CALL)                      \
    switch (dtype) {                                        \
        case KM_F32: return CALL(float);                    \
        case KM_F64: return CALL(double);                   \
        case KM_BF16: return CALL(km_bf16);                 \
        default: return CALL(km_f16);                       \
    }

extern "C" {

// Replaces F.pad + F.conv2d(groups=Bk*C) of kornia/filters/filter.py:131-150.
// k: prepared taps (Bk,kH,kW) in the compute dtype (fp32, or fp64 for f64 data).
int km_filter2d_fwd(const void* x, const void* k, void* y, int B, int C, int H, int W, int Bk, int kH, int kW, int border,
                    int same, int dtype, void* stream) {
    if (B == 0 || C == 0) return 0;  // empty batch
    if (km_filter_validate("km_filter2d_fwd", B, C, H, W, Bk, kH, kW, border, same, dtype)) return -1;
    KM_REQUIRE(x && k && y, "km_filter2d_fwd: null pointer");
    if (km_sep_algo() == 0 && km_filter2d_fast_supported(x, y, H, W, kH, kW, border, same, dtype))
        return km_filter2d_fast_run(x, k, y, B, C, H, W, Bk, kH, border, 0, dtype, (hipStream_t)stream);
    const KmFilterGeom g = km_filter_geom(B, C, H, W, Bk, kH, kW, border, same);
#define CALL(T) km_full_run<T>(0, x, nullptr, k, y, nullptr, g, (hipStream_t)stream)
    KM_DISPATCH_DTYPE(dtype, CALL)
#undef CALL
}

// gradient wrt input (adjoint incl. the pad fold): gy (B,C,Ho,Wo) -> gx (B,C,H,W)
int km_filter2d_bwd_input(const void* gy, const void* k, void* gx, int B, int C, int H, int W, int Bk, int kH, int kW,
                          int border, int same, int dtype, void* stream) {
    if (B == 0 || C == 0) return 0;  // empty batch
    if (km_filter_validate("km_filter2d_bwd_input", B, C, H, W, Bk, kH, kW, border, same, dtype)) return -1;
    KM_REQUIRE(gy && k && gx, "km_filter2d_bwd_input: null pointer");
    KM_REQUIRE(!same || ((kH - 1 - (kH - 1) / 2) < KM_MAX_PRE && (kW - 1 - (kW - 1) / 2) < KM_MAX_PRE), "km_filter2d_bwd_input: kernel too large");
    const KmFilterGeom g = km_filter_geom(B, C, H, W, Bk, kH, kW, border, same);
    {
        // square odd 3/5/7 kernels: interior by the register-tiled kernel on the zero-extended gradient with the rotated taps
        // (for constant padding that is the whole adjoint), then the pad-wide frame by the exact pre-image kernel
        // frame width: reflect folds padded coordinate -p onto pixel p for p = 1..pad, so pixels 0..pad are affected
        const int pd = (kH - 1) / 2 + 1;
        if (km_sep_algo() == 0 && km_filter2d_fast_supported(gy, gx, H, W, kH, kW, border, same, dtype) && H > 2 * pd && W > 2 * pd) {
            const int rc = km_filter2d_fast_run(gy, k, gx, B, C, H, W, Bk, kH, KM_BORDER_CONSTANT, 1, dtype, (hipStream_t)stream);
            if (rc != 0 || border == KM_BORDER_CONSTANT) return rc;
#define CALLF(T) km_full_frame_run<T>(gy, k, gx, g, pd, (hipStream_t)stream)
            KM_DISPATCH_DTYPE(dtype, CALLF)
#undef CALLF
        }
    }
#define CALL(T) km_full_run<T>(1, nullptr, gy, k, gx, nullptr, g, (hipStream_t)stream)
    KM_DISPATCH_DTYPE(dtype, CALL)
#undef CALL
}

// gradient wrt the prepared taps: gk (Bk,kH,kW) fp64, pre-zeroed
int km_filter2d_bwd_kernel(const void* gy, const void* x, void* gk, int B, int C, int H, int W, int Bk, int kH, int kW,
                           int border, int same, int dtype, void* stream) {
    if (B == 0 || C == 0) return 0;  // empty batch
    if (km_filter_validate("km_filter2d_bwd_kernel", B, C, H, W, Bk, kH, kW, border, same, dtype)) return -1;
    KM_REQUIRE(gy && x && gk, "km_filter2d_bwd_kernel: null pointer");
    if (km_sep_algo() == 0 && km_filter2d_fast_supported(x, gy, H, W, kH, kW, border, same, dtype))  // one pass instead of kH*kW
        return km_filter2d_fast_tapgrad_run(gy, x, (double*)gk, B, C, H, W, Bk, kH, border, dtype, (hipStream_t)stream);
    const KmFilterGeom g = km_filter_geom(B, C, H, W, Bk, kH, kW, border, same);
#define CALL(T) km_full_run<T>(2, x, gy, nullptr, nullptr, (double*)gk, g, (hipStream_t)stream)
    KM_DISPATCH_DTYPE(dtype, CALL)
#undef CALL
}

// Replaces filter2d_separable (filter.py:155-207): kx (Bk,kW), ky (Bk,kH) in the compute dtype.
int km_filter2d_sep_fwd(const void* x, const void* kx, const void* ky, void* y, int B, int C, int H, int W, int Bk, int kH,
                        int kW, int border, int same, int dtype, void* stream) {
    if (B == 0 || C == 0) return 0;  // empty batch
    if (km_filter_validate("km_filter2d_sep_fwd", B, C, H, W, Bk, kH, kW, border, same, dtype)) return -1;
    KM_REQUIRE(x && kx && ky && y, "km_filter2d_sep_fwd: null pointer");
    if (km_sep_algo() == 0 && km_blur_fast_supported(x, y, H, W, kH, kW, border, same, dtype))
        return km_blur_fast_run(false, x, kx, ky, y, B, C, H, W, Bk, kH, border, dtype, (hipStream_t)stream);
    if (km_sep_algo() == 0 && km_filter_sep_big_supported(kH, kW, dtype))  // k >= 10: sliding-window LDS kernel
        return km_filter_sep_big_run(x, kx, ky, y, B, C, H, W, Bk, kH, kW, border, same, dtype, (hipStream_t)stream);
    const KmFilterGeom g = km_filter_geom(B, C, H, W, Bk, kH, kW, border, same);
#define CALL(T) km_sep_run<T>(false, x, kx, ky, y, g, (hipStream_t)stream)
    KM_DISPATCH_DTYPE(dtype, CALL)
#undef CALL
}

int km_filter2d_sep_bwd_input(const void* gy, const void* kx, const void* ky, void* gx, int B, int C, int H, int W, int Bk,
                              int kH, int kW, int border, int same, int dtype, void* stream) {
    if (B == 0 || C == 0) return 0;  // empty batch
    if (km_filter_validate("km_filter2d_sep_bwd_input", B, C, H, W, Bk, kH, kW, border, same, dtype)) return -1;
    KM_REQUIRE(gy && kx && ky && gx, "km_filter2d_sep_bwd_input: null pointer");
    if (km_sep_algo() == 0 && km_blur_fast_supported(gy, gx, H, W, kH, kW, border, same, dtype))
        return km_blur_fast_run(true, gy, kx, ky, gx, B, C, H, W, Bk, kH, border, dtype, (hipStream_t)stream);
    KM_REQUIRE(kH - 1 <= KM_FS_TH && kW - 1 <= KM_FS_TW, "km_filter2d_sep_bwd_input: kernel larger than the tile; use the generic path");
    const KmFilterGeom g = km_filter_geom(B, C, H, W, Bk, kH, kW, border, same);
#define CALL(T) km_sep_run<T>(true, gy, kx, ky, gx, g, (hipStream_t)stream)
    KM_DISPATCH_DTYPE(dtype, CALL)
#undef CALL
}

// Which fused separable kernels can take this kernel size (LDS budgets): bit 0 = forward (km_filter2d_sep_fwd),
// bit 1 = gradient wrt the input (km_filter2d_sep_bwd_input).  A caller without bit 1 runs the adjoint as two
// km_filter2d_bwd_input passes (column kernel, then row kernel).
int km_filter2d_sep_supported(int kH, int kW, int same, int dtype) {
    int mask = 0;
    if (km_filter_sep_big_supported(kH, kW, dtype)) mask |= 1;
    if (kH - 1 > KM_FS_TH || kW - 1 > KM_FS_TW) return mask;
    const size_t f = dtype == KM_F64 ? km_sep_fwd_lds<double>(kH, kW) : km_sep_fwd_lds<float>(kH, kW);
    const size_t b = dtype == KM_F64 ? km_sep_bwd_lds<double>(kH, kW, same) : km_sep_bwd_lds<float>(kH, kW, same);
    if (f <= KM_LDS_LIMIT) mask |= 1;
    if (b <= KM_LDS_LIMIT) mask |= 2;
    return mask;
}

}  // extern "C"
