// kornia_amd - the bilinear warps FUSED with the separable K x K blur that follows them (forward), for gfx950:
//   y = filter2d_separable(warp_perspective | warp_affine | homography_warp(src, M, dsize), kx, ky, border)
// i.e. kornia/geometry/transform/imgwarp.py:143-174 / :246-290 / :1539-1546 followed by kornia/filters/filter.py:155-207 (GaussianBlur2d:
// kornia/filters/gaussian.py:32-120) - BASELINE config 2's forward, RandomAffine / RandomPerspective followed by RandomGaussianBlur in an
// augmentation pipeline - as ONE launch that never writes the warped image: 5e bytes per element for the fused op's forward + backward against
// 9e for the two ops ("report it if fusion is added").  A separate public op (kornia_amd.geometry.transform.warp_perspective_blur &
// co.), reported separately by bench.py; the two-op path stays what the headline times.
// Built with -fno-slp-vectorize (kornia_amd/build.py): the kernel is bound by its vector work, and the SLP vectorizer's packed pairs cost more
// issue time than they save once the moves that feed them are counted (0.546 / 0.549 against 0.555 / 0.561 ms, profiles/r04/run39_*).
//
// A 256-thread block owns a 64 x 32 tile of the OUTPUT of the blur.  It needs the warped image on the tile grown by the blur's reach L =
// (K - 1) / 2 on every side - border pixels of that region are the warped image at the indices F.pad would read (reflect / replicate /
// circular: the warp evaluated at the mapped output pixel; constant: zero) - so the block
//   1. finds the source box of the region (the region's corners through the forward's own coordinate arithmetic, km_warp_stage.h) and requests
//      it with 16-byte row loads; the sampling positions of the thread's ~10 region pixels are computed while the requests fly;
//   2. samples the region from LDS - the arithmetic of km_warp_fwd_box_kernel, operation for operation: the warped values are bit-identical to
//      warp_*'s output (a pixel whose footprint is not in the box, or a block whose box does not fit, gathers from global memory instead) -
//      rounds them to the storage type and writes them back into the same LDS (the source box is dead by then);
//   3. blurs from LDS with the register-tiled blur's own scheme (km_blur_fast.hip): lane = output column, a wave walks its 8-row strip with a
//      K-row rolling window of row-pass results, the horizontal neighbours come from the neighbouring lanes' registers (DPP), the L columns
//      beyond the wave's ends from a second LDS read of the end lanes; row pass and column pass are the same fma chains in the same order, so y is bit-identical to
//      filter2d_separable(warp(...)) (tests/test_gpu_warp_blur.py: torch.equal).
// The backward of the op is the two existing launches (blur adjoint, one-read warp backward): a fused adjoint would need the blurred gradient
// of every tile's box next to the tile-owner's 96 KB of LDS.
#include <stdlib.h>

#include "km_warp_args.h"
#include "km_warp_stage.h"

template <typename T>
struct KmWarpBlurArgs {
    const T* src;      // (B,C,H,W)
    const float* mat;  // (B_M,9)
    const float* kx;   // (Bk,K) horizontal taps
    const float* ky;   // (Bk,K) vertical taps
    T* dst;            // (B,C,h,w)
    KmWarpGeom<float> g;
    int Bk, border;
    uint32_t tiles_x, tiles_y, nblocks, reverse, stream_out;
};

template <int K>
struct KmwbShape {
    static constexpr int L = (K - 1) / 2;
    static constexpr int TW = 64, TH = 32;              // output tile
    static constexpr int RW = TW + 2 * L, RH = TH + 2 * L;  // region of the warped image the tile's blur reads
    static constexpr int PITCH = ((RW + 16 + 3) / 4) * 4;   // capacity of the staged source box (KmbWide's slack: 16 columns, 8 rows)
    static constexpr int ROWS = RH + 8;
    static constexpr int NPIX = (RW * RH + 255) / 256;      // region pixels per thread
};
#define KMWB_LDS_FLOATS(K, NC) ((KmwbShape<K>::ROWS * KmwbShape<K>::PITCH > KmwbShape<K>::RH * KmwbShape<K>::RW ? KmwbShape<K>::ROWS * KmwbShape<K>::PITCH : KmwbShape<K>::RH * KmwbShape<K>::RW) * (NC))

__device__ __forceinline__ float kmwb_round(float v, const float*) { return v; }
__device__ __forceinline__ float kmwb_round(float v, const km_bf16*) { return __uint_as_float(((uint32_t)km_f32_to_bf16_bits(v)) << 16); }
__device__ __forceinline__ float kmwb_round(float v, const km_f16*) { KM_OPAQUE(v); return (float)(_Float16)v; }

// one region pixel by gathers from global memory (plain IEEE arithmetic on the same position; every tap predicated): a footprint outside the
// staged box, or a block whose box does not fit
template <typename T, int NC>
__device__ __forceinline__ void kmwb_gather(const T* __restrict__ src_b, size_t src_plane, int W, int H, float x, float y, float (&val)[NC]) {
    KmBilin<float> tr;
    km_bilinear_setup(x, y, W, H, tr);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const T* img = src_b + (size_t)c * src_plane;
        const float v00 = km_ld(img + tr.i00), v01 = km_ld(img + tr.i01), v10 = km_ld(img + tr.i10), v11 = km_ld(img + tr.i11);
        float acc = km_bilinear_masked(tr, (float)v00, (float)v01, (float)v10, (float)v11);  // (out-of-bounds taps: zeros that are still multiplied - ATen's CPU rule)
        val[c] = acc;
    }
}

template <typename T, int CM, int NC, int ALIGN, int K, bool STREAM>
__global__ __launch_bounds__(256, 3) void km_warp_blur_fwd_kernel(const KmWarpBlurArgs<T> a) {
    typedef KmwbShape<K> SH;
    constexpr int L = SH::L, TW = SH::TW, TH = SH::TH, RW = SH::RW, RH = SH::RH, PITCH = SH::PITCH, ROWS = SH::ROWS, NPIX = SH::NPIX;
    constexpr int NCHK = PITCH / 4;        // 16-byte chunks per staged row of one channel
    constexpr int RCPP = 256 / NCHK;       // (row, channel) pairs filled per pass of the block
    constexpr int NPASS = (ROWS * NC + RCPP - 1) / RCPP;
    static_assert(RW <= 128 && RH <= 64, "one wave fills the row table, two the column table");
    const KmWarpGeom<float>& g = a.g;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks, a.reverse);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int J0 = (int)tx * TW, I0 = (int)ty * TH;
    const int W = g.W, H = g.H, border = a.border;
    __shared__ float4 s_cu[RW];   // per region column: (m0 u, m3 u, m6 u, exists)
    __shared__ float4 s_rv[RH];   // per region row:    (m1 v, m4 v, m7 v, exists)  (homography mode: (v, 0, 0, exists))
    __shared__ int s_info[8];
    __shared__ __attribute__((aligned(16))) float s_buf[KMWB_LDS_FLOATS(K, NC)];  // the source box [row][channel][x], then the warped region [channel][row][x]

    float m[9];
    {
        const float* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mp[k])));
    }
    const float Wm1 = (float)(W - 1), Hm1 = (float)(H - 1), hW = (float)W / 2, hH = (float)H / 2;
    const size_t src_plane = (size_t)H * W, dst_plane = (size_t)g.h * g.w;
    const T* __restrict__ src_b = a.src + (size_t)b * NC * src_plane;

    // ---- tables of the region (the output pixel a region pixel stands for: F.pad's index map), the division guard, the source box ----
    if (wave == 0) {
        bool ok = true;
        if (lane < RH) {
            const int im = km_border_map(I0 - L + lane, g.h, border);
            const float v = km_base_y<float, CM>(g, max(im, 0));
            const KmlHalf h = kml_row_half<CM>(m, v);
            s_rv[lane] = make_float4(h.a, h.b, h.c, im >= 0 ? 1.f : 0.f);
            ok = (im < 0) || kml_row_guard<CM>(g, m, v);
        }
        const bool all_ok = __all(ok);
        if (lane == 0) s_info[0] = all_ok ? 1 : 0;
    } else if (wave == 1 || wave == 2) {
        const int c = (wave - 1) * 64 + lane;
        if (c < RW) {
            const int jm = km_border_map(J0 - L + c, g.w, border);
            const KmlHalf h = kml_col_half<CM>(m, km_base_x<float, CM>(g, max(jm, 0)));
            s_cu[c] = make_float4(h.a, h.b, h.c, jm >= 0 ? 1.f : 0.f);
        }
    } else {
        // the output pixels the region stands for lie in [jlo, jhi] x [ilo, ihi] (reflect / replicate map into the tile's own neighbourhood;
        // a circular wrap does not: such a block gathers)
        const int jlo = max(J0 - L, 0), jhi = min(J0 + TW - 1 + L, g.w - 1), ilo = max(I0 - L, 0), ihi = min(I0 + TH - 1 + L, g.h - 1);
        const bool wraps = (border == KM_BORDER_CIRCULAR) && (J0 - L < 0 || I0 - L < 0 || J0 + TW - 1 + L > g.w - 1 || I0 + TH - 1 + L > g.h - 1);
        const int jc = (lane & 1) ? jhi : jlo, ic = (lane & 2) ? ihi : ilo;
        const KmlHalf cu = kml_col_half<CM>(m, km_base_x<float, CM>(g, jc));
        const KmlHalf rv = kml_row_half<CM>(m, km_base_y<float, CM>(g, ic));
        KmlPos p;
        kml_position<CM, false>(m, cu, rv, p);
        const float x = kml_unnormalize<ALIGN>(p.gx, Wm1, hW), y = kml_unnormalize<ALIGN>(p.gy, Hm1, hH);
        const float dsel = (CM == KM_COORD_AFFINE) ? 1.0f : p.den;
        float xmin = x, xmax = x, ymin = y, ymax = y, dmin = dsel, dmax = dsel;
#pragma unroll
        for (int off = 1; off <= 2; off <<= 1) {
            xmin = fminf(xmin, __shfl_down(xmin, off, 64)); xmax = fmaxf(xmax, __shfl_down(xmax, off, 64));
            ymin = fminf(ymin, __shfl_down(ymin, off, 64)); ymax = fmaxf(ymax, __shfl_down(ymax, off, 64));
            dmin = fminf(dmin, __shfl_down(dmin, off, 64)); dmax = fmaxf(dmax, __shfl_down(dmax, off, 64));
        }
        const bool finite4 = __all((lane > 3) || ((x == x) && (y == y) && (km_fabs(x) < 1.0e8f) && (km_fabs(y) < 1.0e8f) && (dsel == dsel)));
        if (lane == 0) {
            const int bx0 = (int)km_floor(xmin) - 1, bx1 = (int)km_floor(xmax) + 2;
            const int by0 = (int)km_floor(ymin) - 1, by1 = (int)km_floor(ymax) + 2;
            const int xs = bx0 & ~3;
            const int wcols = bx1 - xs + 1, nrows = by1 - by0 + 1;
            const bool same_sign = (dmin > 0.f) || (dmax < 0.f);
            const bool vec_ok = (sizeof(T) == 4) ? ((W & 3) == 0 && ((uintptr_t)a.src & 15) == 0) : ((W & 3) == 0 && ((uintptr_t)a.src & 7) == 0);
            s_info[1] = (finite4 && same_sign && !wraps && vec_ok && wcols <= PITCH && nrows <= ROWS) ? 1 : 0;
            s_info[2] = xs;
            s_info[3] = by0;
            s_info[4] = (wcols + 3) >> 2;
            s_info[5] = nrows;
        }
    }
    __syncthreads();
    const KmfBox bx = kmf_read_box(s_info);

    // ---- the fill's requests (block-uniform: only when the box fits) ----
    const int ck = tid % NCHK, rq = tid / NCHK;
    const int xg = bx.xs + 4 * ck;
    const bool filler = bx.staged && (rq < RCPP) && (ck < bx.nch);
    const bool col_in = (xg >= 0) && (xg + 3 < W);
    const int nrc = bx.nrows * NC;
    float v[NPASS][4];
    uint32_t inmask = 0u;
    if (bx.staged) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int rc = ps * RCPP + rq, r = rc / NC, c = rc - r * NC, y = bx.ys + r;
            const bool inb = filler && col_in && (rc < nrc) && (y >= 0) && (y < H);
            inmask |= inb ? (1u << ps) : 0u;
            km_ld4(km_at(src_b + (inb ? c : 0) * src_plane, inb ? (uint32_t)y * (uint32_t)W + (uint32_t)xg : 0u), v[ps]);
        }
    }

    // ---- this thread's region pixels e = tid + 256 k: positions, while the requests fly ----
    float xs[NPIX], ys[NPIX];
    uint32_t exists = 0u;
#pragma unroll
    for (int k = 0; k < NPIX; ++k) {
        const int e = min(tid + 256 * k, RW * RH - 1);
        const int r = e / RW, c = e - r * RW;
        const float4 cu4 = s_cu[c], rv4 = s_rv[r];
        KmlHalf cu, rv;
        cu.a = cu4.x; cu.b = cu4.y; cu.c = cu4.z;
        rv.a = rv4.x; rv.b = rv4.y; rv.c = rv4.z;
        KmlPos p;
        if (bx.fast) kml_position<CM, true>(m, cu, rv, p);
        else kml_position<CM, false>(m, cu, rv, p);
        xs[k] = kml_unnormalize<ALIGN>(p.gx, Wm1, hW);
        ys[k] = kml_unnormalize<ALIGN>(p.gy, Hm1, hH);
        exists |= (cu4.w != 0.f && rv4.w != 0.f && tid + 256 * k < RW * RH) ? (1u << k) : 0u;
    }

    // ---- requests -> LDS ----
    if (bx.staged) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int rc = ps * RCPP + rq;
            if (filler && rc < nrc) {
                float* q = s_buf + rc * PITCH + 4 * ck;
                KM_CHECK_ALIGNED(q, 16);
                const bool inb = (inmask >> ps) & 1u;
                *reinterpret_cast<float4*>(q) = make_float4(inb ? v[ps][0] : 0.f, inb ? v[ps][1] : 0.f, inb ? v[ps][2] : 0.f, inb ? v[ps][3] : 0.f);
            }
        }
    }
    __syncthreads();

    // ---- sample the region: values rounded to the storage type (what warp_* stores and the blur then reads) ----
    float val[NPIX][NC];
    bool all_boxed = bx.staged;
#pragma unroll
    for (int k = 0; k < NPIX; ++k) {
        KmlTaps t;
        kml_taps(xs[k], ys[k], t);
        all_boxed = all_boxed && (kmf_in_box(t, bx) || !((exists >> k) & 1u));
    }
    if (__all(all_boxed)) {
        // the common case, straight-line: every footprint of the wave's pixels lies in the box (a pixel that does not exist - constant
        // border, beyond the region - reads the box origin and is zeroed)
#pragma unroll
        for (int k = 0; k < NPIX; ++k) {
            KmlTaps t;
            kml_taps(xs[k], ys[k], t);
            const bool live = (exists >> k) & 1u;
            const int xi = live ? KM_F2I(t.xf) - bx.xs : 0, yi = live ? KM_F2I(t.yf) - bx.ys : 0;
            const float* q0 = s_buf + __mul24(yi, NC * PITCH) + xi;
            const float* q1 = q0 + NC * PITCH;
            const float w00 = t.wx1 * t.wy1, w01 = t.wx0 * t.wy1, w10 = t.wx1 * t.wy0, w11 = t.wx0 * t.wy0;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float v00 = q0[c * PITCH], v01 = q0[c * PITCH + 1], v10 = q1[c * PITCH], v11 = q1[c * PITCH + 1];
                const float acc = km_fma(v11, w11, km_fma(v10, w10, km_fma(v01, w01, km_fma(v00, w00, 0.0f))));
                val[k][c] = live ? kmwb_round(acc, (const T*)nullptr) : 0.f;
            }
        }
    } else {
#pragma unroll 1
        for (int k = 0; k < NPIX; ++k) {
            KmlTaps t;
            kml_taps(xs[k], ys[k], t);
            const bool live = (exists >> k) & 1u;
            float vk[NC];
            if (bx.staged && kmf_in_box(t, bx)) {
                const int xi = KM_F2I(t.xf) - bx.xs, yi = KM_F2I(t.yf) - bx.ys;
                const float* q0 = s_buf + __mul24(yi, NC * PITCH) + xi;
                const float* q1 = q0 + NC * PITCH;
                const float w00 = t.wx1 * t.wy1, w01 = t.wx0 * t.wy1, w10 = t.wx1 * t.wy0, w11 = t.wx0 * t.wy0;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float v00 = q0[c * PITCH], v01 = q0[c * PITCH + 1], v10 = q1[c * PITCH], v11 = q1[c * PITCH + 1];
                    vk[c] = km_fma(v11, w11, km_fma(v10, w10, km_fma(v01, w01, km_fma(v00, w00, 0.0f))));
                }
            } else if (live) {
                kmwb_gather<T, NC>(src_b, src_plane, W, H, xs[k], ys[k], vk);
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) vk[c] = 0.f;
            }
            // (k is a run-time index here: the values go through the same registers by a select per slot)
#pragma unroll
            for (int kk = 0; kk < NPIX; ++kk)
#pragma unroll
                for (int c = 0; c < NC; ++c) val[kk][c] = (kk == k) ? (live ? kmwb_round(vk[c], (const T*)nullptr) : 0.f) : val[kk][c];
        }
    }
    __syncthreads();  // every tap of the source box has been read: the buffer becomes the warped region [channel][row][RW]
#pragma unroll
    for (int k = 0; k < NPIX; ++k) {
        const int e = tid + 256 * k;
        if (e < RW * RH) {
#pragma unroll
            for (int c = 0; c < NC; ++c) s_buf[c * (RH * RW) + e] = val[k][c];
        }
    }
    __syncthreads();

    // ---- the blur: lane = output column, the wave's strip of TH / 4 rows, K-row rolling window of row-pass results ----
    constexpr int SROWS = TH / 4;
    float kx[K], ky[K];
    {
        const float* px = a.kx + (size_t)(b % (uint32_t)a.Bk) * K;
        const float* py = a.ky + (size_t)(b % (uint32_t)a.Bk) * K;
#pragma unroll
        for (int t = 0; t < K; ++t) { kx[t] = px[t]; ky[t] = py[t]; }
    }
    const int j = J0 + lane, i_strip = I0 + wave * SROWS;
    const int ex_off = lane < L ? lane : (lane > 63 - L ? lane + 2 * L : lane + L);
    T* __restrict__ dst_b = a.dst + (size_t)b * NC * dst_plane;
    float ring[K][NC];
    constexpr int total = SROWS + K - 1;
    for (int it0 = 0; it0 < total; it0 += K) {
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int it = it0 + kk;
            if (it < total) {
                const int rr = wave * SROWS + it;  // region row
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float* rowp = s_buf + c * (RH * RW) + rr * RW;
                    // own column + ONE extra value per lane: the first / last L lanes hold the L region columns left / right of the wave's 64
                    // (the others re-read their own: no exec-mask region).  Neighbour d of lane l is lane l -+ d's own value (DPP shifts), or - for
                    // the lanes within d of an end - an end lane's extra value, shifted the other way
                    float tv[K];
                    const float own = rowp[lane + L], extra = rowp[ex_off];
                    tv[L] = own;
                    float po = own, no = own;         // own, shifted d lanes right / left
                    float esh[L];                     // extra shifted: esh[k] = lane l sees lane (l + k)'s [left side] ...
                    float esr[L];                     // ... and lane (l - k)'s [right side] extra value
                    esh[0] = extra; esr[0] = extra;
#pragma unroll
                    for (int q = 1; q < L; ++q) { esh[q] = km_next64(esh[q - 1]); esr[q] = km_prev64(esr[q - 1]); }
#pragma unroll
                    for (int d = 1; d <= L; ++d) {
                        po = km_prev64(po);
                        no = km_next64(no);
                        // lane l < d needs region column l + L - d = the extra of lane l + L - d; lane l > 63 - d needs column l + L + d = the extra of lane l - (L - d)
                        tv[L - d] = (lane < d) ? esh[L - d] : po;
                        tv[L + d] = (lane > 63 - d) ? esr[L - d] : no;
                    }
                    float acc = 0.f;
#pragma unroll
                    for (int q = 0; q < K; ++q) acc = km_fma(kx[q], tv[q], acc);
                    ring[kk][c] = kmwb_round(acc, (const T*)nullptr);
                }
                if (it >= K - 1) {
                    const int i = i_strip + it - (K - 1);
                    if (j < g.w && i < g.h) {
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            float acc = 0.f;
#pragma unroll
                            for (int p = 0; p < K; ++p) acc = km_fma(ky[p], ring[(kk + 1 + p) % K][c], acc);
                            km_st_c<STREAM>(km_at_mut(dst_b + (size_t)c * dst_plane, (uint32_t)i * (uint32_t)g.w + (uint32_t)j), acc);
                        }
                    }
                }
            }
        }
    }
}

template <typename T, int CM, int NC, int K>
static int kmwb_launch_k(const KmWarpBlurArgs<T>& a, hipStream_t s) {
    if (a.g.align) {
        if (a.stream_out) hipLaunchKernelGGL((km_warp_blur_fwd_kernel<T, CM, NC, 1, K, true>), dim3(a.nblocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((km_warp_blur_fwd_kernel<T, CM, NC, 1, K, false>), dim3(a.nblocks), dim3(256), 0, s, a);
    } else {
        if (a.stream_out) hipLaunchKernelGGL((km_warp_blur_fwd_kernel<T, CM, NC, 0, K, true>), dim3(a.nblocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((km_warp_blur_fwd_kernel<T, CM, NC, 0, K, false>), dim3(a.nblocks), dim3(256), 0, s, a);
    }
    return km_check_launch("km_warp2d_blur_fwd");
}
template <typename T, int CM, int NC>
static int kmwb_launch_nc(const KmWarpBlurArgs<T>& a, int K, hipStream_t s) {
    switch (K) {
        case 3: return kmwb_launch_k<T, CM, NC, 3>(a, s);
        case 5: return kmwb_launch_k<T, CM, NC, 5>(a, s);
        default: return kmwb_launch_k<T, CM, NC, 7>(a, s);
    }
}
template <typename T, int CM>
static int kmwb_launch(const KmWarpBlurArgs<T>& a, int K, hipStream_t s) {
    return a.g.C == 3 ? kmwb_launch_nc<T, CM, 3>(a, K, s) : kmwb_launch_nc<T, CM, 1>(a, K, s);
}
template <typename T>
static int kmwb_run(const void* src, const void* mat, const void* kx, const void* ky, void* dst, int B, int C, int H, int W, int h, int w, int B_M, int Bk,
                    int coord_mode, int norm_coords, int align, int K, int border, hipStream_t s) {
    KmWarpBlurArgs<T> a;
    a.src = (const T*)src; a.mat = (const float*)mat; a.kx = (const float*)kx; a.ky = (const float*)ky; a.dst = (T*)dst;
    km_geom_init(a.g, B, C, H, W, h, w, B_M, coord_mode, norm_coords, KM_INTERP_BILINEAR, KM_PAD_ZEROS, align);
    a.Bk = Bk; a.border = border;
    a.tiles_x = (uint32_t)((w + 63) / 64);
    a.tiles_y = (uint32_t)((h + 31) / 32);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B;
    KM_REQUIRE(nb < (1ull << 31), "km_warp2d_blur_fwd: grid too large");
    if (nb == 0) return 0;
    a.nblocks = (uint32_t)nb;
    a.reverse = km_traversal_next(s);
    a.stream_out = km_stream_stores((uint64_t)B * C * h * w * sizeof(T));
    switch (coord_mode) {
        case KM_COORD_PERSPECTIVE: return kmwb_launch<T, KM_COORD_PERSPECTIVE>(a, K, s);
        case KM_COORD_AFFINE: return kmwb_launch<T, KM_COORD_AFFINE>(a, K, s);
        default: return kmwb_launch<T, KM_COORD_HOMOGRAPHY>(a, K, s);
    }
}

extern "C" {

// 1 if km_warp2d_blur_fwd takes these modes: bilinear + zeros warp, grey / RGB, square odd K in {3, 5, 7}, 'same' output, storage fp32 / bf16 / f16
int km_warp2d_blur_supported(int C, int H, int W, int h, int w, int interp, int pad, int K, int border, int dtype) {
    if (interp != KM_INTERP_BILINEAR || pad != KM_PAD_ZEROS || dtype == KM_F64) return 0;
    if (!(C == 1 || C == 3) || !(K == 3 || K == 5 || K == 7)) return 0;
    if (W < 2 || H < 1 || h < 1 || w < 1) return 0;
    if (border == KM_BORDER_REFLECT && (K - 1) / 2 >= (h < w ? h : w)) return 0;
    return ((uint64_t)H * W * 4 < (1ull << 32) && (uint64_t)h * w * 4 < (1ull << 32) && W < (1 << 23) && H < (1 << 23)) ? 1 : 0;
}

int km_warp2d_blur_fwd(const void* src, const void* mat, const void* kx, const void* ky, void* dst, int B, int C, int H, int W, int h, int w, int B_M, int Bk,
                       int coord_mode, int norm_coords, int align_corners, int K, int border, int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    KM_REQUIRE(km_warp2d_blur_supported(C, H, W, h, w, KM_INTERP_BILINEAR, KM_PAD_ZEROS, K, border, dtype), "km_warp2d_blur_fwd: unsupported modes (km_warp2d_blur_supported)");
    KM_REQUIRE(B_M == 1 || B_M == B, "km_warp2d_blur_fwd: B_M must be 1 or B");
    KM_REQUIRE(Bk >= 1, "km_warp2d_blur_fwd: Bk must be >= 1");
    switch (dtype) {
        case KM_F32: return kmwb_run<float>(src, mat, kx, ky, dst, B, C, H, W, h, w, B_M, Bk, coord_mode, norm_coords, align_corners, K, border, s);
        case KM_BF16: return kmwb_run<km_bf16>(src, mat, kx, ky, dst, B, C, H, W, h, w, B_M, Bk, coord_mode, norm_coords, align_corners, K, border, s);
        default: return kmwb_run<km_f16>(src, mat, kx, ky, dst, B, C, H, W, h, w, B_M, Bk, coord_mode, norm_coords, align_corners, K, border, s);
    }
}

}  // extern "C"
