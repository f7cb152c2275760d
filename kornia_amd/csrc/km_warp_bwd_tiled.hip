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

// matrix-gradient contribution of one output pixel
template <int CM>
__device__ __forceinline__ void kmt_accumulate_gm(float (&gm)[9], const KmCoord<float>& cd, float gix, float giy) {
    typedef float R;
    if (CM == KM_COORD_PERSPECTIVE) {
        const R inv = __frcp_rn(cd.den);
        const R ax = gix * inv, ay = giy * inv;
        const R az = -(gix * cd.gx + giy * cd.gy) * inv;
        gm[0] += ax * cd.u; gm[1] += ax * cd.v; gm[2] += ax;
        gm[3] += ay * cd.u; gm[4] += ay * cd.v; gm[5] += ay;
        gm[6] += az * cd.u; gm[7] += az * cd.v; gm[8] += az;
    } else if (CM == KM_COORD_AFFINE) {
        gm[0] += gix * cd.u; gm[1] += gix * cd.v; gm[2] += gix;
        gm[3] += giy * cd.u; gm[4] += giy * cd.v; gm[5] += giy;
    } else {
        const R s = cd.den;
        const R ax = gix * s, ay = giy * s;
        const R az = cd.live ? -(gix * cd.X + giy * cd.Y) * s * s : (R)0;
        gm[0] += ax * cd.u; gm[1] += ax * cd.v; gm[2] += ax;
        gm[3] += ay * cd.u; gm[4] += ay * cd.v; gm[5] += ay;
        gm[6] += az * cd.u; gm[7] += az * cd.v; gm[8] += az;
    }
}

#define KMT_TAB 256  // capacity of the per-band base-coordinate tables

// block-uniform values computed with VALU float math live in VGPRs unless moved to SGPRs explicitly
__device__ __forceinline__ float kmt_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ int kmt_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// one output pixel of pass B.  Branch-free up to the (exec-masked) atomics: every load is unconditional
// (clamped addresses) so that the loads of two pixels processed back to back can be in flight together.
template <typename T, int CM, bool WANT_GM>
__device__ __forceinline__ void kmt_scatter_q(const KmWarpTiledArgs<T>& a, const float (&m)[9], int e, bool valid, int bw, float inv_bw,
                                             int j0, int ib, bool tab_x, const float* s_u, const float* s_v, int* s_acc, bool finite,
                                             float scale, int cbase, int cc, const T* src_b, const T* gout_b, size_t src_plane,
                                             size_t dst_plane, int X0, int X1, int Y0, int Y1, float (&gm)[9], float bound, bool& exceeded) {
    typedef float R;
    const KmWarpGeom<R>& g = a.g;
    int qi = (int)(((float)e + 0.5f) * inv_bw);
    int qj = e - qi * bw;
    if (qj < 0) { qi -= 1; qj += bw; }
    if (qj >= bw) { qi += 1; qj -= bw; }
    const int jj = j0 + qj, ii = ib + qi;
    const T* go_px = gout_b + (size_t)ii * g.w + jj;
    R go[KMT_CC];
#pragma unroll
    for (int c = 0; c < KMT_CC; ++c) go[c] = (c < cc) ? (R)km_ld(go_px + (size_t)(cbase + c) * dst_plane) : (R)0;
    // the fixed-point scale was chosen for |grad_out| <= bound; anything larger (or NaN) voids the attempt
#pragma unroll
    for (int c = 0; c < KMT_CC; ++c) exceeded = exceeded || !(km_fabs(go[c]) <= bound);
    KmCoord<R> cd;
    km_gen_coord<R, CM>(m, tab_x ? s_u[qj] : km_base_x<R, CM>(g, jj), s_v[qi], cd);
    R mx, my;
    const R x = km_unnormalize(cd.gx, g.W, g.align, mx);
    const R y = km_unnormalize(cd.gy, g.H, g.align, my);
    const bool live = valid && (x >= (R)-1) && (x < (R)g.W) && (y >= (R)-1) && (y < (R)g.H);  // has an in-image tap
    KmBilin<R> t;
    km_bilinear_setup(x, y, g.W, g.H, t);
    const int x0 = (int)fmaxf(fminf(km_floor(x), (R)g.W), (R)-1), y0 = (int)fmaxf(fminf(km_floor(y), (R)g.H), (R)-1);
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool in_x0 = (x0 >= X0 && x0 < X1), in_x1 = (x1 >= X0 && x1 < X1);
    const bool in_y0 = (y0 >= Y0 && y0 < Y1), in_y1 = (y1 >= Y0 && y1 < Y1);
    const bool t00 = live && t.b00 && in_x0 && in_y0, t01 = live && t.b01 && in_x1 && in_y0;
    const bool t10 = live && t.b10 && in_x0 && in_y1, t11 = live && t.b11 && in_x1 && in_y1;
    const int l00 = (y0 - Y0) * KMT_TW + (x0 - X0);
    if (finite) {
        const R w00 = t.w00 * scale, w01 = t.w01 * scale, w10 = t.w10 * scale, w11 = t.w11 * scale;
#pragma unroll
        for (int c = 0; c < KMT_CC; ++c) {
            if (c < cc) {
                int* accp = s_acc + c * (KMT_TH * KMT_TW) + l00;
                if (t00) atomicAdd(accp, __float2int_rn(w00 * go[c]));
                if (t01) atomicAdd(accp + 1, __float2int_rn(w01 * go[c]));
                if (t10) atomicAdd(accp + KMT_TW, __float2int_rn(w10 * go[c]));
                if (t11) atomicAdd(accp + KMT_TW + 1, __float2int_rn(w11 * go[c]));
            }
        }
    } else {
        // inf / NaN in grad_out, vanishing-line tiles, extreme magnification: float LDS atomics
#pragma unroll
        for (int c = 0; c < KMT_CC; ++c) {
            if (c < cc) {
                float* accp = (float*)s_acc + c * (KMT_TH * KMT_TW) + l00;
                if (t00) atomicAdd(accp, t.w00 * go[c]);
                if (t01) atomicAdd(accp + 1, t.w01 * go[c]);
                if (t10) atomicAdd(accp + KMT_TW, t.w10 * go[c]);
                if (t11) atomicAdd(accp + KMT_TW + 1, t.w11 * go[c]);
            }
        }
    }
    if (WANT_GM) {
        // the tile holding the clamped north-west tap owns q's matrix gradient
        const int ox = min(max(x0, 0), g.W - 1), oy = min(max(y0, 0), g.H - 1);
        const bool own = live && (ox >= X0 && ox < X1 && oy >= Y0 && oy < Y1);
        R gix = 0, giy = 0;
        if (own)  // measured: skipping the tap loads of non-owned pixels beats batching them (1.56 -> 1.50 ms)
        {
#pragma unroll
        for (int c = 0; c < KMT_CC; ++c) {
            if (c < cc) {
                const T* img = src_b + (size_t)(cbase + c) * src_plane;
                const R f = (g.pad == KM_PAD_FILL) ? a.fill[cbase + c] : (R)0;
                // unconditional loads (clamped indices); out-of-bounds taps do not exist in the reference's sum
                const R v00 = (R)km_ld(img + t.i00), v01 = (R)km_ld(img + t.i01), v10 = (R)km_ld(img + t.i10), v11 = (R)km_ld(img + t.i11);
                const R s00 = t.b00 ? v00 - f : (R)0, s01 = t.b01 ? v01 - f : (R)0;
                const R s10 = t.b10 ? v10 - f : (R)0, s11 = t.b11 ? v11 - f : (R)0;
                // d/dx = (ne - nw)(y1 - y) + (se - sw)(y - y0) ; d/dy = (sw - nw)(x1 - x) + (se - ne)(x - x0)
                gix += go[c] * ((s01 - s00) * t.wy1 + (s11 - s10) * t.wy0);
                giy += go[c] * ((s10 - s00) * t.wx1 + (s11 - s01) * t.wx0);
            }
        }
        }
        if (own) kmt_accumulate_gm<CM>(gm, cd, gix * mx, giy * my);
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

    // ---- G: source pixel (x, y, 1) -> (Jn, In, D), output index (j, i) = (Jn, In) / D -----------------
    int j0 = 0, j1 = g.w - 1, i0 = 0, i1 = g.h - 1;
    R mult = 16.f;  // bound on the number of output pixels whose footprint covers one source pixel
    bool fixed_ok = false;  // fixed-point accumulation is accurate enough (bounded multiplicity)
    {
        R G[9];
        // adjugate of m (un-normalised inverse: the common scale cancels in the projective divide)
        const R A0 = m[4] * m[8] - m[5] * m[7], A1 = m[2] * m[7] - m[1] * m[8], A2 = m[1] * m[5] - m[2] * m[4];
        const R A3 = m[5] * m[6] - m[3] * m[8], A4 = m[0] * m[8] - m[2] * m[6], A5 = m[2] * m[3] - m[0] * m[5];
        const R A6 = m[3] * m[7] - m[4] * m[6], A7 = m[1] * m[6] - m[0] * m[7], A8 = m[0] * m[4] - m[1] * m[3];
        // pixel -> normalised source coordinate (inverse of km_unnormalize): gn = ax * x + bx
        const R ax = g.align ? (g.W > 1 ? 2.0f / (R)(g.W - 1) : 0.0f) : 2.0f / (R)g.W;
        const R bx = g.align ? -1.0f : 1.0f / (R)g.W - 1.0f;
        const R ay = g.align ? (g.H > 1 ? 2.0f / (R)(g.H - 1) : 0.0f) : 2.0f / (R)g.H;
        const R by = g.align ? -1.0f : 1.0f / (R)g.H - 1.0f;
        const R P0 = A0 * ax, P1 = A1 * ay, P2 = A0 * bx + A1 * by + A2;
        const R P3 = A3 * ax, P4 = A4 * ay, P5 = A3 * bx + A4 * by + A5;
        const R P6 = A6 * ax, P7 = A7 * ay, P8 = A6 * bx + A7 * by + A8;
        R sj, oj, si, oi;
        kmt_index_affine<CM>(g, g.w, g.lin_lo_x, g.lin_step_x, sj, oj);
        kmt_index_affine<CM>(g, g.h, g.lin_lo_y, g.lin_step_y, si, oi);
        G[0] = sj * P0 + oj * P6; G[1] = sj * P1 + oj * P7; G[2] = sj * P2 + oj * P8;
        G[3] = si * P3 + oi * P6; G[4] = si * P4 + oi * P7; G[5] = si * P5 + oi * P8;
        G[6] = P6; G[7] = P7; G[8] = P8;

        // box of output pixels that can touch the tile (corners of the tile grown by the bilinear footprint)
        const R xs[2] = {(R)(X0 - 1), (R)X1}, ys[2] = {(R)(Y0 - 1), (R)Y1};
        R jmin = 3.0e38f, jmax = -3.0e38f, imin = 3.0e38f, imax = -3.0e38f, dmin = 3.0e38f, dmax = -3.0e38f, nmax = 0.f;
        R njx = 0.f, njy = 0.f, nix = 0.f, niy = 0.f;
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
            }
        const bool same_sign = (dmin > 0.f) || (dmax < 0.f);
        const R dabs_min = fminf(fabsf(dmin), fabsf(dmax)), dabs_max = fmaxf(fabsf(dmin), fabsf(dmax));
        const bool ok = same_sign && (dabs_min > 1e-6f * fmaxf(nmax, dabs_max)) && (jmin == jmin) && (jmax == jmax) && (imin == imin) && (imax == imax);
        if (ok) {
            const R big = 1.0e9f;
            j0 = max(0, (int)floorf(fmaxf(jmin, -big)) - 1);
            j1 = min(g.w - 1, (int)ceilf(fminf(jmax, big)) + 1);
            i0 = max(0, (int)floorf(fmaxf(imin, -big)) - 1);
            i1 = min(g.h - 1, (int)ceilf(fminf(imax, big)) + 1);
            // output pixels per source pixel: the 2x2 footprint box maps to at most (2 ex + 1)(2 ey + 1) lattice points
            const R inv_d2 = 1.0f / (dabs_min * dabs_min);
            const R ex = (njx + njy) * inv_d2 + 0.1f, ey = (nix + niy) * inv_d2 + 0.1f;
            mult = fminf((2.f * ex + 1.f) * (2.f * ey + 1.f), 1.0e6f);
            fixed_ok = mult <= 256.f;  // beyond ~7x magnification the head-room would eat the mantissa: float path
        } else {
            // tile crossed by the vanishing line: visit the whole output (correct, slower); no multiplicity bound
            mult = (R)g.w * (R)g.h;
        }
    }
    j0 = kmt_uniform(j0); j1 = kmt_uniform(j1); i0 = kmt_uniform(i0); i1 = kmt_uniform(i1);
    const int bw = j1 - j0 + 1, bh = i1 - i0 + 1;
    const bool empty = (bw <= 0 || bh <= 0);
    const float inv_bw = kmt_uniform(bw > 0 ? 1.0f / (float)bw : 0.f);
    const int hb = kmt_uniform((int)ceilf(log2f(fmaxf(mult, 1.f))) + 1);  // head-room bits
    fixed_ok = kmt_uniform((int)fixed_ok) != 0;

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
            bool exceeded = false;
            if (!empty) {
                for (int ib = i0; ib <= i1; ib += KMT_TAB) {
                    const int ie = min(i1, ib + KMT_TAB - 1);
                    __syncthreads();
                    if (tid <= ie - ib) s_v[tid] = km_base_y<R, CM>(g, ib + tid);
                    __syncthreads();
                    const int nq = bw * (ie - ib + 1);
                    int base = 0;
                    for (; base + KMT_UNROLL * 256 <= nq; base += KMT_UNROLL * 256) {
#pragma unroll
                        for (int s4 = 0; s4 < KMT_UNROLL; ++s4)
                            kmt_scatter_q<T, CM, WANT_GM>(a, m, base + s4 * 256 + tid, true, bw, inv_bw, j0, ib, tab_x, s_u, s_v, s_acc, finite,
                                                          scale, cbase, cc, src_b, gout_b, src_plane, dst_plane, X0, X1, Y0, Y1, gm_try, bound,
                                                          exceeded);
                    }
                    for (; base < nq; base += 256) {
                        const int e = base + tid;
                        kmt_scatter_q<T, CM, WANT_GM>(a, m, min(e, nq - 1), e < nq, bw, inv_bw, j0, ib, tab_x, s_u, s_v, s_acc, finite, scale,
                                                      cbase, cc, src_b, gout_b, src_plane, dst_plane, X0, X1, Y0, Y1, gm_try, bound, exceeded);
                    }
                }
            }
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
