// kornia_amd - gradient of the bilinear warps with respect to the (B,3,3) matrix, for gfx950.
//
// d loss / d M  =  sum over output pixels q of  J_q^T (dL/dx_q, dL/dy_q),  with
//   dL/dx_q = sum_c grad_out[q,c] * ((ne - nw)(y1 - y) + (se - sw)(y - y0))     (image gradient at the sample)
//   dL/dy_q = sum_c grad_out[q,c] * ((sw - nw)(x1 - x) + (se - ne)(x - x0))
// and J_q the Jacobian of the sampling position with respect to the matrix entries (SURVEY.md A.6;
// reference: autograd through kornia/geometry/transform/imgwarp.py:157-174 / linalg.py:219-239).
//
// This is an OUTPUT-pixel-centric reduction: it has the access pattern of the forward warp (coalesced
// grad_out reads, gathered src taps) and none of the scatter of grad_src, so it runs as its own kernel
// shaped like the specialised forward (km_warp_fwd_bz_kernel) instead of riding in the tile-owner
// scatter kernel, where it cost more than the scatter itself (owner predication, 118 VGPRs).
//   * a wave owns a 64-wide x KMG_ROWS-tall strip of the output, lane = column;
//   * per-thread partial sums in fp32 (<= KMG_ROWS * C terms), wave + block reduction and the global
//     accumulation in fp64 (9 atomics per block);
//   * HBM traffic: read grad_out once + src once = 2e bytes / element.
#include <stdlib.h>

#include "km_warp_gm_rows.h"

template <typename T, int CM, int NC, int ALIGN>  // NC = 3 / 1: RGB / grey unrolled ; NC = 0: runtime channel loop
__global__ KMG_BOUNDS void km_warp_gm_kernel(const KmWarpGmArgs<T> a) {
    const KmWarpGeom<float>& g = a.g;
    __shared__ double red[4][9];
    __shared__ float4 s_rv[KMG_TILE_H];  // per row of the tile: (m1 v, m4 v, m7 v, v)
    __shared__ int s_fast;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks, a.reverse);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    // a wave instruction covers a KMG_PATCH_W x (64 / KMG_PATCH_W) patch of the output (see km_warp_fwd_lean_kernel)
    constexpr int PW = KMG_PATCH_W, PH = 64 / PW, WA = 64 / PW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = (int)tx * KMG_TILE_W + (wave % WA) * PW + (lane % PW);
    const int li_base = (wave / WA) * (PH * KMG_ROWS) + lane / PW;  // row r of this thread sits at tile row li_base + r * PH
    const int i_base = (int)ty * KMG_TILE_H + li_base;

    float m[9];
    {
        const float* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = mp[k];
    }
    static_assert(KMG_TILE_H <= 64, "the row table is filled by one wave");
    if (wave == 0) {  // row halves of the numerators + the per-row division guard, combined over the tile's rows
        bool okr = true;
        if (lane < KMG_TILE_H) {
            const float v = km_base_y<float, CM>(g, (int)ty * KMG_TILE_H + lane);
            const KmlHalf h = kml_row_half<CM>(m, v);
            s_rv[lane] = make_float4(h.a, h.b, h.c, v);
            okr = kml_row_guard<CM>(g, m, v);
        }
        const bool all_ok = __all(okr);
        if (lane == 0) s_fast = all_ok ? 1 : 0;
    }
    __syncthreads();

    // u is fixed per thread (lane = column): accumulate S = sum(ax, ay, az) and Sv = sum(v * (ax, ay, az)) over the
    // thread's rows and multiply by u once at the end
    float S[3] = {0, 0, 0}, Sv[3] = {0, 0, 0};
    if (__builtin_amdgcn_readfirstlane(s_fast))
        km_warp_gm_rows<T, CM, NC, ALIGN, true>(a, m, s_rv, b, j, li_base, i_base, S, Sv);
    else
        km_warp_gm_rows<T, CM, NC, ALIGN, false>(a, m, s_rv, b, j, li_base, i_base, S, Sv);
    kmg_block_reduce<CM>(S, Sv, km_base_x<float, CM>(g, j < g.w ? j : 0), a.gmat + (size_t)(g.B_M == 1 ? 0 : b) * 9, red);
}

// LDS-staged matrix gradient (RGB / grey): the forward-shaped reduction with the source taps read from a staged box instead
// of gathered from global memory (km_warp_stage.h) - the gather kernel above is bound by the texture-address path like the
// gather forward.  Per pixel the memory pipeline sees NC coalesced grad_out loads instead of those + 2 NC gathers.
template <typename T, int CM, int NC, int ALIGN>
__global__ __launch_bounds__(256) void km_warp_gm_lds_kernel(const KmWarpGmArgs<T> a) {
    const KmWarpGeom<float>& g = a.g;
    __shared__ double red[4][9];
    __shared__ float4 s_rv[KMF_T];  // per row of the tile: (m1 v, m4 v, m7 v, v)
    __shared__ int s_info[8];
    __shared__ __attribute__((aligned(16))) float s_src[KMF_ROWS * NC * KMF_PITCH];
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    const int tid = threadIdx.x;
    const int j = (int)tx * KMF_T + (tid % KMF_T);
    const int li_base = tid / KMF_T;
    const int i_base = (int)ty * KMF_T + li_base;

    float m[9];
    {
        const float* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = mp[k];
    }
    kmf_tile_setup<CM, ALIGN>(g, m, (int)tx * KMF_T, (int)ty * KMF_T, s_rv, s_info, true);
    __syncthreads();
    const KmfBox bx = kmf_read_box(s_info);
    const bool col_ok = j < g.w;
    const float u = km_base_x<float, CM>(g, col_ok ? j : 0);
    double* gmat_b = a.gmat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
    float S[3] = {0, 0, 0}, Sv[3] = {0, 0, 0};
    bool done = false;
    if (bx.staged) {  // block-uniform
        const int W = g.W, H = g.H;
        const float Wm1 = (float)(W - 1), Hm1 = (float)(H - 1), hW = (float)W / 2, hH = (float)H / 2;
        const float mx = ALIGN ? Wm1 / 2 : hW, my = ALIGN ? Hm1 / 2 : hH;
        const size_t src_plane = (size_t)H * W, dst_plane = (size_t)g.h * g.w;
        const T* __restrict__ src_b = a.src + (size_t)b * NC * src_plane;
        const T* __restrict__ gout_b = a.gout + (size_t)b * NC * dst_plane;
        const bool is_fill = (g.pad == KM_PAD_FILL);
        float oob[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) oob[c] = is_fill ? a.fill[c] : 0.f;  // (v - fill) of a tap outside the image is 0: it does not exist
        // grad_out of this thread's pixels: issued before the staging so that both are in flight together
        float go[KMF_RPT][NC];
        bool ok[KMF_RPT];
#pragma unroll
        for (int r = 0; r < KMF_RPT; ++r) {
            const int i = i_base + r * KMF_RSTEP;
            ok[r] = col_ok & (i < g.h);
            const uint32_t off = ok[r] ? (uint32_t)i * (uint32_t)g.w + (uint32_t)j : 0u;
#pragma unroll
            for (int c = 0; c < NC; ++c) go[r][c] = (float)km_ld(km_at(gout_b + c * dst_plane, off));
        }
        kmf_stage_box<T, NC>(src_b, src_plane, W, H, bx, s_src, oob);
        __syncthreads();
        const KmlHalf cu = kml_col_half<CM>(m, u);
        KmlPos p[KMF_RPT];
        KmlTaps t[KMF_RPT];
        float vrow[KMF_RPT];
        bool inbox = true;
#pragma unroll
        for (int r = 0; r < KMF_RPT; ++r) {
            const float4 rv4 = s_rv[li_base + r * KMF_RSTEP];
            KmlHalf rv;
            rv.a = rv4.x; rv.b = rv4.y; rv.c = rv4.z;
            vrow[r] = rv4.w;
            if (bx.fast) kml_position<CM, true>(m, cu, rv, p[r]);
            else kml_position<CM, false>(m, cu, rv, p[r]);
            kml_taps(kml_unnormalize<ALIGN>(p[r].gx, Wm1, hW), kml_unnormalize<ALIGN>(p[r].gy, Hm1, hH), t[r]);
            inbox = inbox & (kmf_in_box(t[r], bx) | !ok[r]);
        }
        if (__syncthreads_and((int)inbox)) {  // block-uniform (the reduction below has a barrier)
            done = true;
#pragma unroll
            for (int r = 0; r < KMF_RPT; ++r) {
                const float* q0 = kmf_tap_ptr<NC>(s_src, t[r], bx, ok[r]);
                const float* q1 = q0 + NC * KMF_PITCH;
                float gix = 0, giy = 0;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float s00 = q0[c * KMF_PITCH], s01 = q0[c * KMF_PITCH + 1], s10 = q1[c * KMF_PITCH], s11 = q1[c * KMF_PITCH + 1];
                    if (is_fill) {  // same rounding sequence as the oracle: (v - fill) first
                        const float f = a.fill[c];
                        s00 -= f; s01 -= f; s10 -= f; s11 -= f;
                    }
                    gix = km_fma(go[r][c], km_fma(s01 - s00, t[r].wy1, (s11 - s10) * t[r].wy0), gix);
                    giy = km_fma(go[r][c], km_fma(s10 - s00, t[r].wx1, (s11 - s01) * t[r].wx0), giy);
                }
                const float gx_ = ok[r] ? gix * mx : 0.0f, gy_ = ok[r] ? giy * my : 0.0f;
                float ax, ay, az;
                if (bx.fast) kmg_terms<CM, true>(p[r], gx_, gy_, ax, ay, az);
                else kmg_terms<CM, false>(p[r], gx_, gy_, ax, ay, az);
                ax = ok[r] ? ax : 0.0f; ay = ok[r] ? ay : 0.0f; az = ok[r] ? az : 0.0f;  // (a padding pixel's 0 * inf must not reach the sums)
                S[0] += ax; S[1] += ay; S[2] += az;
                Sv[0] = km_fma(ax, vrow[r], Sv[0]); Sv[1] = km_fma(ay, vrow[r], Sv[1]); Sv[2] = km_fma(az, vrow[r], Sv[2]);
            }
        }
    }
    if (!done) {  // the box does not fit / a footprint is not covered: gathers from global memory
        if (bx.fast) km_warp_gm_rows<T, CM, NC, ALIGN, true, KMF_RPT, KMF_RSTEP>(a, m, s_rv, b, j, li_base, i_base, S, Sv);
        else km_warp_gm_rows<T, CM, NC, ALIGN, false, KMF_RPT, KMF_RSTEP>(a, m, s_rv, b, j, li_base, i_base, S, Sv);
    }
    kmg_block_reduce<CM>(S, Sv, u, gmat_b, red);
}

template <typename T, int CM, int NC>
static void kmg_launch_nc(const KmWarpGmArgs<T>& a, hipStream_t s) {
    if (a.g.align)
        hipLaunchKernelGGL((km_warp_gm_kernel<T, CM, NC, 1>), dim3(a.nblocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((km_warp_gm_kernel<T, CM, NC, 0>), dim3(a.nblocks), dim3(256), 0, s, a);
}
static int kmg_algo() {  // KM_WARP_GM_ALGO: "generic" | "lds" (LDS-staged kernel, A/B timing) | default: the lean gather kernel
    return km_config().warp_gm_algo == 2 ? 1 : 0;
}
template <typename T, int CM, int NC>
static void kmg_lds_launch_nc(const KmWarpGmArgs<T>& a, hipStream_t s) {
    KmWarpGmArgs<T> b = a;
    b.tiles_x = (uint32_t)((a.g.w + KMF_T - 1) / KMF_T);
    b.tiles_y = (uint32_t)((a.g.h + KMF_T - 1) / KMF_T);
    b.nblocks = b.tiles_x * b.tiles_y * (uint32_t)a.g.B;
    if (a.g.align)
        hipLaunchKernelGGL((km_warp_gm_lds_kernel<T, CM, NC, 1>), dim3(b.nblocks), dim3(256), 0, s, b);
    else
        hipLaunchKernelGGL((km_warp_gm_lds_kernel<T, CM, NC, 0>), dim3(b.nblocks), dim3(256), 0, s, b);
}
template <typename T, int CM>
static int kmg_launch(const KmWarpGmArgs<T>& a, hipStream_t s) {
    const uint64_t nb = (uint64_t)((a.g.w + KMF_T - 1) / KMF_T) * (uint64_t)((a.g.h + KMF_T - 1) / KMF_T) * (uint64_t)a.g.B;
    if ((a.g.C == 3 || a.g.C == 1) && (a.g.W & 3) == 0 && ((uintptr_t)a.src % (4 * sizeof(T))) == 0 && nb < (1ull << 31) && kmg_algo() == 1) {
        if (a.g.C == 3) kmg_lds_launch_nc<T, CM, 3>(a, s);
        else kmg_lds_launch_nc<T, CM, 1>(a, s);
        return km_check_launch("km_warp2d_bwd(matrix gradient)");
    }
    if (a.g.C == 3) kmg_launch_nc<T, CM, 3>(a, s);
    else if (a.g.C == 1) kmg_launch_nc<T, CM, 1>(a, s);
    else kmg_launch_nc<T, CM, 0>(a, s);
    return km_check_launch("km_warp2d_bwd(matrix gradient)");
}

// the box form (km_warp_gm_box.hip): launches and returns 1 when it takes the case, 0 otherwise
int km_warp_gm_box_try(const void* args, int coord_mode, hipStream_t s);

template <typename T>
static int kmg_run(const void* gout, const void* src, const void* mat, double* gmat, int B, int C, int H, int W, int h, int w, int B_M,
                   int coord_mode, int norm_coords, int pad, int align, const void* fill, hipStream_t s) {
    KmWarpGmArgs<T> a;
    a.src = (const T*)src; a.gout = (const T*)gout; a.mat = (const float*)mat; a.gmat = gmat; a.fill = (const float*)fill;
    KmWarpGeom<float>& g = a.g;
    km_geom_init(g, B, C, H, W, h, w, B_M, coord_mode, norm_coords, KM_INTERP_BILINEAR, pad, align);
    a.tiles_x = (uint32_t)((w + KMG_TILE_W - 1) / KMG_TILE_W);
    a.tiles_y = (uint32_t)((h + KMG_TILE_H - 1) / KMG_TILE_H);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B;
    KM_REQUIRE(nb < (1ull << 31), "km_warp2d_bwd: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    a.reverse = km_traversal_next(s);
    if constexpr (sizeof(T) == 4) {  // fp32 storage, zeros padding, RGB / grey: the box form unless KM_WARP_GM_ALGO says otherwise
        if (km_config().warp_gm_algo == 0 && km_warp_gm_box_try(&a, coord_mode, s)) return km_check_launch("km_warp2d_bwd(matrix gradient, box)");
    }
    switch (coord_mode) {
        case KM_COORD_PERSPECTIVE: return kmg_launch<T, KM_COORD_PERSPECTIVE>(a, s);
        case KM_COORD_AFFINE: return kmg_launch<T, KM_COORD_AFFINE>(a, s);
        default: return kmg_launch<T, KM_COORD_HOMOGRAPHY>(a, s);
    }
}

// 1 if this kernel computes the matrix gradient for these modes (bilinear, zeros / fill padding, fp32 compute)
int km_warp_gm_supported(int interp, int pad, int dtype, int H, int W, int h, int w) {
    if (km_config().warp_gm_algo == 1) return 0;  // KM_WARP_GM_ALGO=generic: the atomic scatter kernel computes it (A/B timing)
    if (!(interp == KM_INTERP_BILINEAR && (pad == KM_PAD_ZEROS || pad == KM_PAD_FILL) && dtype != KM_F64)) return 0;
    if (W < 2) return 0;  // the pair loads need two columns
    // 32-bit byte offsets inside a plane
    return ((uint64_t)H * W * 4 < (1ull << 32) && (uint64_t)h * w * 4 < (1ull << 32)) ? 1 : 0;
}

int km_warp_gm_run(const void* gout, const void* src, const void* mat, double* gmat, int B, int C, int H, int W, int h, int w, int B_M,
                   int coord_mode, int norm_coords, int pad, int align, const void* fill, int dtype, hipStream_t s) {
    switch (dtype) {
        case KM_F32: return kmg_run<float>(gout, src, mat, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, s);
        case KM_BF16: return kmg_run<km_bf16>(gout, src, mat, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, s);
        default: return kmg_run<km_f16>(gout, src, mat, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, s);
    }
}
