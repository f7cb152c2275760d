// kornia_amd - gradient of the bilinear warps with respect to the (B,3,3) matrix ONLY (the image needs no gradient: a learned homography over
// fixed images, BASELINE config 5; kornia/geometry/transform/imgwarp.py:1476-1549 homography_warp, :143-174 warp_perspective, :246-290
// warp_affine, the autograd backward of their F.grid_sample with respect to the grid pushed through the grid's construction), in the SHAPE OF
// THE BOX FORWARD (km_warp.hip, km_warp_fwd_box_kernel): fp32 storage, zeros padding, RGB / grey.
//
// The gather kernel (km_warp_gm.hip) is bound by the dependent chain position -> address -> 2 x 8-byte gathers per channel -> terms of each
// group of rows (its own ablations: two thirds of its time is that chain, not the bytes), 364 us at 256 x 3 x 512^2 where the forward - the
// same two streams' worth of bytes - takes 281; the first LDS-staged form (km_warp_gm_lds_kernel: 32 x 32 tiles, 56 x 56 boxes through
// registers, box -> barrier -> positions -> sample) lost to it the way round 2's LDS forward lost to the gather forward.  This kernel takes
// what made the box forward work:
//   * a block owns a 64 x 32 region of the output, its source box (up to 80 x 40 pixels, all channels) comes by LDS-DMA - no register holds it;
//   * the box's requests and the thread's 8 x NC grad_out loads are issued FIRST, the thread's positions are computed while they fly, the
//     only wait of the block is the one in front of its barrier; 4 blocks per CU overlap their phases;
//   * every tap is an LDS read (lane = output column: neighbouring lanes, neighbouring banks); what the forward stores, this kernel loads.
// The arithmetic per pixel is km_warp_gm_rows' (same differences, same fma chain, kmg_terms, S / Sv sums, kmg_block_reduce): the same values up
// to the order of the fp32 partial sums (8 rows per thread here, 16 there) and of the fp64 atomics.
// A region whose box does not fit (rotations beyond ~7 degrees, minification) or a wave with a footprint outside its box takes
// km_warp_gm_rows' gathers: the result never depends on the box estimate.
#include <stdlib.h>

#include "km_warp_gm_rows.h"

#ifndef KMGB_PITCH
#define KMGB_PITCH 80
#define KMGB_ROWS 40
#endif
#define KMGB_TW 64
#define KMGB_TH 32
#ifndef KMGB_WAVES_PER_EU
#define KMGB_WAVES_PER_EU 4
#endif
#define KMGB_RPT (KMGB_TH / (256 / KMGB_TW))  // output rows per thread (consecutive)
#ifndef KMGB_ILP
#define KMGB_ILP 2
#endif
#ifndef KMGB_ABL
#define KMGB_ABL 0  // timing experiments only (wrong results): 1 no block reduction / atomics, 2 no sampling pass, 4 no grad_out loads
#endif

// the region once its box is known to fit: requests, positions (in-box vote), barrier, sampling.  Returns false - block-uniform, nothing added to
// the sums - when a footprint of the region lies outside the box.  FAST: the division operands of every row are in the shared-reciprocal range.
template <int CM, int NC, int ALIGN, bool FAST>
__device__ __forceinline__ bool kmgb_staged(const KmWarpGmArgs<float>& a, const float (&m)[9], const KmfBox& bx, uint32_t b, int j, int li_base, int i_base, bool col_ok,
                                            float u, const float4* s_rv, float* s_src, float (&S)[3], float (&Sv)[3]) {
    constexpr int PITCH = KMGB_PITCH, ROWS = KMGB_ROWS, RPT = KMGB_RPT;
    constexpr int ILP = FAST ? KMGB_ILP : 1;  // (the IEEE divisions of the rare path: one row at a time, or its registers spill)
    const KmWarpGeom<float>& g = a.g;
    bool done = false;
    {
        const int W = g.W, H = g.H;
        const float Wm1 = (float)(W - 1), Hm1 = (float)(H - 1), hW = (float)W / 2, hH = (float)H / 2;
        const float mx = ALIGN ? Wm1 / 2 : hW, my = ALIGN ? Hm1 / 2 : hH;
        const size_t src_plane = (size_t)H * W, dst_plane = (size_t)g.h * g.w;
        const float* __restrict__ src_b = a.src + (size_t)b * NC * src_plane;
        const float* __restrict__ gout_b = a.gout + (size_t)b * NC * dst_plane;
        // ---- requests: the box (LDS-DMA), then grad_out of this thread's pixels (unconditional loads: a pixel outside the output reads the
        //      region's first pixel and contributes nothing) ----
        kmf_stage_box_dma<NC, PITCH, ROWS>(src_b, src_plane, W, H, bx, s_src);
        float go[RPT][NC];
        bool ok[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = i_base + r;
            ok[r] = col_ok & (i < g.h);
            const uint32_t off = ok[r] ? (uint32_t)i * (uint32_t)g.w + (uint32_t)j : 0u;
#pragma unroll
            for (int c = 0; c < NC; ++c) go[r][c] = (KMGB_ABL & 4) ? (float)(off & 7u) : km_ld(km_at(gout_b + c * dst_plane, off));
        }
        // ---- positions, while the requests fly: only to know that every footprint lies in the box.  They are formed AGAIN at the sampling
        //      (the same instructions on the same operands: the same bits) - eight rows' worth of position records, held across the barrier
        //      beside the 24 registers of grad_out, would not fit the 128 registers that four blocks per CU leave a thread ----
        const KmlHalf cu = kml_col_half<CM>(m, u);
        bool inbox = true;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const float4 rv4 = s_rv[li_base + r];
            KmlHalf rv;
            rv.a = rv4.x; rv.b = rv4.y; rv.c = rv4.z;
            KmlPos p;
            kml_position<CM, FAST>(m, cu, rv, p);
            KmlTaps t;
            kml_taps(kml_unnormalize<ALIGN>(p.gx, Wm1, hW), kml_unnormalize<ALIGN>(p.gy, Hm1, hH), t);
            inbox = inbox & (kmf_in_box(t, bx) | !ok[r]);
            if (r % ILP == ILP - 1) KM_SCHED_FENCE();
        }
        KM_VMCNT0();  // (this wave's LDS-DMA requests have landed - and its grad_out loads with them)
        if (__syncthreads_and((int)inbox)) {  // block-uniform (the reduction below has a barrier)
            done = true;
#if KMGB_ABL & 2
            S[0] = go[0][0] + go[RPT - 1][NC - 1] + s_src[threadIdx.x];
            return done;
#endif
            KmlHalf cu2 = cu;
            KM_OPAQUE(cu2.a); KM_OPAQUE(cu2.b); KM_OPAQUE(cu2.c);  // (a new value to the optimiser: nothing of the first pass is kept alive for this one)
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const float4 rv4 = s_rv[li_base + r];
                KmlHalf rv;
                rv.a = rv4.x; rv.b = rv4.y; rv.c = rv4.z;
                KmlPos p;
                kml_position<CM, FAST>(m, cu2, rv, p);
                KmlTaps t;
                kml_taps(kml_unnormalize<ALIGN>(p.gx, Wm1, hW), kml_unnormalize<ALIGN>(p.gy, Hm1, hH), t);
                const int xi = ok[r] ? KM_F2I(t.xf) - bx.xs : 0, yi = ok[r] ? KM_F2I(t.yf) - bx.ys : 0;
                const float* q0 = s_src + __mul24(yi, NC * PITCH) + xi;
                const float* q1 = q0 + NC * PITCH;
                float gix = 0, giy = 0;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float s00 = q0[c * PITCH], s01 = q0[c * PITCH + 1], s10 = q1[c * PITCH], s11 = q1[c * PITCH + 1];
                    gix = km_fma(go[r][c], km_fma(s01 - s00, t.wy1, (s11 - s10) * t.wy0), gix);
                    giy = km_fma(go[r][c], km_fma(s10 - s00, t.wx1, (s11 - s01) * t.wx0), giy);
                }
                const float gx_ = ok[r] ? gix * mx : 0.0f, gy_ = ok[r] ? giy * my : 0.0f;
                float ax, ay, az;
                kmg_terms<CM, FAST>(p, gx_, gy_, ax, ay, az);
                ax = ok[r] ? ax : 0.0f; ay = ok[r] ? ay : 0.0f; az = ok[r] ? az : 0.0f;  // (a padding pixel's 0 * inf must not reach the sums)
                S[0] += ax; S[1] += ay; S[2] += az;
                Sv[0] = km_fma(ax, rv4.w, Sv[0]); Sv[1] = km_fma(ay, rv4.w, Sv[1]); Sv[2] = km_fma(az, rv4.w, Sv[2]);
                if (r % ILP == ILP - 1) KM_SCHED_FENCE();  // (ILP rows' taps in flight together, not all eight: the registers)
            }
        }
    }
    return done;
}

template <int CM, int NC, int ALIGN>
__global__ __launch_bounds__(256, KMGB_WAVES_PER_EU) void km_warp_gm_box_kernel(const KmWarpGmArgs<float> a) {
    constexpr int TW = KMGB_TW, TH = KMGB_TH, PITCH = KMGB_PITCH, ROWS = KMGB_ROWS, RPT = KMGB_RPT;
    const KmWarpGeom<float>& g = a.g;
    __shared__ double red[4][9];
    __shared__ float4 s_rv[TH];  // per row of the region: (m1 v, m4 v, m7 v, v)
    __shared__ int s_info[8];
    __shared__ __attribute__((aligned(16))) float s_src[ROWS * NC * PITCH];  // [row][channel][x]
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks, a.reverse);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    const int tid = threadIdx.x;
    const int J0 = (int)tx * TW, I0 = (int)ty * TH;
    const int j = J0 + (tid % TW);
    const int li_base = (tid / TW) * RPT;
    const int i_base = I0 + li_base;
    float m[9];
    {
        // (block-uniform, into scalar registers before anything is requested: see kmb_region)
        const float* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mp[k])));
    }
    kmf_tile_setup<CM, ALIGN, 0, TW, TH, PITCH, ROWS>(g, m, J0, I0, s_rv, s_info, true);
    __syncthreads();
    const KmfBox bx = kmf_read_box(s_info);
    const bool col_ok = j < g.w;
    const float u = km_base_x<float, CM>(g, col_ok ? j : 0);
    double* gmat_b = a.gmat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
    float S[3] = {0, 0, 0}, Sv[3] = {0, 0, 0};
    bool done = false;
    // (block-uniform.  Rows whose division operands leave the shared-reciprocal range - kml_row_guard: extreme matrices - take the gathers: with the
    // IEEE divisions in it as well, this kernel's registers spill)
    if (bx.staged && bx.fast) done = kmgb_staged<CM, NC, ALIGN, true>(a, m, bx, b, j, li_base, i_base, col_ok, u, s_rv, s_src, S, Sv);
    if (!done) {  // the box does not fit / a footprint is not covered: gathers from global memory (consecutive rows: PH = 1)
        if (bx.fast) km_warp_gm_rows<float, CM, NC, ALIGN, true, RPT, 1>(a, m, s_rv, b, j, li_base, i_base, S, Sv);
        else km_warp_gm_rows<float, CM, NC, ALIGN, false, RPT, 1>(a, m, s_rv, b, j, li_base, i_base, S, Sv);
    }
#if KMGB_ABL & 1
    {
        float keep = ((S[0] + S[1]) + S[2]) + ((Sv[0] + Sv[1]) + Sv[2]);
        KM_OPAQUE(keep);
        if (keep == 12345.678f) gmat_b[0] = (double)u;
        (void)red;
    }
#else
    kmg_block_reduce<CM>(S, Sv, u, gmat_b, red);
#endif
}

template <int CM, int NC>
static void kmgb_launch_nc(const KmWarpGmArgs<float>& a, hipStream_t s) {
    if (a.g.align) hipLaunchKernelGGL((km_warp_gm_box_kernel<CM, NC, 1>), dim3(a.nblocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((km_warp_gm_box_kernel<CM, NC, 0>), dim3(a.nblocks), dim3(256), 0, s, a);
}

// Launches the box form and returns 1 when it takes the case (fp32 storage, zeros padding, C in {1, 3}, 16-byte rows), 0 otherwise
// (km_warp_gm.hip then runs the gather kernel).  `args`: a KmWarpGmArgs<float> with the geometry filled in.
int km_warp_gm_box_try(const void* args, int coord_mode, hipStream_t s) {
    KmWarpGmArgs<float> a = *reinterpret_cast<const KmWarpGmArgs<float>*>(args);
    const KmWarpGeom<float>& g = a.g;
    if (!(g.C == 3 || g.C == 1) || g.pad != KM_PAD_ZEROS || (g.W & 3) != 0 || ((uintptr_t)a.src & 15) != 0 || g.W < 4) return 0;
    if (!((uint64_t)g.H * g.W * 4 < (1ull << 32) && (uint64_t)g.h * g.w * 4 < (1ull << 32) && g.W < (1 << 23) && g.H < (1 << 23))) return 0;
    a.tiles_x = (uint32_t)((g.w + KMGB_TW - 1) / KMGB_TW);
    a.tiles_y = (uint32_t)((g.h + KMGB_TH - 1) / KMGB_TH);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)g.B;
    if (nb == 0 || nb >= (1ull << 31)) return 0;
    a.nblocks = (uint32_t)nb;
    switch (coord_mode) {
        case KM_COORD_PERSPECTIVE: g.C == 3 ? kmgb_launch_nc<KM_COORD_PERSPECTIVE, 3>(a, s) : kmgb_launch_nc<KM_COORD_PERSPECTIVE, 1>(a, s); break;
        case KM_COORD_AFFINE: g.C == 3 ? kmgb_launch_nc<KM_COORD_AFFINE, 3>(a, s) : kmgb_launch_nc<KM_COORD_AFFINE, 1>(a, s); break;
        default: g.C == 3 ? kmgb_launch_nc<KM_COORD_HOMOGRAPHY, 3>(a, s) : kmgb_launch_nc<KM_COORD_HOMOGRAPHY, 1>(a, s); break;
    }
    return 1;
}
