// kornia_amd - warp kernels (warp_perspective / warp_affine / homography_warp) for gfx950.
//
// One launch does what the reference does with ~25 elementwise launches + grid_sample
// (kornia/geometry/transform/imgwarp.py:157-174, :271-290, :1541-1546): the sampling grid is
// generated in registers from the (B,3,3) matrix and never touches HBM.
//
// Work decomposition (forward and generic backward):
//   * block = 4 waves; a wave owns a 64-wide x KM_ROWS-tall strip of the OUTPUT image; lane l
//     owns column x0+l, so every tap load / store instruction of a wave touches a contiguous run
//     of ~64 pixels (coalesced NCHW reads; consecutive rows of the strip re-use the same source
//     rows out of L1).
//   * the per-pixel coordinate (two IEEE divisions for the projective case) is computed once and
//     shared by all C channels.
//   * blockIdx is remapped so that one XCD (private L2) processes whole images.
// HBM roofline: forward moves 2e bytes / element (read src once, write dst once), backward 3e
// (read grad_out, read src, write grad_src) - see DESIGN.md.
#include <stdlib.h>

#include "km_warp_args.h"
#include "km_warp_stage.h"

#ifndef KM_ROWS
#define KM_ROWS 4   // output rows per thread
#endif
#define KM_TILE_W 64
#define KM_TILE_H (4 * KM_ROWS)
#ifndef KM_PATCH_W
#define KM_PATCH_W 32  // output columns covered by one wave instruction of the specialised forward (64, 32 or 16);
                       // measured at 256x3x512^2: 64 -> 0.434 ms (0.78 at 20 deg rotation), 32 -> 0.422 (0.67), 16 -> 0.492 (0.64)
#endif

template <typename T, int CM, int INTERP>
__global__ __launch_bounds__(256) void km_warp_fwd_kernel(const KmWarpArgs<T> a) {
    typedef typename KmTraits<T>::R R;
    const KmWarpGeom<R>& g = a.g;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;

    // 32 x 2 output patch per wave instruction (see km_warp_fwd_bz_kernel): compact gathers under rotation
    constexpr int PW = KM_PATCH_W, PH = 64 / PW, WA = 64 / PW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = (int)tx * KM_TILE_W + (wave % WA) * PW + (lane % PW);
    const int li_base = (wave / WA) * (PH * KM_ROWS) + lane / PW;  // this thread's row r sits at tile row li_base + r * PH
    const int i_base = (int)ty * KM_TILE_H + li_base;
    if (a.apply && !a.apply[b]) {  // block-uniform: this sample is not transformed
        km_fwd_copy_rows<T>(a, b, j, i_base, PH, KM_ROWS);
        return;
    }
    __shared__ R s_v[KM_TILE_H];
    if (threadIdx.x < KM_TILE_H) s_v[threadIdx.x] = km_base_y<R, CM>(g, (int)ty * KM_TILE_H + (int)threadIdx.x);
    __syncthreads();
    if (j >= g.w) return;

    R m[9];
    {
        const R* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = (CM == KM_COORD_GRID) ? (R)0 : mp[k];  // grid mode has no matrix
    }
    const int spad = (g.pad == KM_PAD_FILL) ? KM_PAD_ZEROS : g.pad;
    const size_t src_plane = (size_t)g.H * g.W, dst_plane = (size_t)g.h * g.w;
    const T* src_b = a.src + (size_t)b * g.C * src_plane;
    T* dst_b = a.dst + (size_t)b * g.C * dst_plane;
    const R u = km_base_x<R, CM>(g, j);

#pragma unroll
    for (int r = 0; r < KM_ROWS; ++r) {
        const int i = i_base + r * PH;
        if (i >= g.h) break;
        const R v = s_v[li_base + r * PH];  // row base coordinate (one IEEE divide per row per block, not per lane)
        KmCoord<R> cd;
        km_gen_coord<R, CM>(m, u, v, cd);
        if (CM == KM_COORD_GRID) km_grid_coord(a.grid, g, b, i, j, cd);
        R mx, my, gdx, gdy;
        R x = km_unnormalize(cd.gx, g.W, g.align, mx);
        R y = km_unnormalize(cd.gy, g.H, g.align, my);
        T* out_px = dst_b + (size_t)i * g.w + j;

        if (INTERP == KM_INTERP_BILINEAR) {
            x = km_compute_coord(x, g.W, spad, g.align, gdx);
            y = km_compute_coord(y, g.H, spad, g.align, gdy);
            KmBilin<R> t;
            km_bilinear_setup(x, y, g.W, g.H, t);
            R inv_mask = 0;
            if (g.pad == KM_PAD_FILL) {
                // 1 - grid_sample(ones): imgwarp.py:316
                inv_mask = (R)1 - km_bilinear_ones(t);
            }
            if (__all(t.b00 && t.b01 && t.b10 && t.b11)) {
                // whole wave samples strictly inside the image (the common case): no masking, same fma chain
                for (int c = 0; c < g.C; ++c) {
                    const T* img = src_b + (size_t)c * src_plane;
                    const R v00 = km_ld(img + t.i00), v01 = km_ld(img + t.i01);
                    const R v10 = km_ld(img + t.i10), v11 = km_ld(img + t.i11);
                    R acc = km_fma(v00, t.w00, (R)0);
                    acc = km_fma(v01, t.w01, acc);
                    acc = km_fma(v10, t.w10, acc);
                    acc = km_fma(v11, t.w11, acc);
                    if (g.pad == KM_PAD_FILL) acc = acc + inv_mask * a.fill[c];
                    km_st(out_px + (size_t)c * dst_plane, acc);
                }
            } else {
                for (int c = 0; c < g.C; ++c) {
                    const T* img = src_b + (size_t)c * src_plane;
                    const R v00 = km_ld(img + t.i00), v01 = km_ld(img + t.i01);
                    const R v10 = km_ld(img + t.i10), v11 = km_ld(img + t.i11);
                    R acc = km_bilinear_masked(t, (R)v00, (R)v01, (R)v10, (R)v11);  // (out-of-bounds taps: zeros that are still multiplied - ATen's CPU rule)
                    if (g.pad == KM_PAD_FILL) acc = acc + inv_mask * a.fill[c];
                    km_st(out_px + (size_t)c * dst_plane, acc);
                }
            }
        } else if (INTERP == KM_INTERP_NEAREST) {
            x = km_compute_coord(x, g.W, spad, g.align, gdx);
            y = km_compute_coord(y, g.H, spad, g.align, gdy);
            const R xr = km_rint(x), yr = km_rint(y);
            const bool inb = (xr >= (R)0) && (xr <= (R)(g.W - 1)) && (yr >= (R)0) && (yr <= (R)(g.H - 1));
            const int idx = inb ? (int)yr * g.W + (int)xr : 0;
            for (int c = 0; c < g.C; ++c) {
                R acc = inb ? km_ld(src_b + (size_t)c * src_plane + idx) : (R)0;
                if (g.pad == KM_PAD_FILL) acc = acc + ((R)1 - (inb ? (R)1 : (R)0)) * a.fill[c];
                km_st(out_px + (size_t)c * dst_plane, acc);
            }
        } else {
            const R xf = km_floor(x), yf = km_floor(y);
            R cx[4], cy[4];
            km_cubic_coeffs(x - xf, cx);
            km_cubic_coeffs(y - yf, cy);
            // all 16 taps of every lane inside the image: every padding transform is the identity on them, and the
            // four taps of a row are contiguous -> 4 wide loads per channel instead of 16 predicated ones
            const bool interior = (xf >= (R)1) && (xf <= (R)(g.W - 3)) && (yf >= (R)1) && (yf <= (R)(g.H - 3));
            if (__all(interior)) {
                const int base = ((int)yf - 1) * g.W + ((int)xf - 1);
                R inv_mask = 0;
                if (g.pad == KM_PAD_FILL) {  // same expression as the general case with every tap present
                    const R one = 1;
                    const R r1 = one * cx[0] + one * cx[1] + one * cx[2] + one * cx[3];
                    inv_mask = (R)1 - (r1 * cy[0] + r1 * cy[1] + r1 * cy[2] + r1 * cy[3]);
                }
                for (int c = 0; c < g.C; ++c) {
                    const T* img = src_b + (size_t)c * src_plane + base;
                    R rows[4];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        R tt[4];
                        km_ld4u(img + rr * g.W, tt);
                        rows[rr] = tt[0] * cx[0] + tt[1] * cx[1] + tt[2] * cx[2] + tt[3] * cx[3];
                    }
                    R acc = rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3];
                    if (g.pad == KM_PAD_FILL) acc = acc + inv_mask * a.fill[c];
                    km_st(out_px + (size_t)c * dst_plane, acc);
                }
                continue;
            }
            int idx[4][4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int q = 0; q < 4; ++q) idx[rr][q] = km_tap_index(xf - 1 + q, yf - 1 + rr, g.W, g.H, spad, g.align);
            R inv_mask = 0;
            if (g.pad == KM_PAD_FILL) {
                R rows[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    R tt[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) tt[q] = idx[rr][q] >= 0 ? (R)1 : (R)0;
                    rows[rr] = tt[0] * cx[0] + tt[1] * cx[1] + tt[2] * cx[2] + tt[3] * cx[3];
                }
                inv_mask = (R)1 - (rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3]);
            }
            for (int c = 0; c < g.C; ++c) {
                const T* img = src_b + (size_t)c * src_plane;
                R rows[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    R tt[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) tt[q] = idx[rr][q] >= 0 ? km_ld(img + idx[rr][q]) : (R)0;
                    rows[rr] = tt[0] * cx[0] + tt[1] * cx[1] + tt[2] * cx[2] + tt[3] * cx[3];
                }
                R acc = rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3];
                if (g.pad == KM_PAD_FILL) acc = acc + inv_mask * a.fill[c];
                km_st(out_px + (size_t)c * dst_plane, acc);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Bilinear + zeros forward for the cases the lean kernel below does not take (explicit grids, fp64): any coordinate mode, any dtype.
// Same arithmetic as the generic kernel (bit-identical results) with everything that costs scalar/branch
// instructions per pixel removed: no padding-mode dispatch, channel count known at compile time for RGB,
// rows predicated instead of early-exited, (x0, x0+1) fetched with one load when the whole wave samples
// inside the image.  Ablation on MI355X (256x3x512^2): ALU/issue alone 0.29 ms, loads +0.12, stores +0.11,
// barely overlapped in the generic kernel - instruction count is what bounds this kernel, not HBM.
template <typename T, int CM, int NC>  // NC = 3 / 1: RGB / grey unrolled ; NC = 0: runtime channel loop
__global__ __launch_bounds__(256) void km_warp_fwd_bz_kernel(const KmWarpArgs<T> a) {
    typedef typename KmTraits<T>::R R;
    const KmWarpGeom<R>& g = a.g;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    // One wave instruction covers a KM_PATCH_W x (64 / KM_PATCH_W) patch of the output rather than a 64 x 1 row: under
    // rotation the taps of a 64 x 1 row are spread over up to 64 * sin(angle) source rows (one cache line each), a
    // squarer patch keeps them within ~(PW sin + PH cos) rows.  Stores stay KM_PATCH_W * 4-byte contiguous runs.
    constexpr int PW = KM_PATCH_W, PH = 64 / PW, WA = 64 / PW;  // patch, waves across the 64-wide tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = (int)tx * KM_TILE_W + (wave % WA) * PW + (lane % PW);
    const int li_base = (wave / WA) * (PH * KM_ROWS) + lane / PW;  // row inside the tile of this thread's row r: li_base + r * PH
    const int i_base = (int)ty * KM_TILE_H + li_base;
    if (a.apply && !a.apply[b]) {  // block-uniform: this sample is not transformed
        km_fwd_copy_rows<T>(a, b, j, i_base, PH, KM_ROWS);
        return;
    }
    __shared__ R s_v[KM_TILE_H];
    if (threadIdx.x < KM_TILE_H) s_v[threadIdx.x] = km_base_y<R, CM>(g, (int)ty * KM_TILE_H + (int)threadIdx.x);
    __syncthreads();
    if (j >= g.w) return;

    R m[9];
    {
        const R* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = (CM == KM_COORD_GRID) ? (R)0 : mp[k];  // grid mode has no matrix
    }
    const int W = g.W, H = g.H, align = g.align;
    const int C = (NC > 0) ? NC : g.C;
    const size_t src_plane = (size_t)H * W, dst_plane = (size_t)g.h * g.w;
    const T* __restrict__ src_b = a.src + (size_t)b * C * src_plane;
    T* __restrict__ dst_b = a.dst + (size_t)b * C * dst_plane;
    const R u = km_base_x<R, CM>(g, j);

    // The kernel is bound by memory latency, not bandwidth or ALU (38 VGPRs, one row's 6 loads in flight per wave when the
    // rows are processed one after the other): all KM_ROWS sampling positions are computed first, then - when every lane of
    // the wave samples inside the image for all of them - all their loads are issued back to back before the first use.
    KmBilin<R> t[KM_ROWS];
    bool row_ok[KM_ROWS];
    bool inside = true;
#pragma unroll
    for (int r = 0; r < KM_ROWS; ++r) {
        const int i = i_base + r * PH;
        row_ok[r] = i < g.h;
        KmCoord<R> cd;
        km_gen_coord<R, CM>(m, u, s_v[li_base + r * PH], cd);
        if (CM == KM_COORD_GRID) km_grid_coord(a.grid, g, b, row_ok[r] ? i : 0, j, cd);
        R mx, my;
        const R x = km_unnormalize(cd.gx, W, align, mx);
        const R y = km_unnormalize(cd.gy, H, align, my);
        km_bilinear_setup(x, y, W, H, t[r]);
        inside = inside && t[r].b00 && t[r].b01 && t[r].b10 && t[r].b11;
    }
    if (NC > 0 && __all(inside)) {
        constexpr int NCC = NC > 0 ? NC : 1;
        R v[KM_ROWS][NCC][4];
#pragma unroll
        for (int r = 0; r < KM_ROWS; ++r)
#pragma unroll
            for (int c = 0; c < NCC; ++c) {
                km_ld2(src_b + c * src_plane + t[r].i00, v[r][c][0], v[r][c][1]);
                km_ld2(src_b + c * src_plane + t[r].i10, v[r][c][2], v[r][c][3]);
            }
#pragma unroll
        for (int r = 0; r < KM_ROWS; ++r) {
            T* __restrict__ out_px = dst_b + (size_t)(row_ok[r] ? i_base + r * PH : 0) * g.w + j;
#pragma unroll
            for (int c = 0; c < NCC; ++c) {
                const R acc = km_fma(v[r][c][3], t[r].w11, km_fma(v[r][c][2], t[r].w10, km_fma(v[r][c][1], t[r].w01, km_fma(v[r][c][0], t[r].w00, (R)0))));
                if (row_ok[r]) km_st(out_px + c * dst_plane, acc);
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < KM_ROWS; ++r) {
        T* __restrict__ out_px = dst_b + (size_t)(row_ok[r] ? i_base + r * PH : 0) * g.w + j;
        const KmBilin<R>& tr = t[r];
        if (__all(tr.b00 && tr.b01 && tr.b10 && tr.b11)) {
            for (int c = 0; c < C; ++c) {
                const T* img = src_b + (size_t)c * src_plane;
                R v00, v01, v10, v11;
                km_ld2(img + tr.i00, v00, v01);
                km_ld2(img + tr.i10, v10, v11);
                const R acc = km_fma(v11, tr.w11, km_fma(v10, tr.w10, km_fma(v01, tr.w01, km_fma(v00, tr.w00, (R)0))));
                if (row_ok[r]) km_st(out_px + (size_t)c * dst_plane, acc);
            }
        } else {
            for (int c = 0; c < C; ++c) {
                const T* img = src_b + (size_t)c * src_plane;
                const R v00 = km_ld(img + tr.i00), v01 = km_ld(img + tr.i01);
                const R v10 = km_ld(img + tr.i10), v11 = km_ld(img + tr.i11);
                R acc = km_bilinear_masked(tr, (R)v00, (R)v01, (R)v10, (R)v11);  // (out-of-bounds taps: zeros that are still multiplied - ATen's CPU rule)
                if (row_ok[r]) km_st(out_px + (size_t)c * dst_plane, acc);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Specialised forward for the hot configuration: bilinear + zeros padding, fp32 compute (any coordinate mode, fp32 / bf16 /
// f16 storage).  Same arithmetic as the generic kernel, bit for bit, on the lean front end of km_lean.h: the kernel is bound
// by the NUMBER of vector instructions per pixel (ablation on MI355X, 256x3x512^2: ALU/issue alone 0.29 ms, loads +0.12,
// stores +0.11 with the first version's 107 VALU instructions per pixel, 24 of them two IEEE divisions and 9 64-bit address
// computations), so everything per pixel that is not arithmetic the reference performs is gone:
//   * row halves of the numerators (m1 v, m4 v, m7 v) come from a per-block LDS table, column halves are per-thread constants;
//   * both quotients share one refined reciprocal when the block's operands are in the safe range (kml_div_guard's test,
//     evaluated per row by the threads that fill the table);
//   * one inside / not-inside decision for the 2 x 2 footprint; all KM_ROWS rows' loads are issued back to back when the
//     whole wave samples inside the image, with 32-bit offsets from wave-uniform plane bases (saddr form);
//   * (x0, x0 + 1) come with one 8-byte load; RGB / grey unrolled.
// A wave instruction covers a KM_PATCH_W x (64 / KM_PATCH_W) patch of the output rather than a 64 x 1 row: under rotation
// the taps of a 64 x 1 row are spread over up to 64 sin(angle) source rows (one cache line each), a squarer patch keeps
// them within ~(PW sin + PH cos) rows.  Stores stay KM_PATCH_W * 4-byte contiguous runs.
template <typename T, int CM, int NC, int ALIGN, bool FAST, int PH = 64 / KM_PATCH_W, bool STREAM = true>  // PH: tile rows between a thread's consecutive rows
__device__ __forceinline__ void km_warp_fwd_lean_rows(const KmWarpArgs<T>& a, const float (&m)[9], const float4* s_rv, uint32_t b, int j, int li_base,
                                                      int i_base) {
    const KmWarpGeom<float>& g = a.g;
    const int W = g.W, H = g.H;
    const int C = (NC > 0) ? NC : g.C;
    const size_t src_plane = (size_t)H * W, dst_plane = (size_t)g.h * g.w;
    const T* __restrict__ src_b = a.src + (size_t)b * C * src_plane;
    T* __restrict__ dst_b = a.dst + (size_t)b * C * dst_plane;
    const float Wm1 = (float)(W - 1), Hm1 = (float)(H - 1), hW = (float)W / 2, hH = (float)H / 2, Wm2 = (float)(W - 2), Hm2 = (float)(H - 2);
    const KmlHalf cu = kml_col_half<CM>(m, km_base_x<float, CM>(g, j));

    KmlTaps t[KM_ROWS];
    float xs[KM_ROWS], ys[KM_ROWS];
    bool inside = true;
#pragma unroll
    for (int r = 0; r < KM_ROWS; ++r) {
        const float4 rv4 = s_rv[li_base + r * PH];
        KmlHalf rv;
        rv.a = rv4.x; rv.b = rv4.y; rv.c = rv4.z;
        KmlPos p;
        kml_position<CM, FAST>(m, cu, rv, p);
        xs[r] = kml_unnormalize<ALIGN>(p.gx, Wm1, hW);
        ys[r] = kml_unnormalize<ALIGN>(p.gy, Hm1, hH);
        kml_taps(xs[r], ys[r], t[r]);
        inside = inside & kml_inside(t[r], Wm2, Hm2) & (i_base + r * PH < g.h);  // (no short circuit: straight-line code)
    }
    const uint32_t out0 = (uint32_t)i_base * (uint32_t)g.w + (uint32_t)j;
    if (NC > 0 && __all(inside)) {
        constexpr int NCC = NC > 0 ? NC : 1;
        const T* __restrict__ sp[NCC];
        T* __restrict__ dp[NCC];
#pragma unroll
        for (int c = 0; c < NCC; ++c) { sp[c] = src_b + c * src_plane; dp[c] = dst_b + c * dst_plane; }  // wave-uniform plane bases
        // The memory pipeline is bound by the BYTES the lanes request, whatever the instruction mix (profiles/r02_hbm_shapes.txt:
        // a 2 x 2 footprint per pixel - 16 bytes for 4 bytes of output - costs 0.39 ms at this size, 8 bytes 0.30, a plain copy
        // 0.29): a lane loads its own column of the footprint, (x0, y0) and (x0, y0 + 1), and takes the x0 + 1 column from
        // the next lane when that lane's footprint starts exactly one pixel to the right - which it does for most lanes of any
        // warp of scale ~1; the others (and the last lane of each output row of the patch) load it themselves.  Same values either way.
        // (Fetching the edge lanes' columns with ONE helper load - lane k loads item k - and handing them over through ds_bpermute
        // measured slower: 0.425 vs 0.378 ms.  Ablations on one box, wrong results, timing only: 0.39 ms as is; without the fallback
        // loads of the lanes that have no right neighbour 0.362; with one source row on top of that - 3 loads + 3 stores per 64 pixels, the
        // shape of a copy - 0.324; identity / pure translation matrices 0.355 - 0.366: the floor of a lane-per-pixel tile copy is ~0.32.
        // Rows per thread 2 / 3 / 4 (62 / 80 / 94 registers, 8 / 6 / 5 waves per SIMD): 0.381 / 0.389 / 0.389 - occupancy is not it.)
        float v[KM_ROWS][NCC][4];
        if constexpr (sizeof(T) == 2) {
            // 16-bit storage: (x0, x0 + 1) IS one 4-byte request - the cheapest class the memory pipeline has - so every lane loads its own
            // pair and there is neither a neighbour exchange nor a fallback load (config 3's rotated 224^2 bf16 images: 108 -> see DESIGN.md)
#pragma unroll
            for (int r = 0; r < KM_ROWS; ++r) {
                const uint32_t off = (uint32_t)__mul24((int)t[r].yf, W) + (uint32_t)(int)t[r].xf;
#pragma unroll
                for (int c = 0; c < NCC; ++c) {
                    km_ld2(km_at(sp[c], off), v[r][c][0], v[r][c][1]);
                    km_ld2(km_at(sp[c], off + (uint32_t)W), v[r][c][2], v[r][c][3]);
                }
            }
        } else {
            bool nb[KM_ROWS];
#pragma unroll
            for (int r = 0; r < KM_ROWS; ++r) {
                const uint32_t off = (uint32_t)__mul24((int)t[r].yf, W) + (uint32_t)(int)t[r].xf;
                nb[r] = (km_next64(off) == off + 1u);  // (lane 63 reads itself: false; lane 31's neighbour is on another row: false)
#pragma unroll
                for (int c = 0; c < NCC; ++c) {
                    v[r][c][0] = (float)km_ld(km_at(sp[c], off));
                    v[r][c][2] = (float)km_ld(km_at(sp[c], off + (uint32_t)W));
                    if (!nb[r]) {
                        v[r][c][1] = (float)km_ld(km_at(sp[c], off + 1u));
                        v[r][c][3] = (float)km_ld(km_at(sp[c], off + (uint32_t)W + 1u));
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < KM_ROWS; ++r)
#pragma unroll
                for (int c = 0; c < NCC; ++c) {
                    const float n0 = km_next64(v[r][c][0]), n2 = km_next64(v[r][c][2]);
                    if (nb[r]) { v[r][c][1] = n0; v[r][c][3] = n2; }
                }
        }
#pragma unroll
        for (int r = 0; r < KM_ROWS; ++r) {
            const float w00 = t[r].wx1 * t[r].wy1, w01 = t[r].wx0 * t[r].wy1, w10 = t[r].wx1 * t[r].wy0, w11 = t[r].wx0 * t[r].wy0;
            const uint32_t oo = out0 + (uint32_t)(r * PH) * (uint32_t)g.w;
#pragma unroll
            for (int c = 0; c < NCC; ++c) {
                const float acc = km_fma(v[r][c][3], w11, km_fma(v[r][c][2], w10, km_fma(v[r][c][1], w01, km_fma(v[r][c][0], w00, 0.0f))));
                km_st_c<STREAM>(km_at_mut(dp[c], oo), acc);  // (compile-time: a run-time policy test at each of the 12 stores cost 4 % of the kernel)
            }
        }
        return;
    }
    // some lane touches the border (or the tile hangs over the bottom edge): per row, taps predicated individually
#pragma unroll
    for (int r = 0; r < KM_ROWS; ++r) {
        const bool row_ok = i_base + r * PH < g.h;
        KmBilin<float> tr;
        km_bilinear_setup(xs[r], ys[r], W, H, tr);
        T* __restrict__ out_px = dst_b + (size_t)(row_ok ? i_base + r * PH : 0) * g.w + j;
        if (__all(tr.b00 && tr.b01 && tr.b10 && tr.b11)) {
            for (int c = 0; c < C; ++c) {
                const T* img = src_b + (size_t)c * src_plane;
                float v00, v01, v10, v11;
                km_ld2(img + tr.i00, v00, v01);
                km_ld2(img + tr.i10, v10, v11);
                const float acc = km_fma(v11, tr.w11, km_fma(v10, tr.w10, km_fma(v01, tr.w01, km_fma(v00, tr.w00, 0.0f))));
                if (row_ok) km_st(out_px + (size_t)c * dst_plane, acc);
            }
        } else {
            for (int c = 0; c < C; ++c) {
                const T* img = src_b + (size_t)c * src_plane;
                const float v00 = km_ld(img + tr.i00), v01 = km_ld(img + tr.i01);
                const float v10 = km_ld(img + tr.i10), v11 = km_ld(img + tr.i11);
                float acc = km_bilinear_masked(tr, (float)v00, (float)v01, (float)v10, (float)v11);  // (out-of-bounds taps: zeros that are still multiplied - ATen's CPU rule)
                if (row_ok) km_st(out_px + (size_t)c * dst_plane, acc);
            }
        }
    }
}

// A block takes KML_GROUPS vertically adjacent 64 x 16 groups of rows one after the other: what a block pays once - matrix
// and argument loads, the row table with its IEEE divisions, a barrier, ~2 us before the first pixel - is spread over
// 64 x 16 x KML_GROUPS pixels (measured: the kernel is bound by these serial latencies and the memory latency behind them,
// not by bytes or instructions).
#ifndef KML_GROUPS
#define KML_GROUPS 2   // measured on one box, 256x3x512^2: 1 -> 0.49 ms, 2 -> 0.41, 4 -> 0.42 (round-1 kernel there: 0.44)
#endif
#define KML_TILE_H (KM_TILE_H * KML_GROUPS)
#ifndef KML_BOUNDS
#define KML_BOUNDS __launch_bounds__(256)
#endif
template <typename T, int CM, int NC, int ALIGN>  // NC = 3 / 1: RGB / grey unrolled ; NC = 0: runtime channel loop
__global__ KML_BOUNDS void km_warp_fwd_lean_kernel(const KmWarpArgs<T> a) {
    static_assert(KML_TILE_H <= 64, "the row table is filled by one wave");
    const KmWarpGeom<float>& g = a.g;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks, a.reverse);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    constexpr int PW = KM_PATCH_W, PH = 64 / PW, WA = 64 / PW;  // patch, waves across the 64-wide tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = (int)tx * KM_TILE_W + (wave % WA) * PW + (lane % PW);
    const int li_base = (wave / WA) * (PH * KM_ROWS) + lane / PW;  // row inside a group of this thread's row r: li_base + r * PH
    const int i_base = (int)ty * KML_TILE_H + li_base;
    __shared__ float4 s_rv[KML_TILE_H];
    __shared__ int s_fast;
    if (a.apply && !a.apply[b]) {  // block-uniform: this sample is not transformed
        for (int gr = 0; gr < KML_GROUPS; ++gr) km_fwd_copy_rows<T>(a, b, j, i_base + gr * KM_TILE_H, PH, KM_ROWS);
        return;
    }

    float m[9];
    {
        const float* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = mp[k];
    }
    if (wave == 0) {  // row halves of the numerators + the per-row division guard, combined over the tile's rows
        bool ok = true;
        if (lane < KML_TILE_H) {
            const float v = km_base_y<float, CM>(g, (int)ty * KML_TILE_H + lane);
            const KmlHalf h = kml_row_half<CM>(m, v);
            s_rv[lane] = make_float4(h.a, h.b, h.c, 0.f);
            ok = kml_row_guard<CM>(g, m, v);
        }
        const bool all_ok = __all(ok);
        if (lane == 0) s_fast = all_ok ? 1 : 0;
    }
    __syncthreads();
    if (j >= g.w) return;
    const bool fast = __builtin_amdgcn_readfirstlane(s_fast) != 0;
    for (int gr = 0; gr < KML_GROUPS; ++gr) {
        const int ib = i_base + gr * KM_TILE_H;
        if ((int)ty * KML_TILE_H + gr * KM_TILE_H >= g.h) break;  // block-uniform: the group lies below the image
        constexpr int PH_ = 64 / KM_PATCH_W;
        if (a.stream_out) {  // (kernel-uniform) streaming stores for outputs of km_stream_stores' size, plain ones below
            if (fast) km_warp_fwd_lean_rows<T, CM, NC, ALIGN, true, PH_, true>(a, m, s_rv, b, j, li_base + gr * KM_TILE_H, ib);
            else km_warp_fwd_lean_rows<T, CM, NC, ALIGN, false, PH_, true>(a, m, s_rv, b, j, li_base + gr * KM_TILE_H, ib);
        } else {
            if (fast) km_warp_fwd_lean_rows<T, CM, NC, ALIGN, true, PH_, false>(a, m, s_rv, b, j, li_base + gr * KM_TILE_H, ib);
            else km_warp_fwd_lean_rows<T, CM, NC, ALIGN, false, PH_, false>(a, m, s_rv, b, j, li_base + gr * KM_TILE_H, ib);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Generic backward: scatter-add into grad_src with hardware fp32/fp64 atomics (what ATen's
// grid_sampler_2d_backward does) + per-image reduction of the matrix gradient.
// Matrix gradient, r = (u, v, 1):
//   perspective:  d/dm0k = ggx r_k / den ; d/dm1k = ggy r_k / den ; d/dm2k = -(ggx gx + ggy gy) r_k / den
//   affine:       d/dm0k = ggx r_k ; d/dm1k = ggy r_k
//   homography:   d/dH0k = ggx s r_k ; d/dH1k = ggy s r_k ; d/dH2k = -(ggx X + ggy Y) s^2 r_k (live only)
// Per-thread partial sums are fp32/fp64 (<= KM_ROWS*C terms), the wave/block reduction and the global
// accumulation are fp64 (the reference's own fp32 result is only ~1e-1 accurate).
template <typename T, int CM, int INTERP>
__global__ __launch_bounds__(256) void km_warp_bwd_kernel(const KmWarpArgs<T> a) {
    typedef typename KmTraits<T>::R R;
    const KmWarpGeom<R>& g = a.g;
    __shared__ double red[4][9];

    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;

    constexpr int PW = KM_PATCH_W, PH = 64 / PW, WA = 64 / PW;  // 32 x 2 output patch per wave instruction
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = (int)tx * KM_TILE_W + (wave % WA) * PW + (lane % PW);
    const int i_base = (int)ty * KM_TILE_H + (wave / WA) * (PH * KM_ROWS) + lane / PW;
    const bool want_gg = (CM == KM_COORD_GRID) && (a.ggrid != nullptr);  // gradient wrt the explicit grid
    const bool want_gm = ((a.gmat != nullptr) || want_gg) && (INTERP != KM_INTERP_NEAREST);  // needs d out / d (x, y)
    // the chain from (gix, giy) to the matrix runs for nearest too, with gix = giy = 0 (ATen's zero grid gradient): autograd multiplies those
    // zeros by the coordinates' partial derivatives, so a NaN / inf coordinate makes the matrix gradient NaN (tests/golden/nonfinite_coords.npz)
    const bool acc_gm = (CM != KM_COORD_GRID) && (a.gmat != nullptr);
    const bool want_gs = (a.gsrc != nullptr);

    R m[9];
    {
        const R* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = (CM == KM_COORD_GRID) ? (R)0 : mp[k];  // grid mode has no matrix
    }
    const int spad = (g.pad == KM_PAD_FILL) ? KM_PAD_ZEROS : g.pad;
    const size_t src_plane = (size_t)g.H * g.W, dst_plane = (size_t)g.h * g.w;
    const T* src_b = a.src + (size_t)b * g.C * src_plane;
    const T* gout_b = a.gout + (size_t)b * g.C * dst_plane;
    R* gsrc_b = want_gs ? a.gsrc + (size_t)b * g.C * src_plane : nullptr;

    R gm[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) gm[k] = 0;

    if (j < g.w) {
        const R u = km_base_x<R, CM>(g, j);
#pragma unroll
        for (int r = 0; r < KM_ROWS; ++r) {
            const int i = i_base + r * PH;
            if (i >= g.h) break;
            const R v = km_base_y<R, CM>(g, i);
            KmCoord<R> cd;
            km_gen_coord<R, CM>(m, u, v, cd);
            if (CM == KM_COORD_GRID) km_grid_coord(a.grid, g, b, i, j, cd);
            R mx, my, gdx = 1, gdy = 1;
            R x = km_unnormalize(cd.gx, g.W, g.align, mx);
            R y = km_unnormalize(cd.gy, g.H, g.align, my);
            const T* go_px = gout_b + (size_t)i * g.w + j;
            R gix = 0, giy = 0;

            if (INTERP == KM_INTERP_BILINEAR) {
                x = km_compute_coord(x, g.W, spad, g.align, gdx);
                y = km_compute_coord(y, g.H, spad, g.align, gdy);
                KmBilin<R> t;
                km_bilinear_setup(x, y, g.W, g.H, t);
                for (int c = 0; c < g.C; ++c) {
                    const R go = km_ld(go_px + (size_t)c * dst_plane);
                    if (want_gs) {
                        R* gi = gsrc_b + (size_t)c * src_plane;
                        if (t.b00) km_atomic_add(gi + t.i00, t.w00 * go);
                        if (t.b01) km_atomic_add(gi + t.i01, t.w01 * go);
                        if (t.b10) km_atomic_add(gi + t.i10, t.w10 * go);
                        if (t.b11) km_atomic_add(gi + t.i11, t.w11 * go);
                    }
                    if (want_gm) {
                        const T* img = src_b + (size_t)c * src_plane;
                        const R f = (g.pad == KM_PAD_FILL) ? a.fill[c] : (R)0;
                        km_bilinear_grid_terms<R>(t, (R)km_ld(img + t.i00) - f, (R)km_ld(img + t.i01) - f, (R)km_ld(img + t.i10) - f, (R)km_ld(img + t.i11) - f, go, gix, giy);
                    }
                }
                gix = gix * (mx * gdx);
                giy = giy * (my * gdy);
            } else if (INTERP == KM_INTERP_NEAREST) {
                x = km_compute_coord(x, g.W, spad, g.align, gdx);
                y = km_compute_coord(y, g.H, spad, g.align, gdy);
                const R xr = km_rint(x), yr = km_rint(y);
                const bool inb = (xr >= (R)0) && (xr <= (R)(g.W - 1)) && (yr >= (R)0) && (yr <= (R)(g.H - 1));
                if (want_gs && inb) {
                    const int idx = (int)yr * g.W + (int)xr;
                    for (int c = 0; c < g.C; ++c)
                        km_atomic_add(gsrc_b + (size_t)c * src_plane + idx, (R)km_ld(go_px + (size_t)c * dst_plane));
                }
            } else {
                const R xf = km_floor(x), yf = km_floor(y);
                R cx[4], cy[4], dx[4], dy[4];
                km_cubic_coeffs(x - xf, cx);
                km_cubic_coeffs(y - yf, cy);
                km_cubic_coeffs_grad(x - xf, dx);
                km_cubic_coeffs_grad(y - yf, dy);
                for (int c = 0; c < g.C; ++c) {
                    const R go = km_ld(go_px + (size_t)c * dst_plane);
                    const T* img = src_b + (size_t)c * src_plane;
                    R* gi = want_gs ? gsrc_b + (size_t)c * src_plane : nullptr;
                    const R f = (g.pad == KM_PAD_FILL) ? a.fill[c] : (R)0;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int idx = km_tap_index(xf - 1 + q, yf - 1 + rr, g.W, g.H, spad, g.align);
                            if (idx >= 0 && want_gs) km_atomic_add(gi + idx, go * cx[q] * cy[rr]);
                            if (want_gm) {
                                // (get_value_bounded: a tap outside the image is a zero that is still multiplied - NaN coefficients of a
                                // non-finite position reach the grid gradient; fill: taps are unpadded, idx >= 0 <=> in bounds)
                                const R s = idx >= 0 ? (R)km_ld(img + idx) - ((g.pad == KM_PAD_FILL) ? f : (R)0) : (R)0;
                                gix -= s * dx[q] * cy[rr] * go;
                                giy -= s * dy[rr] * cx[q] * go;
                            }
                        }
                }
                gix = gix * mx;
                giy = giy * my;
            }

            if (want_gg) {
                // d loss / d grid[b, i, j, :] (zero for nearest: ATen returns a zero grid gradient there)
                R* gp = a.ggrid + (((size_t)b * g.h + i) * g.w + j) * 2;
                gp[0] = (INTERP == KM_INTERP_NEAREST) ? (R)0 : gix;
                gp[1] = (INTERP == KM_INTERP_NEAREST) ? (R)0 : giy;
            }
            if (acc_gm) {
                if (CM == KM_COORD_PERSPECTIVE) {
                    const R inv = (R)1 / cd.den;
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
        }
    }

    if (acc_gm) {  // block-uniform
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double s = km_wave_sum((double)gm[k]);
            if (lane == 0) red[wave][k] = s;
        }
        __syncthreads();
        if (threadIdx.x < 9) {
            const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            km_atomic_add(a.gmat + (size_t)(g.B_M == 1 ? 0 : b) * 9 + threadIdx.x, s);
        }
    }
}

// ------------------------------------------------------------------------------------------------
static bool km_fwd_generic_forced() {
    return km_config().warp_fwd_algo == 1;  // KM_WARP_FWD_ALGO=generic: A/B timing
}

// ------------------------------------------------------------------------------------------------
// Box forward (bilinear + zeros, RGB / grey, fp32 compute): the source box of an output tile through LDS, with the block's serial
// latencies overlapped.
//
// What the gather forward costs is the bytes its lanes REQUEST (8 - 12 bytes per channel and pixel for 4 bytes of
// output; a lane-per-pixel tile copy has a floor of ~0.32 ms at config 2, a 16-byte tile copy 0.285).  Here every source byte of
// the tile's box is requested once, with 16-byte row loads, and every tap is a ds_read2_b32.  The first LDS forward
// (round 2's km_warp_fwd_lds_kernel, removed: 32 x 32 tiles, 112 registers, box -> barrier -> fill -> barrier -> positions -> sample, one after
// the other) lost to the gathers (0.42 - 0.44 against 0.39 ms); this one is built around what that one lacked:
//   * the fill's loads are issued FIRST and the thread's positions (~70 instructions each, independent of the loads) are computed
//     while they fly; the only wait of the block is the one before the LDS stores;
//   * the block-uniform matrix sits in scalar registers before the fill is requested (left in flight, the wait for it lands behind
//     the fill's requests - which sit under lane predicates the compiler cannot count - and becomes a wait for the whole fill);
//   * no persistent loop and no double buffer inside a block: 4 - 6 blocks per CU, whose phases interleave, are the overlap;
//   * lane = output column (ds_read2_b32 of neighbouring lanes hit neighbouring banks, stores are whole 128 / 256-byte runs).
// Same positions, same fma chain (nw, ne, sw, se from 0), zeros staged outside the image (fma(0, w, acc) == acc for the finite
// weights of a finite position): bit-identical to the other forwards and to the oracle.
// A wave with a footprint outside its box (box estimate off, NaN position) gathers its rows one at a time: the result never depends on
// the box estimate.
// A block owns a 64 x 32 region of the output and tries, in this order (every decision block-uniform):
//   1. KmbWide:   the region as ONE tile, box up to 80 x 40 source pixels - axis-aligned maps up to ~7 degrees, the flagship homographies;
//   2. KmbSquare: its two 32 x 32 halves one after the other, box up to 52 x 50 each - ANY rotation at scale ~1 (measured at 5 / 20 / 45
//                 degrees: 0.37 - 0.39 ms where the gather kernel takes 0.47 / 0.71 / 0.92);
//   3. the gather rows of km_warp_fwd_lean_kernel: for a half whose box still does not fit, and - before the squares are tried - for a region
//      whose output rows stay nearly horizontal in the source (its wide box was too large because the map minifies: gathers are fine there).
#ifndef KMB_WIDE_PITCH
#define KMB_WIDE_PITCH 80
#define KMB_WIDE_ROWS 40
#endif
struct KmbWide   { static constexpr int TW = 64, TH = 32, PITCH = KMB_WIDE_PITCH, ROWS = KMB_WIDE_ROWS; };
struct KmbSquare { static constexpr int TW = 32, TH = 32, PITCH = 52, ROWS = 50; };
#define KMB_LDS_FLOATS(NC) ((KmbWide::ROWS * KmbWide::PITCH > KmbSquare::ROWS * KmbSquare::PITCH ? KmbWide::ROWS * KmbWide::PITCH : KmbSquare::ROWS * KmbSquare::PITCH) * (NC))
#ifndef KMB_WAVES_PER_EU
#define KMB_WAVES_PER_EU 4   // what the LDS of a block allows (38 KB: 4 blocks of 4 waves per CU)
#endif
#ifndef KMB_TRY_WIDE
#define KMB_TRY_WIDE 1       // 0: squares only (A/B)
#endif
#ifndef KMB_FILL_DMA
#define KMB_FILL_DMA 1       // fp32 storage: the box by LDS-DMA (global_load_lds_dwordx4) instead of through registers
#endif
#ifndef KMB_ST16
#define KMB_ST16 0           // 1: fp32 storage, a quad of lanes transposes its 4 x 4 results and stores 16 bytes per lane (0: one dword per lane and row).
                             // MEASURED SLOWER (round 6, profiles/r06/run1_*: same box, bit-identical): flagship 312.0 / 313.1 us against 288.6 / 287.2,
                             // 5 degrees 338 against 310, 20 degrees 422 against 386 - a quarter of the store instructions, 8 % more time: see DESIGN.md 4.6
#endif
#ifndef KMB_TILT_ROWS
#define KMB_TILT_ROWS 4      // a region whose output rows span at most this many source rows takes the gather rows when its wide box does not fit
#endif

// the rows of a thread by gathers, one row at a time (plain IEEE divisions: any operands; every tap predicated): what a wave with a
// footprint outside its box falls back to.  Deliberately small - the registers of the kernel are the maximum over its paths.
template <typename T, int CM, int NC, int ALIGN, int RPT>
__device__ __forceinline__ void kmb_gather_rows(const KmWarpArgs<T>& a, const float (&m)[9], const float4* s_rv, uint32_t b, int j, int li_base, int i_base) {
    const KmWarpGeom<float>& g = a.g;
    const int W = g.W, H = g.H;
    const size_t src_plane = (size_t)H * W, dst_plane = (size_t)g.h * g.w;
    const T* __restrict__ src_b = a.src + (size_t)b * NC * src_plane;
    T* __restrict__ dst_b = a.dst + (size_t)b * NC * dst_plane;
    const float Wm1 = (float)(W - 1), Hm1 = (float)(H - 1), hW = (float)W / 2, hH = (float)H / 2;
    const KmlHalf cu = kml_col_half<CM>(m, km_base_x<float, CM>(g, j));
#pragma unroll 1
    for (int r = 0; r < RPT; ++r) {
        if (i_base + r >= g.h) break;
        const float4 rv4 = s_rv[li_base + r];
        KmlHalf rv;
        rv.a = rv4.x; rv.b = rv4.y; rv.c = rv4.z;
        KmlPos p;
        kml_position<CM, false>(m, cu, rv, p);
        KmBilin<float> tr;
        km_bilinear_setup(kml_unnormalize<ALIGN>(p.gx, Wm1, hW), kml_unnormalize<ALIGN>(p.gy, Hm1, hH), W, H, tr);
        T* __restrict__ out_px = dst_b + (size_t)(i_base + r) * g.w + j;
#pragma unroll 1
        for (int c = 0; c < NC; ++c) {
            const T* img = src_b + (size_t)c * src_plane;
            const float v00 = km_ld(img + tr.i00), v01 = km_ld(img + tr.i01);
            const float v10 = km_ld(img + tr.i10), v11 = km_ld(img + tr.i11);
            float acc = km_bilinear_masked(tr, (float)v00, (float)v01, (float)v10, (float)v11);  // (out-of-bounds taps: zeros that are still multiplied - ATen's CPU rule)
            km_st(out_px + (size_t)c * dst_plane, acc);
        }
    }
}

// One output tile of shape SH with its upper left corner at (j0, i0), by the whole block.  Returns false - block-uniform, nothing written,
// one barrier passed - when the tile's box does not fit SH's capacity.  The caller has a barrier between two calls (s_rv, s_info, s_src).
// fast_out: the division operands of every row of the tile are in the safe range (what a gather fallback of the caller wants to know).
template <typename T, int CM, int NC, int ALIGN, bool STREAM, typename SH>
__device__ __forceinline__ void kmb_tile_body(const KmWarpArgs<T>& a, const float (&m)[9], uint32_t b, int j0, int i0, const KmfBox& bx, const float4* s_rv, float* s_src);

template <typename T, int CM, int NC, int ALIGN, bool STREAM, typename SH>
__device__ __forceinline__ bool kmb_tile(const KmWarpArgs<T>& a, const float (&m)[9], uint32_t b, int j0, int i0, float4* s_rv, int* s_info, float* s_src,
                                         bool& fast_out) {
    kmf_tile_setup<CM, ALIGN, 0, SH::TW, SH::TH, SH::PITCH, SH::ROWS>(a.g, m, j0, i0, s_rv, s_info, false);
    __syncthreads();
    const KmfBox bx = kmf_read_box(s_info);
    fast_out = bx.fast;
    if (!bx.staged) return false;  // block-uniform
    kmb_tile_body<T, CM, NC, ALIGN, STREAM, SH>(a, m, b, j0, i0, bx, s_rv, s_src);
    return true;
}

// the tile once its box is known to fit (bx.staged): fill, positions, sample.  One barrier inside.
template <typename T, int CM, int NC, int ALIGN, bool STREAM, typename SH>
__device__ __forceinline__ void kmb_tile_body(const KmWarpArgs<T>& a, const float (&m)[9], uint32_t b, int j0, int i0, const KmfBox& bx, const float4* s_rv, float* s_src) {
    constexpr int TW = SH::TW, TH = SH::TH, PITCH = SH::PITCH, ROWS = SH::ROWS;
    constexpr int RSLOTS = 256 / TW;       // threads per output column
    constexpr int RPT = TH / RSLOTS;       // output rows per thread (consecutive)
    constexpr int NCHK = PITCH / 4;        // 16-byte chunks per staged row of one channel
    constexpr int RCPP = 256 / NCHK;       // (row, channel) pairs filled per pass of the block
    static_assert(TW == 64 || TW == 32, "lane = output column");
    static_assert(TH <= 64 && PITCH % 4 == 0, "one wave fills the row table; whole chunks");
    const KmWarpGeom<float>& g = a.g;
    const int tid = km_tid_pinned();
    const int j = j0 + (tid % TW);
    const int li_base = (tid / TW) * RPT;             // this thread's rows: li_base + r
    const int i_base = i0 + li_base;
    const int W = g.W, H = g.H;
    const size_t src_plane = (size_t)H * W, dst_plane = (size_t)g.h * g.w;
    const T* __restrict__ src_b = a.src + (size_t)b * NC * src_plane;
    T* __restrict__ dst_b = a.dst + (size_t)b * NC * dst_plane;

    // ---- the fill's requests: thread = (chunk of 4 columns, (row, channel) pair of the pass) ----
    constexpr int NPASS = (ROWS * NC + RCPP - 1) / RCPP;
    const int ck = tid % NCHK, rq = tid / NCHK;
    const int xg = bx.xs + 4 * ck;
    const bool filler = (rq < RCPP) && (ck < bx.nch);
    const bool col_in = (xg >= 0) && (xg + 3 < W);  // W % 4 == 0 and xs % 4 == 0: a chunk is inside or outside as a whole
    const int nrc = bx.nrows * NC;
#if KMB_FILL_DMA
    // fp32 storage: every chunk straight into LDS (KM_GLDS16: the wave's 64 pieces land at consecutive 16-byte cells - a pass of the block is
    // RCPP * NCHK consecutive chunks, thread t's at chunk t); no register holds the box on its way (40 of this kernel's 127), no ds_write pass.
    // Chunks outside the image are zeros, written the ordinary way.  The requests have landed after KM_VMCNT0() + the barrier below.
    constexpr bool DMA = sizeof(T) == 4;
#else
    constexpr bool DMA = false;
#endif
    // (UNCONDITIONAL loads - a chunk outside the image or beyond the box reads the first chunk of the image and is replaced by zeros on its
    // way to LDS: with `v = 0; if (inside) load` the compiler keeps the whole array as one register tuple and copies - or spills - all of
    // it at every conditional definition)
    float v[DMA ? 1 : NPASS][4];
    uint32_t inmask = 0u;
    if constexpr (DMA) {
        const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int rc = ps * RCPP + rq, r = rc / NC, c = rc - r * NC, y = bx.ys + r;
            const bool live = filler && (rc < nrc);
            const bool inb = live && col_in && (y >= 0) && (y < H);
            if (inb) KM_GLDS16(km_at(reinterpret_cast<const float*>(src_b) + c * src_plane, (uint32_t)y * (uint32_t)W + (uint32_t)xg), s_src + ps * (RCPP * PITCH) + 4 * wbase);
            else if (live) *reinterpret_cast<float4*>(s_src + rc * PITCH + 4 * ck) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int rc = ps * RCPP + rq, r = rc / NC, c = rc - r * NC, y = bx.ys + r;
            const bool inb = filler && col_in && (rc < nrc) && (y >= 0) && (y < H);
            inmask |= inb ? (1u << ps) : 0u;
            km_ld4(km_at(src_b + (inb ? c : 0) * src_plane, inb ? (uint32_t)y * (uint32_t)W + (uint32_t)xg : 0u), v[ps]);
        }
    }

    // ---- this thread's positions, while the requests fly ----
    const float Wm1 = (float)(W - 1), Hm1 = (float)(H - 1), hW = (float)W / 2, hH = (float)H / 2;
    const KmlHalf cu = kml_col_half<CM>(m, km_base_x<float, CM>(g, j));
    float xs[RPT], ys[RPT];  // (positions only: the footprints are formed again at the sampling - 6 registers per row would not fit beside the fill)
    bool inbox = true;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const float4 rv4 = s_rv[li_base + r];
        KmlHalf rv;
        rv.a = rv4.x; rv.b = rv4.y; rv.c = rv4.z;
        KmlPos p;
        if (bx.fast) kml_position<CM, true>(m, cu, rv, p);
        else kml_position<CM, false>(m, cu, rv, p);
        xs[r] = kml_unnormalize<ALIGN>(p.gx, Wm1, hW);
        ys[r] = kml_unnormalize<ALIGN>(p.gy, Hm1, hH);
        KmlTaps t;
        kml_taps(xs[r], ys[r], t);
        // (rows of the tile below the image bottom and columns right of it are not stored: no constraint)
        inbox = inbox & (kmf_in_box(t, bx) | (i_base + r >= g.h) | (j >= g.w));
    }

    // ---- requests -> LDS ----
    if constexpr (DMA) {
        KM_VMCNT0();
    } else {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int rc = ps * RCPP + rq;
            if (filler && rc < nrc) {
                float* q = s_src + rc * PITCH + 4 * ck;
                KM_CHECK_ALIGNED(q, 16);
                const bool inb = (inmask >> ps) & 1u;
                *reinterpret_cast<float4*>(q) = make_float4(inb ? v[ps][0] : 0.f, inb ? v[ps][1] : 0.f, inb ? v[ps][2] : 0.f, inb ? v[ps][3] : 0.f);
            }
        }
    }
    __syncthreads();
    const bool all_in = __all(inbox);  // (by every lane of the wave: columns right of the image count as inside)
#if KMB_ST16
    if constexpr (sizeof(T) == 4) {
        // 16-BYTE STORES (round 6).  The taps keep lane = output column (neighbouring lanes read neighbouring LDS banks); what changes is the
        // way out: the thread's rows are taken four at a time, each quad of lanes transposes its 4 rows x 4 columns per channel in registers
        // (km_quad_transpose4: quad-permute moves and selects - VALU this kernel has spare, profiles/r05_final_pmc_units.json: VALU 19 % busy,
        // the texture-address unit 71 %, two thirds of its instructions these stores) and lane q of the quad writes row q's four pixels with
        // ONE global_store_dwordx4: a quarter of the store instructions for the same bytes, the same values.
        // Output rows must be 16-byte aligned runs of whole quads (block-uniform test); every lane of the wave walks the groups - lanes right
        // of the image or below it compute nothing and store nothing, but take part in the transposes of their quad.
        const bool vec16 = ((g.w & 3) == 0) && ((j0 & 3) == 0) && (((uintptr_t)a.dst & 15) == 0);
        if (vec16 && all_in) {  // wave-uniform
            static_assert(RPT % 4 == 0, "rows of a thread in groups of four");
            const int q4 = tid & 3;
            const bool col_ok = j < g.w;
            const uint32_t outq = (uint32_t)i_base * (uint32_t)g.w + (uint32_t)(j & ~3);
            float* __restrict__ dp[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) dp[c] = reinterpret_cast<float*>(dst_b) + c * dst_plane;
#pragma unroll
            for (int r0 = 0; r0 < RPT; r0 += 4) {
                const bool grp_ok = col_ok && (i_base + r0 < g.h);  // (the tile hangs over the bottom / right edge of the image)
                if (!__any(grp_ok)) break;                          // wave-uniform (the groups that follow lie lower still)
                float acc[NC][4];
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) acc[c][rr] = 0.f;
                if (grp_ok) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        // (a row below the image samples the group's first row again: its own position need not lie in the box; never stored)
                        const bool rv = i_base + r0 + rr < g.h;
                        KmlTaps t;
                        kml_taps(rv ? xs[r0 + rr] : xs[r0], rv ? ys[r0 + rr] : ys[r0], t);
                        const int xi = KM_F2I(t.xf) - bx.xs, yi = KM_F2I(t.yf) - bx.ys;
                        const float* q0 = s_src + __mul24(yi, NC * PITCH) + xi;
                        const float* q1 = q0 + NC * PITCH;
                        const float w00 = t.wx1 * t.wy1, w01 = t.wx0 * t.wy1, w10 = t.wx1 * t.wy0, w11 = t.wx0 * t.wy0;
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const float v00 = q0[c * PITCH], v01 = q0[c * PITCH + 1], v10 = q1[c * PITCH], v11 = q1[c * PITCH + 1];
                            acc[c][rr] = km_fma(v11, w11, km_fma(v10, w10, km_fma(v01, w01, km_fma(v00, w00, 0.0f))));
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) km_quad_transpose4(acc[c], q4);
                if (grp_ok && (i_base + r0 + q4 < g.h)) {
                    const uint32_t oo = outq + (uint32_t)(r0 + q4) * (uint32_t)g.w;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        float* p = km_at_mut(dp[c], oo);
                        KM_CHECK_ALIGNED(p, 16);
                        km_st4_c<STREAM && !KM_FWD_PLAIN_ST>(p, acc[c]);
                    }
                }
            }
            return;
        }
    }
#endif
    if (j < g.w) {
        if (!all_in) {  // (never seen for boxes that fit; keeps the result independent of the box estimate)
            kmb_gather_rows<T, CM, NC, ALIGN, RPT>(a, m, s_rv, b, j, li_base, i_base);
        } else {
            // ---- sample ----
            T* __restrict__ dp[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) dp[c] = dst_b + c * dst_plane;
            const uint32_t out0 = (uint32_t)i_base * (uint32_t)g.w + (uint32_t)j;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                if (i_base + r >= g.h) break;  // (the tile hangs over the bottom edge; the rows that follow do too)
                KmlTaps t;
                kml_taps(xs[r], ys[r], t);
                const int xi = KM_F2I(t.xf) - bx.xs, yi = KM_F2I(t.yf) - bx.ys;
                const float* q0 = s_src + __mul24(yi, NC * PITCH) + xi;
                const float* q1 = q0 + NC * PITCH;
                const float w00 = t.wx1 * t.wy1, w01 = t.wx0 * t.wy1, w10 = t.wx1 * t.wy0, w11 = t.wx0 * t.wy0;
                const uint32_t oo = out0 + (uint32_t)r * (uint32_t)g.w;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float v00 = q0[c * PITCH], v01 = q0[c * PITCH + 1], v10 = q1[c * PITCH], v11 = q1[c * PITCH + 1];
                    const float acc = km_fma(v11, w11, km_fma(v10, w10, km_fma(v01, w01, km_fma(v00, w00, 0.0f))));
                    km_st_c<STREAM>(km_at_mut(dp[c], oo), acc);
                }
            }
        }
    }
}

#ifndef KMB_PERSISTENT
#define KMB_PERSISTENT 0     // 1: the launch is KMB_WAVES_PER_EU workgroups per CU, each walking its share of the regions (measured: slower, see below)
#endif
// one 64 x 32 region of the output (logical block `bid` of a.nblocks), by the whole workgroup
template <typename T, int CM, int NC, int ALIGN, bool STREAM>
__device__ __forceinline__ void kmb_region(const KmWarpArgs<T>& a, uint32_t bid, float4* s_rv, int* s_info, float* s_src) {
    const KmWarpGeom<float>& g = a.g;
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    const int tid = km_tid_pinned();
    const int J0 = (int)tx * KmbWide::TW, I0 = (int)ty * KmbWide::TH;
    if (a.apply && !a.apply[b]) {  // block-uniform: this sample is not transformed
        km_fwd_copy_rows<T>(a, b, J0 + (tid % KmbWide::TW), I0 + (tid / KmbWide::TW) * (KmbWide::TH / 4), 1, KmbWide::TH / 4);
        return;
    }
    float m[9];
    {
        // (block-uniform, into scalar registers HERE: left in flight, the wait for them lands behind the fill's requests - which sit
        // under lane predicates the compiler cannot count - and becomes a wait for the whole fill in front of the positions)
        const float* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mp[k])));
    }
    // ---- 1. the region as one wide tile ----
    kmf_tile_setup<CM, ALIGN, 0, KmbWide::TW, KmbWide::TH, KmbWide::PITCH, KmbWide::ROWS>(g, m, J0, I0, s_rv, s_info, false);
    __syncthreads();
    const KmfBox bxw = kmf_read_box(s_info);
    const int tilt = __builtin_amdgcn_readfirstlane(s_info[6]);  // source rows spanned by one output row of the region
    if (KMB_TRY_WIDE && bxw.staged) {
        kmb_tile_body<T, CM, NC, ALIGN, STREAM, KmbWide>(a, m, b, J0, I0, bxw, s_rv, s_src);
        return;
    }
    // ---- 3 before 2. output rows that stay (nearly) horizontal in the source - the box is too large because the map MINIFIES, not because
    //      it rotates: the gather rows do not suffer there (scale 0.8 / 0.5: 0.37 / 0.48 ms against 0.43 / 0.61 through the squares), in
    //      the very shape of km_warp_fwd_lean_kernel (32 x 2 patches per wave instruction, two groups of 64 x 16) ----
    if (tilt <= KMB_TILT_ROWS) {
        constexpr int PW = KM_PATCH_W, PH = 64 / PW, WA = 64 / PW;
        const int lane = tid & 63, wave = tid >> 6;
        const int j = J0 + (wave % WA) * PW + (lane % PW);
        const int li_base = (wave / WA) * (PH * KM_ROWS) + lane / PW;
        if (j >= g.w) return;  // (the kernel ends in this branch: lanes right of the image leave, as in km_warp_fwd_lean_kernel)
        {
#pragma unroll 1
            for (int gr = 0; gr < KmbWide::TH / KM_TILE_H; ++gr) {
                if (I0 + gr * KM_TILE_H >= g.h) break;
                if (bxw.fast) km_warp_fwd_lean_rows<T, CM, NC, ALIGN, true, PH, STREAM>(a, m, s_rv, b, j, li_base + gr * KM_TILE_H, I0 + li_base + gr * KM_TILE_H);
                else km_warp_fwd_lean_rows<T, CM, NC, ALIGN, false, PH, STREAM>(a, m, s_rv, b, j, li_base + gr * KM_TILE_H, I0 + li_base + gr * KM_TILE_H);
            }
        }
        return;
    }
    // ---- 2. the two square halves, one after the other ----
    bool fast = false;
    for (int half = 0; half < 2; ++half) {
        const int j0 = J0 + half * KmbSquare::TW;
        if (j0 >= g.w) break;  // block-uniform
        __syncthreads();  // the previous attempt's readers are done with s_info, s_rv and s_src
        if (kmb_tile<T, CM, NC, ALIGN, STREAM, KmbSquare>(a, m, b, j0, I0, s_rv, s_info, s_src, fast)) continue;
        // (every lane takes part - the rows exchange registers between neighbouring lanes and vote as a wave, and this block still has a
        // barrier ahead: a lane right of the image computes the image's last column again and stores the same values to the same places)
        const int j = min(j0 + (tid % KmbSquare::TW), g.w - 1), li_base = (tid / KmbSquare::TW) * KM_ROWS;
        if (fast) km_warp_fwd_lean_rows<T, CM, NC, ALIGN, true, 1, STREAM>(a, m, s_rv, b, j, li_base, I0 + li_base);
        else km_warp_fwd_lean_rows<T, CM, NC, ALIGN, false, 1, STREAM>(a, m, s_rv, b, j, li_base, I0 + li_base);
    }
}


// One workgroup per region.  (The counters say a workgroup lives 8.7 us and the CUs hold 11.1 of their 16 waves on average
// (profiles/r05_pmc_units.json); KMB_PERSISTENT = 1 - KMB_WAVES_PER_EU workgroups per CU that WALK the regions, no dispatch between two of
// them - was nevertheless 8 % SLOWER on the flagship maps and 25 - 45 % slower under rotation / minification, where regions differ in cost
// and a fixed share per workgroup does not balance: profiles/r05/fwd_box_variants.txt.  The loop stays for that switch: the default launch
// has one region per workgroup and runs it once.)
template <typename T, int CM, int NC, int ALIGN, bool STREAM>
__global__ __launch_bounds__(256, KMB_WAVES_PER_EU) void km_warp_fwd_box_kernel(const KmWarpArgs<T> a) {
    static_assert(KmbWide::TH == KmbSquare::TH && KmbWide::TW == 2 * KmbSquare::TW, "a wide tile is two square ones side by side");
    static_assert(KmbSquare::TH / (256 / KmbSquare::TW) == KM_ROWS, "the gather rows walk KM_ROWS rows of a square half per thread");
    __shared__ float4 s_rv[KmbWide::TH];
    __shared__ int s_info[8];
    __shared__ __attribute__((aligned(16))) float s_src[KMB_LDS_FLOATS(NC)];  // [row][channel][x]
    // (inside the loop every helper takes its thread index through km_tid_pinned(): what depends on the thread index alone would otherwise be
    // hoisted out of the loop and stay live across the whole body - 280 bytes of spills in a kernel that sits at its register limit)
#if KMB_PERSISTENT
    for (uint32_t lb = blockIdx.x; lb < a.nblocks; lb += gridDim.x) {
        if (lb != blockIdx.x) __syncthreads();  // the previous region's readers are done with s_rv, s_info and s_src
        kmb_region<T, CM, NC, ALIGN, STREAM>(a, km_xcd_remap(lb, a.nblocks, a.reverse), s_rv, s_info, s_src);
    }
#else
    kmb_region<T, CM, NC, ALIGN, STREAM>(a, km_xcd_remap(blockIdx.x, a.nblocks, a.reverse), s_rv, s_info, s_src);
#endif
}

// LDS-staged bicubic forward (km_warp_cubic.hip): launches and returns 1 when it takes the case, 0 otherwise
int km_warp_fwd_cubic_try_any(int dtype, int coord_mode, const void* args, hipStream_t s);

// the specialised forward: bilinear + zeros, fp32 compute, 32-bit byte offsets inside a plane, 24-bit row / column counts
template <typename T, int CM>
static bool km_fwd_lean_ok(const KmWarpArgs<T>& a) {
    if (CM == KM_COORD_GRID || sizeof(typename KmTraits<T>::R) != sizeof(float)) return false;
    if (a.g.pad != KM_PAD_ZEROS || a.g.W < 2 || km_fwd_generic_forced()) return false;
    return (uint64_t)a.g.H * a.g.W * 4 < (1ull << 32) && (uint64_t)a.g.h * a.g.w * 4 < (1ull << 32) && a.g.W < (1 << 23) && a.g.H < (1 << 23);
}
template <typename T, int CM, int NC>
static void km_warp_fwd_lean_launch_nc(const KmWarpArgs<T>& a0, hipStream_t s) {
    KmWarpArgs<T> a = a0;
    a.tiles_y = (uint32_t)((a.g.h + KML_TILE_H - 1) / KML_TILE_H);
    a.nblocks = a.tiles_x * a.tiles_y * (uint32_t)a.g.B;  // (<= the 64 x 16 grid the caller checked)
    a.reverse = km_traversal_next(s);
    a.stream_out = km_stream_stores((uint64_t)a.g.B * a.g.C * a.g.h * a.g.w * sizeof(T));
    if (a.g.align)
        hipLaunchKernelGGL((km_warp_fwd_lean_kernel<T, CM, NC, 1>), dim3(a.nblocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((km_warp_fwd_lean_kernel<T, CM, NC, 0>), dim3(a.nblocks), dim3(256), 0, s, a);
}
template <typename T, int CM, int NC>
static void km_warp_fwd_box_launch_nc(const KmWarpArgs<T>& a, hipStream_t s) {
    KmWarpArgs<T> b = a;
    b.tiles_x = (uint32_t)((a.g.w + KmbWide::TW - 1) / KmbWide::TW);
    b.tiles_y = (uint32_t)((a.g.h + KmbWide::TH - 1) / KmbWide::TH);
    b.nblocks = b.tiles_x * b.tiles_y * (uint32_t)a.g.B;
    b.reverse = km_traversal_next(s);
    b.stream_out = km_stream_stores((uint64_t)a.g.B * a.g.C * a.g.h * a.g.w * sizeof(T));
    uint32_t grid = b.nblocks;
    if (KMB_PERSISTENT) {
        // (a multiple of 8: a launch block and its logical blocks then share an XCD)
        const uint32_t resident = (uint32_t)(km_device_cus() > 0 ? km_device_cus() : 1) * KMB_WAVES_PER_EU;
        if (grid > resident) grid = resident;
    }
    if (a.g.align) {
        if (b.stream_out) hipLaunchKernelGGL((km_warp_fwd_box_kernel<T, CM, NC, 1, true>), dim3(grid), dim3(256), 0, s, b);
        else hipLaunchKernelGGL((km_warp_fwd_box_kernel<T, CM, NC, 1, false>), dim3(grid), dim3(256), 0, s, b);
    } else {
        if (b.stream_out) hipLaunchKernelGGL((km_warp_fwd_box_kernel<T, CM, NC, 0, true>), dim3(grid), dim3(256), 0, s, b);
        else hipLaunchKernelGGL((km_warp_fwd_box_kernel<T, CM, NC, 0, false>), dim3(grid), dim3(256), 0, s, b);
    }
}
template <typename T>
static uint64_t km_warp_fwd_box_blocks(const KmWarpArgs<T>& a) {
    return (uint64_t)((a.g.w + KmbWide::TW - 1) / KmbWide::TW) * (uint64_t)((a.g.h + KmbWide::TH - 1) / KmbWide::TH) * (uint64_t)a.g.B;
}
#ifndef KM_FWD_BOX_DEFAULT
#define KM_FWD_BOX_DEFAULT 1      // fp32 storage: the box forward is what the hot configuration runs (KM_WARP_FWD_ALGO=rows: the gather kernel)
#endif
#ifndef KM_FWD_BOX_DEFAULT_16
#define KM_FWD_BOX_DEFAULT_16 1   // 16-bit storage (config 3, 224^2 bf16 under +-15 degrees: 97 -> 73 us through the square tiles)
#endif
template <typename T, int CM>
static void km_warp_fwd_lean_launch(const KmWarpArgs<T>& a, hipStream_t s) {
    if constexpr (CM != KM_COORD_GRID && sizeof(typename KmTraits<T>::R) == sizeof(float)) {  // (never instantiated otherwise)
        {
            const uint64_t nbb = km_warp_fwd_box_blocks(a);
            const int algo = km_config().warp_fwd_algo;  // 0 default, 3 "box", 4 "rows"
            // (an output with fewer pixels than the source is, as a rule, a map that MINIFIES - warp to a smaller dsize: the boxes would not fit
            // and the kernel would end in its gather rows after paying for the attempt, 7 - 10 % behind the gather kernel; that one is launched then)
            const bool shrinks = (uint64_t)a.g.H * (uint64_t)a.g.W * 10u > (uint64_t)a.g.h * (uint64_t)a.g.w * 13u;
            const bool want_box = algo == 3 || ((sizeof(T) == 2 ? KM_FWD_BOX_DEFAULT_16 : KM_FWD_BOX_DEFAULT) && algo == 0 && !shrinks);
            if (want_box && (a.g.C == 3 || a.g.C == 1) && (a.g.W & 3) == 0 && ((uintptr_t)a.src % (4 * sizeof(T))) == 0 && nbb < (1ull << 31)) {
                if (a.g.C == 3) km_warp_fwd_box_launch_nc<T, CM, 3>(a, s);
                else km_warp_fwd_box_launch_nc<T, CM, 1>(a, s);
                return;
            }
        }
        if (a.g.C == 3) km_warp_fwd_lean_launch_nc<T, CM, 3>(a, s);
        else if (a.g.C == 1) km_warp_fwd_lean_launch_nc<T, CM, 1>(a, s);
        else km_warp_fwd_lean_launch_nc<T, CM, 0>(a, s);
    }
}

template <typename T, int CM, int INTERP>
static int km_warp_launch(bool bwd, const KmWarpArgs<T>& a, hipStream_t s) {
    if (bwd)
        hipLaunchKernelGGL((km_warp_bwd_kernel<T, CM, INTERP>), dim3(a.nblocks), dim3(256), 0, s, a);
    else if (INTERP == KM_INTERP_BILINEAR && km_fwd_lean_ok<T, CM>(a)) {
        km_warp_fwd_lean_launch<T, CM>(a, s);
    } else if (INTERP == KM_INTERP_BILINEAR && a.g.pad == KM_PAD_ZEROS && a.g.W >= 2 && !km_fwd_generic_forced()) {
        if (a.g.C == 3)
            hipLaunchKernelGGL((km_warp_fwd_bz_kernel<T, CM, 3>), dim3(a.nblocks), dim3(256), 0, s, a);
        else if (a.g.C == 1)
            hipLaunchKernelGGL((km_warp_fwd_bz_kernel<T, CM, 1>), dim3(a.nblocks), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((km_warp_fwd_bz_kernel<T, CM, 0>), dim3(a.nblocks), dim3(256), 0, s, a);
    } else if (INTERP == KM_INTERP_BICUBIC && !km_fwd_generic_forced() && km_warp_fwd_cubic_try_any(KmTraits<T>::code, CM, &a, s)) {
    } else
        hipLaunchKernelGGL((km_warp_fwd_kernel<T, CM, INTERP>), dim3(a.nblocks), dim3(256), 0, s, a);
    return km_check_launch(bwd ? "km_warp2d_bwd" : "km_warp2d_fwd");
}

template <typename T, int CM>
static int km_warp_dispatch_interp(bool bwd, const KmWarpArgs<T>& a, hipStream_t s) {
    switch (a.g.interp) {
        case KM_INTERP_NEAREST: return km_warp_launch<T, CM, KM_INTERP_NEAREST>(bwd, a, s);
        case KM_INTERP_BILINEAR: return km_warp_launch<T, CM, KM_INTERP_BILINEAR>(bwd, a, s);
        default: return km_warp_launch<T, CM, KM_INTERP_BICUBIC>(bwd, a, s);
    }
}

template <typename T>
static int km_warp_run(bool bwd, const void* src, const void* mat, void* dst, const void* gout, void* gsrc, double* gmat,
                       int B, int C, int H, int W, int h, int w, int B_M, int coord_mode, int norm_coords, int interp,
                       int pad, int align, const void* fill, hipStream_t s, const void* grid = nullptr, void* ggrid = nullptr,
                       const void* apply = nullptr) {
    typedef typename KmTraits<T>::R R;
    KmWarpArgs<T> a;
    a.apply = (const uint8_t*)apply;
    a.src = (const T*)src;
    a.mat = (const R*)mat;
    a.dst = (T*)dst;
    a.gout = (const T*)gout;
    a.gsrc = (R*)gsrc;
    a.gmat = gmat;
    a.fill = (const R*)fill;
    a.grid = (const T*)grid;
    a.ggrid = (R*)ggrid;
    KmWarpGeom<R>& g = a.g;
    km_geom_init(g, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align);
    a.tiles_x = (uint32_t)((w + KM_TILE_W - 1) / KM_TILE_W);
    a.tiles_y = (uint32_t)((h + KM_TILE_H - 1) / KM_TILE_H);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B;
    KM_REQUIRE(nb < (1ull << 31), "km_warp2d: grid too large (%llu blocks)", (unsigned long long)nb);
    a.nblocks = (uint32_t)nb;
    a.reverse = 0;
    a.stream_out = 0;
    if (nb == 0) return 0;
    switch (coord_mode) {
        case KM_COORD_PERSPECTIVE: return km_warp_dispatch_interp<T, KM_COORD_PERSPECTIVE>(bwd, a, s);
        case KM_COORD_AFFINE: return km_warp_dispatch_interp<T, KM_COORD_AFFINE>(bwd, a, s);
        case KM_COORD_GRID: return km_warp_dispatch_interp<T, KM_COORD_GRID>(bwd, a, s);
        default: return km_warp_dispatch_interp<T, KM_COORD_HOMOGRAPHY>(bwd, a, s);
    }
}

static int km_warp_validate(const char* fn, const void* src, const void* mat, int B, int C, int H, int W, int h, int w,
                            int B_M, int coord_mode, int interp, int pad, const void* fill, int dtype) {
    KM_REQUIRE(src && mat, "%s: null pointer", fn);  // (grid mode passes the grid as `mat`)
    KM_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && h >= 0 && w >= 0, "%s: bad shape B=%d C=%d H=%d W=%d h=%d w=%d", fn, B, C, H, W, h, w);
    KM_REQUIRE((int64_t)H * W < (1ll << 31) && (int64_t)h * w < (1ll << 31), "%s: image plane exceeds 2^31 elements", fn);
    KM_REQUIRE(B_M == 1 || B_M == B, "%s: matrix batch %d must be 1 or %d", fn, B_M, B);
    KM_REQUIRE(coord_mode >= 0 && coord_mode <= 2, "%s: bad coord_mode %d", fn, coord_mode);
    KM_REQUIRE(interp >= 0 && interp <= 2, "%s: bad interp %d", fn, interp);
    KM_REQUIRE(pad >= 0 && pad <= 3, "%s: bad pad %d", fn, pad);
    KM_REQUIRE(pad != KM_PAD_FILL || fill, "%s: pad=fill needs fill values", fn);
    KM_REQUIRE(dtype >= 0 && dtype <= 3, "%s: bad dtype %d", fn, dtype);
    return 0;
}

// owner-computes grad_src for bilinear + zeros/fill (km_warp_bwd_tiled.hip)
int km_warp_bwd_tiled_supported(int interp, int pad, int dtype, const void* gsrc);
int km_warp_bwd_tiled_dims_ok(int h, int w);
int km_warp_bwd_tiled_run(const void* gout, const void* mat, void* gsrc, int B, int C, int H, int W, int h, int w, int B_M, int coord_mode, int norm_coords,
                          int pad, int align, int dtype, hipStream_t s);

// matrix gradient of the bilinear warps (km_warp_gm.hip)
int km_warp_gm_supported(int interp, int pad, int dtype, int H, int W, int h, int w);
int km_warp_gm_run(const void* gout, const void* src, const void* mat, double* gmat, int B, int C, int H, int W, int h, int w, int B_M,
                   int coord_mode, int norm_coords, int pad, int align, const void* fill, int dtype, hipStream_t s);

// both gradients from one read of grad_out (km_warp_bwd_fused.hip)
int km_warp_bwd_fused_supported(int interp, int pad, int dtype, int C, int H, int W, int h, int w);
size_t km_warp_bwd_fused_workspace(int B, int C, int H, int W);
int km_warp_bwd_fused_run(const void* gout, const void* src, const void* mat, void* gsrc, double* gmat, void* ws, int B, int C, int H, int W, int h, int w,
                          int B_M, int coord_mode, int norm_coords, int pad, int align, const void* fill, int dtype, hipStream_t s);

extern "C" {

// Replaces the eager grid construction + F.grid_sample of
// kornia/geometry/transform/imgwarp.py:157-174 (warp_perspective), :271-290 (warp_affine),
// :1541-1546 (homography_warp) and _fill_and_warp :293-320.
int km_warp2d_fwd(const void* src, const void* mat, void* dst, int B, int C, int H, int W, int h, int w, int B_M,
                  int coord_mode, int norm_coords, int interp, int pad, int align, const void* fill, int dtype,
                  void* stream) {
    if (B == 0 || C == 0 || h == 0 || w == 0) return 0;  // empty output
    if (km_warp_validate("km_warp2d_fwd", src, mat, B, C, H, W, h, w, B_M, coord_mode, interp, pad, fill, dtype)) return -1;
    KM_REQUIRE(dst, "km_warp2d_fwd: null dst");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case KM_F32: return km_warp_run<float>(false, src, mat, dst, nullptr, nullptr, nullptr, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align, fill, s);
        case KM_F64: return km_warp_run<double>(false, src, mat, dst, nullptr, nullptr, nullptr, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align, fill, s);
        case KM_BF16: return km_warp_run<km_bf16>(false, src, mat, dst, nullptr, nullptr, nullptr, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align, fill, s);
        default: return km_warp_run<km_f16>(false, src, mat, dst, nullptr, nullptr, nullptr, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align, fill, s);
    }
}

// km_warp2d_fwd with the augmentation layer's per-sample switch folded in (kornia/augmentation/base.py:348-393): apply (B) uint8 on
// the device; a sample whose entry is 0 is copied instead of warped.  Needs h == H and w == W (what torch.where needs too).
int km_warp2d_fwd_masked(const void* src, const void* mat, void* dst, const void* apply, int B, int C, int H, int W, int h, int w, int B_M,
                         int coord_mode, int norm_coords, int interp, int pad, int align, const void* fill, int dtype, void* stream) {
    if (B == 0 || C == 0 || h == 0 || w == 0) return 0;  // empty output
    if (km_warp_validate("km_warp2d_fwd_masked", src, mat, B, C, H, W, h, w, B_M, coord_mode, interp, pad, fill, dtype)) return -1;
    KM_REQUIRE(dst, "km_warp2d_fwd_masked: null dst");
    KM_REQUIRE(!apply || (h == H && w == W), "km_warp2d_fwd_masked: a per-sample switch needs equal source and destination sizes (%dx%d -> %dx%d)", H, W, h, w);
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case KM_F32: return km_warp_run<float>(false, src, mat, dst, nullptr, nullptr, nullptr, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align, fill, s, nullptr, nullptr, apply);
        case KM_F64: return km_warp_run<double>(false, src, mat, dst, nullptr, nullptr, nullptr, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align, fill, s, nullptr, nullptr, apply);
        case KM_BF16: return km_warp_run<km_bf16>(false, src, mat, dst, nullptr, nullptr, nullptr, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align, fill, s, nullptr, nullptr, apply);
        default: return km_warp_run<km_f16>(false, src, mat, dst, nullptr, nullptr, nullptr, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align, fill, s, nullptr, nullptr, apply);
    }
}

// Replaces autograd of the above: aten::grid_sampler_2d_backward + the reverse of the grid chain.
// gsrc: (B,C,H,W) in the COMPUTE dtype (fp32 for f32/bf16/f16 data, fp64 for f64), pre-zeroed, nullable.
// gmat: (B_M,9) fp64 accumulators, pre-zeroed, nullable.
int km_warp2d_bwd_ws(const void* gout, const void* src, const void* mat, void* gsrc, double* gmat, int B, int C, int H,
                     int W, int h, int w, int B_M, int coord_mode, int norm_coords, int interp, int pad, int align,
                     const void* fill, int dtype, void* workspace, long long workspace_bytes, void* stream) {
    if (B == 0 || C == 0) return 0;
    if (km_warp_validate("km_warp2d_bwd", src, mat, B, C, H, W, h, w, B_M, coord_mode, interp, pad, fill, dtype)) return -1;
    KM_REQUIRE(gout, "km_warp2d_bwd: null gout");
    if (!gsrc && !gmat) return 0;
    hipStream_t s = (hipStream_t)stream;
    // bilinear + zeros/fill: grad_src by the tile-owner scatter, the matrix gradient by its own forward-shaped kernel
    const bool gm_fast = gmat && km_warp_gm_supported(interp, pad, dtype, H, W, h, w);
    // border / reflection padding: the tile-owner kernel of the one-read backward is the only scatter without global atomics for these
    // modes, so it also serves a call that wants the image gradient alone (gmat == nullptr: nothing is committed)
    if (gsrc && workspace && (pad == KM_PAD_BORDER || pad == KM_PAD_REFLECTION) && km_warp_bwd_tiled_dims_ok(h, w) && !km_config().warp_bwd_generic &&
        km_warp_bwd_fused_supported(interp, pad, dtype, C, H, W, h, w) &&
        (unsigned long long)workspace_bytes >= (unsigned long long)km_warp_bwd_fused_workspace(B, C, H, W) && ((uintptr_t)workspace & 15) == 0)
        return km_warp_bwd_fused_run(gout, src, mat, gsrc, gmat, workspace, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, dtype, s);
    if (km_warp_bwd_tiled_supported(interp, pad, dtype, gsrc)) {
        if (km_warp_bwd_tiled_dims_ok(h, w)) {
            // both gradients wanted: one persistent launch that reads grad_out once (3e bytes per element instead of 4e)
            if (gmat && workspace && km_warp_bwd_fused_supported(interp, pad, dtype, C, H, W, h, w) &&
                (unsigned long long)workspace_bytes >= (unsigned long long)km_warp_bwd_fused_workspace(B, C, H, W) && ((uintptr_t)workspace & 15) == 0)
                return km_warp_bwd_fused_run(gout, src, mat, gsrc, gmat, workspace, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, dtype, s);
            // scatter first, matrix gradient second: with the alternating batch traversal (km_traversal_next) the second launch starts on the
            // part of grad_out the first one read last.  (The other order measured 1.868 against 1.845 ms per step - no better than a fixed direction.)
            const int rc = km_warp_bwd_tiled_run(gout, mat, gsrc, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, dtype, s);
            if (rc != 0 || !gmat) return rc;
            if (gm_fast) return km_warp_gm_run(gout, src, mat, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, dtype, s);
            gsrc = nullptr;  // matrix gradient by the generic kernel below (W < 2)
        } else {
            // planes beyond 2^30 pixels: generic atomic scatter; the caller was told not to zero grad_src (needs_zero_init == 0)
            const size_t esz = 4;  // fp32 accumulators (fp64 never reaches this branch)
            const hipError_t e = hipMemsetAsync(gsrc, 0, (size_t)B * C * H * W * esz, s);
            if (e != hipSuccess) { km_set_error("km_warp2d_bwd: hipMemsetAsync failed: %s", hipGetErrorString(e)); return (int)e; }
        }
    } else if (!gsrc && gm_fast) {  // matrix gradient only (learned-homography loops: SURVEY.md config 5)
        return km_warp_gm_run(gout, src, mat, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, pad, align, fill, dtype, s);
    }
    switch (dtype) {
        case KM_F32: return km_warp_run<float>(true, src, mat, nullptr, gout, gsrc, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align, fill, s);
        case KM_F64: return km_warp_run<double>(true, src, mat, nullptr, gout, gsrc, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align, fill, s);
        case KM_BF16: return km_warp_run<km_bf16>(true, src, mat, nullptr, gout, gsrc, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align, fill, s);
        default: return km_warp_run<km_f16>(true, src, mat, nullptr, gout, gsrc, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align, fill, s);
    }
}

// km_warp2d_bwd_ws without a workspace: the image gradient and the matrix gradient as two launches (each reads grad_out)
int km_warp2d_bwd(const void* gout, const void* src, const void* mat, void* gsrc, double* gmat, int B, int C, int H,
                  int W, int h, int w, int B_M, int coord_mode, int norm_coords, int interp, int pad, int align,
                  const void* fill, int dtype, void* stream) {
    return km_warp2d_bwd_ws(gout, src, mat, gsrc, gmat, B, C, H, W, h, w, B_M, coord_mode, norm_coords, interp, pad, align, fill, dtype, nullptr, 0, stream);
}

// Bytes of workspace with which km_warp2d_bwd_ws computes BOTH gradients from one read of grad_out (0: these modes / sizes have no
// such path - pass no workspace).  16-byte aligned device memory, contents irrelevant, not read after the call returns.
long long km_warp2d_bwd_workspace_bytes(int B, int C, int H, int W, int h, int w, int interp, int pad, int dtype) {
    int dummy = 0;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return 0;
    const bool in_image_pad = (pad == KM_PAD_BORDER || pad == KM_PAD_REFLECTION);  // (served by the one-read kernel alone)
    if (km_config().warp_bwd_generic || !km_warp_bwd_tiled_dims_ok(h, w)) return 0;
    if (!in_image_pad && !km_warp_bwd_tiled_supported(interp, pad, dtype, &dummy)) return 0;
    if (!km_warp_bwd_fused_supported(interp, pad, dtype, C, H, W, h, w)) return 0;
    return (long long)km_warp_bwd_fused_workspace(B, C, H, W);
}

// ---- explicit sampling grid: replaces F.grid_sample(input, grid) as called by remap (imgwarp.py:702) and
// HomographyWarper's cached-grid forward (homography_warper.py:182).
//   grid (B_G,h,w,2) normalised (x, y) pairs in the IMAGE dtype, B_G in {1, B}; pad 0 zeros / 1 border / 2 reflection.
int km_grid_sample2d_fwd(const void* src, const void* grid, void* dst, int B, int C, int H, int W, int h, int w, int B_G,
                         int interp, int pad, int align, int dtype, void* stream) {
    if (B == 0 || C == 0 || h == 0 || w == 0) return 0;
    if (km_warp_validate("km_grid_sample2d_fwd", src, grid, B, C, H, W, h, w, B_G, 0, interp, pad, nullptr, dtype)) return -1;
    KM_REQUIRE(dst, "km_grid_sample2d_fwd: null dst");
    KM_REQUIRE(pad != KM_PAD_FILL, "km_grid_sample2d_fwd: pad must be zeros / border / reflection");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case KM_F32: return km_warp_run<float>(false, src, grid, dst, nullptr, nullptr, nullptr, B, C, H, W, h, w, B_G, KM_COORD_GRID, 0, interp, pad, align, nullptr, s, grid);
        case KM_F64: return km_warp_run<double>(false, src, grid, dst, nullptr, nullptr, nullptr, B, C, H, W, h, w, B_G, KM_COORD_GRID, 0, interp, pad, align, nullptr, s, grid);
        case KM_BF16: return km_warp_run<km_bf16>(false, src, grid, dst, nullptr, nullptr, nullptr, B, C, H, W, h, w, B_G, KM_COORD_GRID, 0, interp, pad, align, nullptr, s, grid);
        default: return km_warp_run<km_f16>(false, src, grid, dst, nullptr, nullptr, nullptr, B, C, H, W, h, w, B_G, KM_COORD_GRID, 0, interp, pad, align, nullptr, s, grid);
    }
}

// gsrc (B,C,H,W) compute dtype, PRE-ZEROED (fp32 atomics), nullable; ggrid (B,h,w,2) compute dtype, written, nullable
// (with B_G == 1 the caller sums it over the batch, as autograd does for the reference's expand()).
int km_grid_sample2d_bwd(const void* gout, const void* src, const void* grid, void* gsrc, void* ggrid, int B, int C, int H, int W,
                         int h, int w, int B_G, int interp, int pad, int align, int dtype, void* stream) {
    if (B == 0 || C == 0 || h == 0 || w == 0) return 0;
    if (km_warp_validate("km_grid_sample2d_bwd", src, grid, B, C, H, W, h, w, B_G, 0, interp, pad, nullptr, dtype)) return -1;
    KM_REQUIRE(gout, "km_grid_sample2d_bwd: null gout");
    KM_REQUIRE(pad != KM_PAD_FILL, "km_grid_sample2d_bwd: pad must be zeros / border / reflection");
    if (!gsrc && !ggrid) return 0;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case KM_F32: return km_warp_run<float>(true, src, grid, nullptr, gout, gsrc, nullptr, B, C, H, W, h, w, B_G, KM_COORD_GRID, 0, interp, pad, align, nullptr, s, grid, ggrid);
        case KM_F64: return km_warp_run<double>(true, src, grid, nullptr, gout, gsrc, nullptr, B, C, H, W, h, w, B_G, KM_COORD_GRID, 0, interp, pad, align, nullptr, s, grid, ggrid);
        case KM_BF16: return km_warp_run<km_bf16>(true, src, grid, nullptr, gout, gsrc, nullptr, B, C, H, W, h, w, B_G, KM_COORD_GRID, 0, interp, pad, align, nullptr, s, grid, ggrid);
        default: return km_warp_run<km_f16>(true, src, grid, nullptr, gout, gsrc, nullptr, B, C, H, W, h, w, B_G, KM_COORD_GRID, 0, interp, pad, align, nullptr, s, grid, ggrid);
    }
}

// 1 if km_warp2d_bwd with these modes accumulates with atomics and therefore needs gsrc zeroed by the
// caller; 0 if it overwrites gsrc completely (owner-computes path).
int km_warp2d_bwd_needs_zero_init(int interp, int pad, int dtype) {
    int dummy = 0;
    return km_warp_bwd_tiled_supported(interp, pad, dtype, &dummy) ? 0 : 1;
}

}  // extern "C"
