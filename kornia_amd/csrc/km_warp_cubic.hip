// kornia_amd - LDS-staged bicubic forward warp for gfx950 (BASELINE config 4: `warp_affine(mode="bicubic")` at 64x1x1080x1920).
// Reference: F.grid_sample(mode="bicubic") behind kornia/geometry/transform/imgwarp.py:157-174, :271-290, :1541-1546.
// Own translation unit: built with -fno-slp-vectorize (kornia_amd/build.py) - the kernel is VALU-issue bound and the SLP
// vectorizer's v_pk_mul_f32 / v_pk_add_f32 pairs cost more issue time than the scalar ops they replace once the v_mov
// shuffles that feed them are counted (profiles/r02_valu_rates.txt: v_pk_* 2.1 ns for two results, v_mul / v_add 1.2 ns each).
#include "km_warp_args.h"
#include "km_warp_stage.h"

// ------------------------------------------------------------------------------------------------
// LDS-staged BICUBIC forward (zeros padding, RGB / grey, fp32 compute) - BASELINE config 4's `warp_affine(mode="bicubic")`.
//
// A bicubic sample reads a 4 x 4 footprint: as gathers that is four unaligned 16-byte loads per pixel and channel (64 bytes
// requested per output pixel - the texture-address path, not HBM, bounds the generic kernel: 0.85 ms at 64x1x1080x1920 for
// 1.06 GB of compulsory traffic), and neighbouring pixels re-request 3/4 of each other's taps.  Here the block stages the
// source box of its KMF_T x KMF_T output tile in LDS once (16-byte row loads, zeros outside the image = what zeros padding
// reads there, so image borders need no special case) and every pixel reads its 16 taps from LDS.
//   * arithmetic = the generic kernel's, operation for operation: positions from the lean front end (bit-identical to
//     km_gen_coord, km_lean.h), km_cubic_coeffs, row sums t0 c0 + t1 c1 + t2 c2 + t3 c3, column sum the same way -> results are
//     bit-identical to km_warp_fwd_kernel<INTERP = bicubic> (tests compare the two);
//   * a wave whose footprints are not all inside the staged box (box larger than the LDS tile under minification, NaN
//     positions, vanishing line) gathers its taps from global memory with the generic kernel's per-tap bounds test.
#ifndef KMQ_FILL_DMA
#define KMQ_FILL_DMA 1   // fp32 storage: the source box by LDS-DMA (kmf_stage_box_dma)
#endif
#ifndef KMQ_TW
#define KMQ_TW 64
#define KMQ_TH 32
#define KMQ_PITCH 80
#define KMQ_ROWS 80
#endif
// TOLERANCE SPENT (round 6, KMQ_FMA = 1).  BASELINE.json asks for 1e-5 against the reference; rounds 2-5 kept this kernel's arithmetic operation for
// operation that of the generic kernel (35 vector instructions per channel for the 4 x 4 taps - 20 multiplies, 15 adds, every product rounded
// - and 25 per axis for the coefficients), bit-identical to it, <= 1e-6 from the reference (whose own vectorised CPU kernel sums in another
// order) - and sat on its issue time: ~125 vector instructions per grey pixel, 0.37 ms = 0.36 of the HBM roofline at config 4.  With fused
// multiply-adds the tap sums are 1 multiply + 3 fma per row and per column (20 per channel) and the coefficient polynomials Horner chains
// (15 per axis): the same association, products no longer rounded before they are added - each result at least as close to the exact value
// as the reference's, <= 2e-6 from the oracle and <= 1e-5 from the live reference (asserted: tests/test_gpu_config_parity.py,
// tests/test_gpu_warp.py, tests/test_oracle_live_sweep.py).  Both paths of this kernel (LDS taps, gather rows) use the same form, so a result
// still never depends on the box estimate.  KMQ_FMA = 0 restores the exact form (A/B).
#ifndef KMQ_FMA
#define KMQ_FMA 1
#endif
__device__ __forceinline__ float kmq_dot4(float t0, float t1, float t2, float t3, const float (&c)[4]) {
#if KMQ_FMA
    return km_fma(t3, c[3], km_fma(t2, c[2], km_fma(t1, c[1], t0 * c[0])));
#else
    return t0 * c[0] + t1 * c[1] + t2 * c[2] + t3 * c[3];
#endif
}
__device__ __forceinline__ void kmq_coeffs(float t, float (&c)[4]) {
#if KMQ_FMA
    const float A = -0.75f;
    float x = t + 1.0f;
    c[0] = km_fma(km_fma(km_fma(A, x, -5 * A), x, 8 * A), x, -4 * A);
    x = t;
    c[1] = km_fma(km_fma(A + 2, x, -(A + 3)) * x, x, 1.0f);
    x = 1.0f - t;
    c[2] = km_fma(km_fma(A + 2, x, -(A + 3)) * x, x, 1.0f);
    x = x + 1.0f;
    c[3] = km_fma(km_fma(km_fma(A, x, -5 * A), x, 8 * A), x, -4 * A);
#else
    km_cubic_coeffs(t, c);
#endif
}
#ifndef KMQ_ST16
#define KMQ_ST16 0   // 1: fp32 storage, the thread's results leave as 16-byte stores after a 4 x 4 transpose inside each quad of lanes (km_quad_transpose4).
                     // MEASURED SLOWER (round 6, profiles/r06/run2_*, same box, same bits): config 4 361.6 / 359.8 us against 333.7 / 331.2 with one dword
                     // per lane and row (and 397.8 against 365.2 on the exact-rounding form): like the box forward, DESIGN.md 4.6
#endif

template <typename T, int CM, int NC, int ALIGN>
__device__ __forceinline__ void kmq_rows_gather(const KmWarpArgs<T>& a, const float (&m)[9], const float4* s_rv, const KmlHalf& cu, bool fast, uint32_t b,
                                                          int j, int li_base, int i_base, int rstep, int rpt) {
    const KmWarpGeom<float>& g = a.g;
    const int W = g.W, H = g.H;
    const float Wm1 = (float)(W - 1), Hm1 = (float)(H - 1), hW = (float)W / 2, hH = (float)H / 2;
    const size_t src_plane = (size_t)H * W, dst_plane = (size_t)g.h * g.w;
    const T* __restrict__ src_b = a.src + (size_t)b * NC * src_plane;
    T* __restrict__ dst_b = a.dst + (size_t)b * NC * dst_plane;
    for (int r = 0; r < rpt; ++r) {
        const int i = i_base + r * rstep;
        if (i >= g.h) break;
        const float4 rv4 = s_rv[li_base + r * rstep];
        KmlHalf rv;
        rv.a = rv4.x; rv.b = rv4.y; rv.c = rv4.z;
        KmlPos p;
        if (fast) kml_position<CM, true>(m, cu, rv, p);
        else kml_position<CM, false>(m, cu, rv, p);
        const float x = kml_unnormalize<ALIGN>(p.gx, Wm1, hW), y = kml_unnormalize<ALIGN>(p.gy, Hm1, hH);
        const float xf = km_floor(x), yf = km_floor(y);
        float cx[4], cy[4];
        kmq_coeffs(x - xf, cx);
        kmq_coeffs(y - yf, cy);
        int idx[4][4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int q = 0; q < 4; ++q) idx[rr][q] = km_tap_index(xf - 1 + q, yf - 1 + rr, W, H, KM_PAD_ZEROS, 0);
        T* __restrict__ out_px = dst_b + (size_t)i * g.w + j;
        for (int c = 0; c < NC; ++c) {
            const T* img = src_b + (size_t)c * src_plane;
            float rows[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float tt[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) tt[q] = idx[rr][q] >= 0 ? km_ld(img + idx[rr][q]) : 0.0f;
                rows[rr] = kmq_dot4(tt[0], tt[1], tt[2], tt[3], cx);
            }
            km_st(out_px + (size_t)c * dst_plane, kmq_dot4(rows[0], rows[1], rows[2], rows[3], cy));
        }
    }
}

// TW x TH output tile (TW in {32, 64}: a wave covers TW x (64 / TW) pixels), PITCH x ROWS staged box
template <typename T, int CM, int NC, int ALIGN, int TW, int TH, int PITCH, int ROWS>
__global__ __launch_bounds__(256) void km_warp_fwd_cubic_kernel(const KmWarpArgs<T> a) {
    constexpr int RSTEP = 256 / TW, RPT = TH / RSTEP;  // tile rows between a thread's consecutive rows, rows per thread
    static_assert((TW == 32 || TW == 64) && TH % RSTEP == 0, "tile shape");
    const KmWarpGeom<float>& g = a.g;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t b = bid / a.tiles_y;
    const int tid = threadIdx.x;
    const int j = (int)tx * TW + (tid % TW);
    const int li_base = tid / TW;                 // this thread's rows: li_base + r * RSTEP
    const int i_base = (int)ty * TH + li_base;
    __shared__ float4 s_rv[TH];
    __shared__ int s_info[8];
    __shared__ __attribute__((aligned(16))) float s_src[ROWS * NC * PITCH];  // [row][channel][x]

    if (a.apply && !a.apply[b]) {  // block-uniform: this sample is not transformed
        km_fwd_copy_rows<T>(a, b, j, i_base, RSTEP, RPT);
        return;
    }
    float m[9];
    {
        const float* mp = a.mat + (size_t)(g.B_M == 1 ? 0 : b) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = mp[k];
    }
    kmf_tile_setup<CM, ALIGN, 1, TW, TH, PITCH, ROWS>(g, m, (int)tx * TW, (int)ty * TH, s_rv, s_info, false);
    __syncthreads();
    const KmfBox bx = kmf_read_box(s_info);
    const int W = g.W, H = g.H;
    const float Wm1 = (float)(W - 1), Hm1 = (float)(H - 1), hW = (float)W / 2, hH = (float)H / 2;
    const size_t src_plane = (size_t)H * W, dst_plane = (size_t)g.h * g.w;
    const T* __restrict__ src_b = a.src + (size_t)b * NC * src_plane;
    T* __restrict__ dst_b = a.dst + (size_t)b * NC * dst_plane;
    if (bx.staged) {  // block-uniform
        float oob[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) oob[c] = 0.f;
#if KMQ_FILL_DMA
        if constexpr (sizeof(T) == 4) {
            // fp32 storage: the whole box by LDS-DMA, every pass in flight together (the register copy is three round trips for an 80-row box)
            kmf_stage_box_dma<NC, PITCH, ROWS>(reinterpret_cast<const float*>(src_b), src_plane, W, H, bx, s_src);
            KM_VMCNT0();
        } else
#endif
            kmf_stage_box_dyn<T, NC, PITCH>(src_b, src_plane, W, H, bx, s_src, oob);
        __syncthreads();
    }
    if (j >= g.w) return;

    const KmlHalf cu = kml_col_half<CM>(m, km_base_x<float, CM>(g, j));
    T* __restrict__ out_col = dst_b + j;
    bool all_in = bx.staged;
    if (bx.staged) {
        // ---- hot loop: no branch inside.  Every footprint is expected in the box (the box is the hull of the tile's corners + a
        // margin); the loop only RECORDS whether one was not, reads from a clamped LDS offset in that case, and the wave redoes its
        // rows with gathers afterwards - so the result never depends on the box estimate, and the common case pays four compares.
        // The LDS offset of tap (-1, -1) in floats: (yf - 1 - ys) * NC * PITCH + (xf - 1 - xs), exact for every in-box footprint.
        // The results stay in registers until the wave knows that all its footprints were in the box (nothing is stored twice).
        const float xrel = (float)(bx.xs + 1), yrel = (float)(bx.ys + 1);
        const float off_max = (float)(ROWS * NC * PITCH - (3 * NC * PITCH + (NC - 1) * PITCH + 4));
        float res[NC][RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
#pragma unroll
            for (int c = 0; c < NC; ++c) res[c][r] = 0.f;
            if ((int)ty * TH + r * RSTEP >= g.h) continue;  // block-uniform (RSTEP rows of the tile at a time)
            const int i = i_base + r * RSTEP;
            const bool row_ok = i < g.h;  // (TW == 64: wave-uniform)
            const float4 rv4 = s_rv[li_base + r * RSTEP];
            KmlHalf rv;
            rv.a = rv4.x; rv.b = rv4.y; rv.c = rv4.z;
            KmlPos p;
            if (bx.fast) kml_position<CM, true>(m, cu, rv, p);
            else kml_position<CM, false>(m, cu, rv, p);
            const float x = kml_unnormalize<ALIGN>(p.gx, Wm1, hW), y = kml_unnormalize<ALIGN>(p.gy, Hm1, hH);
            const float xf = km_floor(x), yf = km_floor(y);
            float cx[4], cy[4];
            kmq_coeffs(x - xf, cx);
            kmq_coeffs(y - yf, cy);
            all_in = all_in & (kmf_in_box_cubic(xf, yf, bx) | !row_ok);
            const float offf = fminf(fmaxf(km_fma(yf - yrel, (float)(NC * PITCH), xf - xrel), 0.0f), off_max);  // (NaN -> 0)
            const float* q = s_src + KM_F2I(offf);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float rows[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float* qr = q + (rr * NC + c) * PITCH;
                    rows[rr] = kmq_dot4(qr[0], qr[1], qr[2], qr[3], cx);
                }
                res[c][r] = kmq_dot4(rows[0], rows[1], rows[2], rows[3], cy);
            }
        }
        if (__all(all_in)) {  // wave-uniform: every footprint of the wave was in the box - the results are final
#if KMQ_ST16
            if constexpr (sizeof(T) == 4 && RPT % 4 == 0) {
                // 16-byte stores: the quad's 4 columns x 4 of the thread's rows (RSTEP apart) transposed in registers, lane q of the quad writes
                // the four pixels of row r0 + q.  (Output rows are 16-byte aligned runs of whole quads: lanes right of the image left above.)
                if (((g.w & 3) == 0) && (((uintptr_t)a.dst & 15) == 0)) {
                    const int q4 = tid & 3;
#pragma unroll
                    for (int r0 = 0; r0 < RPT; r0 += 4) {
                        if ((int)ty * TH + r0 * RSTEP >= g.h) break;  // block-uniform
                        const int i = i_base + (r0 + q4) * RSTEP;
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            float v4[4] = {res[c][r0], res[c][r0 + 1], res[c][r0 + 2], res[c][r0 + 3]};
                            km_quad_transpose4(v4, q4);
                            if (i < g.h) {
                                float* p = reinterpret_cast<float*>(dst_b) + (size_t)c * dst_plane + (size_t)i * g.w + (size_t)(j & ~3);
                                KM_CHECK_ALIGNED(p, 16);
                                km_st4_c<true>(p, v4);
                            }
                        }
                    }
                    return;
                }
            }
#endif
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int i = i_base + r * RSTEP;
                if (i < g.h) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) km_st(out_col + (size_t)c * dst_plane + (size_t)i * g.w, res[c][r]);
                }
            }
            return;
        }
    }
    // ---- cold path: the box did not fit the LDS tile (minification), or a footprint of this wave was outside it: per-tap gathers
    kmq_rows_gather<T, CM, NC, ALIGN>(a, m, s_rv, cu, bx.fast, b, j, li_base, i_base, RSTEP, RPT);
}

template <typename T, int CM, int NC, int TW, int TH, int PITCH, int ROWS>
static void km_warp_fwd_cubic_launch_nc(const KmWarpArgs<T>& a, hipStream_t s) {
    const uint32_t tiles_x = (uint32_t)((a.g.w + TW - 1) / TW), tiles_y = (uint32_t)((a.g.h + TH - 1) / TH);
    KmWarpArgs<T> b = a;
    b.tiles_x = tiles_x; b.tiles_y = tiles_y; b.nblocks = tiles_x * tiles_y * (uint32_t)a.g.B;
    if (a.g.align)
        hipLaunchKernelGGL((km_warp_fwd_cubic_kernel<T, CM, NC, 1, TW, TH, PITCH, ROWS>), dim3(b.nblocks), dim3(256), 0, s, b);
    else
        hipLaunchKernelGGL((km_warp_fwd_cubic_kernel<T, CM, NC, 0, TW, TH, PITCH, ROWS>), dim3(b.nblocks), dim3(256), 0, s, b);
}
// bicubic + zeros, RGB / grey, rows of whole 4-element chunks, fp32 compute; KM_WARP_FWD_ALGO=generic keeps the per-pixel gathers (A/B timing; checked by the caller).
// Grey: 64 x 32 output tiles with an 80 x 80 box (25.6 KB: fits any rotation at scale >= 1); RGB: 32 x 32 tiles with a 56 x 56 box per channel.
template <typename T, int CM>
static bool km_warp_fwd_cubic_try(const KmWarpArgs<T>& a, hipStream_t s) {
    if constexpr (CM != KM_COORD_GRID && sizeof(typename KmTraits<T>::R) == sizeof(float)) {
        const uint64_t nb = (uint64_t)((a.g.w + 31) / 32) * (uint64_t)((a.g.h + 31) / 32) * (uint64_t)a.g.B;
        if (a.g.pad != KM_PAD_ZEROS || !(a.g.C == 3 || a.g.C == 1) || (a.g.W & 3) != 0 || ((uintptr_t)a.src % (4 * sizeof(T))) != 0 || nb >= (1ull << 31) ||
            a.g.W >= (1 << 23) || a.g.H >= (1 << 23))
            return false;
        if (a.g.C == 3) km_warp_fwd_cubic_launch_nc<T, CM, 3, 32, 32, 56, 56>(a, s);
        else km_warp_fwd_cubic_launch_nc<T, CM, 1, KMQ_TW, KMQ_TH, KMQ_PITCH, KMQ_ROWS>(a, s);
        return true;
    }
    return false;
}


int km_warp_fwd_cubic_try_any(int dtype, int coord_mode, const void* args, hipStream_t s) {
#define KMQ_CM(T)                                                                                                              \
    switch (coord_mode) {                                                                                                      \
        case KM_COORD_PERSPECTIVE: return km_warp_fwd_cubic_try<T, KM_COORD_PERSPECTIVE>(*(const KmWarpArgs<T>*)args, s) ? 1 : 0; \
        case KM_COORD_AFFINE: return km_warp_fwd_cubic_try<T, KM_COORD_AFFINE>(*(const KmWarpArgs<T>*)args, s) ? 1 : 0;           \
        case KM_COORD_HOMOGRAPHY: return km_warp_fwd_cubic_try<T, KM_COORD_HOMOGRAPHY>(*(const KmWarpArgs<T>*)args, s) ? 1 : 0;   \
        default: return 0;                                                                                                     \
    }
    switch (dtype) {
        case KM_F32: KMQ_CM(float)
        case KM_BF16: KMQ_CM(km_bf16)
        case KM_F16: KMQ_CM(km_f16)
        default: return 0;  // fp64 computes in double: the generic kernel
    }
#undef KMQ_CM
}
