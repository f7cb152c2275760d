// kornia_amd - register-tiled separable filter for small odd kernels (the GaussianBlur2d hot path).
//
// Why not LDS here: a K-tap separable blur re-uses each input K times in x and K times in y.  With
// 16-byte loads per lane the x re-use happens in registers (each lane owns 4 adjacent columns and
// reads the two neighbouring float4s, which are L1 hits because its neighbour lanes load them too)
// and the y re-use is a rolling window of K row-pass results held in registers while the lane walks
// down its strip.  No LDS traffic, no __syncthreads, full 16-byte coalesced NCHW reads and writes.
// HBM traffic = read x once (halo rows/columns come out of L1/L2) + write y once = 2e bytes/element.
//
//   forward  (BWD = false): taps constant, border = index map (rows: address select; columns: the two
//            edge lanes of a row rebuild their out-of-image float4 by a static register swizzle).
//            fma chain in tap order from 0 - bit-identical to oracle/ko_impl.h ko_filter2d_fwd.
//   adjoint  (BWD = true): gx = A^T gy with A = A_y (x) A_x.  Each 1-D adjoint is a correlation of the
//            ZERO-extended gradient with position-dependent taps  W[p][i] = sum_t k[t] [map(i+t-l) == p]
//            which differ from the flipped taps only within K-1 pixels of a border (the pad fold).
//            Circular is periodic: unmodified flipped taps on periodic indexing.
//
// Requirements checked by the host: K odd, 3 <= K <= 9, kW == kH == K, 'same' padding, W % 4 == 0,
// W >= 8, H >= K, 16-byte aligned planes.  Everything else takes the generic LDS kernel.
#include <stdlib.h>

#include "km_regtile.h"


#ifdef KMB_ROWS_OVERRIDE
#define KMB_ROWS KMB_ROWS_OVERRIDE
#else
#define KMB_ROWS 32  // output rows per thread (strip height) of the large form
#endif
// A wave walks its strip row by row: (ROWS + K - 1) / ROWS input rows are read per output row (the extra ones mostly L2 hits), and a launch with few
// waves per SIMD has little to overlap its row loads with.  Three strip heights are compiled - 32, 16 and 8 rows - and km_blur_rows() picks one per
// launch from what profiles/r04/run49_blur_strip_heights.txt measured (round 2 had measured 8 -> 0.32 ms, 16 -> 0.31, 32 -> 0.295 at 256x3x512^2 on the
// kernel of that round; with the row loads one row ahead, round 4, the forward is fastest at 16):
//   fp32   few waves (16x3x512^2: 37.2 -> 24.6 us forward, 44.5 -> 30.8 adjoint; 4x3x512^2: 21.0 -> 13.1): 16
//          K = 3: 16 (294-307 -> 278-281 us forward, 297-309 -> 280-283 adjoint); K = 5: forward 16 (298-302 -> 290-291), adjoint 32 (295-298 against
//          305); K >= 7: 32 (the adjoint's position-dependent taps: K = 7 413 against 464 us, K = 9 824 against 1 031)
//   16-bit 32 (256x3x512^2: 201 / 229 us forward / adjoint against 211 / 258 with 16; 256x3x224^2, BASELINE config 3: 54-56 / 62 us against 56-58 / 71 with
//          16 and 55-58 / 82 with 8 - round 3's kernel had gained from 8-row strips for such launches, 56.2 -> 52.0 us; with the row loads one row ahead
//          it no longer does: profiles/r04/run49_blur_strip_heights.txt, second table)
// The arithmetic per output pixel is the same in all three: bit-identical results (tests force each height: KM_BLUR_ROWS / km_config_set).
#define KMB_ROWS_MID 16
#define KMB_ROWS_SMALL 8
#define KMB_FEW_WAVES_F32 4096       // fp32: 4 waves per SIMD of the 32-row form

template <typename T>
struct KmVec4;
template <>
struct KmVec4<float> {
    typedef float4 V;
};
template <>
struct KmVec4<km_bf16> {
    typedef uint2 V;
};
template <>
struct KmVec4<km_f16> {
    typedef uint2 V;
};

__device__ __forceinline__ float km_round_store(float v, const float*) { return v; }
__device__ __forceinline__ float km_round_store(float v, const km_bf16*) { return __uint_as_float(((uint32_t)km_f32_to_bf16_bits(v)) << 16); }
__device__ __forceinline__ float km_round_store(float v, const km_f16*) { KM_OPAQUE(v); return (float)(_Float16)v; }  // (never fused into the fma before it: km_common.h)


template <typename T>
struct KmBlurArgs {
    const T* x;       // fwd: input ; bwd: grad_out      (B*C, H, W)
    T* y;             // fwd: output ; bwd: grad_in
    const float* kx;  // (Bk, K)
    const float* ky;  // (Bk, K)
    int C, H, W, Bk, border;
    uint32_t groups_x;   // W / 4
    uint32_t bx, by;     // blocks per plane in x / y
    uint32_t nblocks;
    uint32_t reverse;    // the XCDs walk their block ranges backwards (km_traversal_next)
    uint32_t rows;       // strip height of the launch (host-side choice of the instantiation: KMB_ROWS, KMB_ROWS_MID or KMB_ROWS_SMALL)
    uint32_t stream_out; // streaming stores (km_stream_stores)
};

// Adjoint taps along one axis for output position p:  w[d] multiplies the (zero-extended) gradient at
// p - R + d, d in [0,K), R = K - 1 - L = rear pad, L = front pad = (K-1)/2:
//   w[d] = sum_t k[t] * [ map(i + t - L) == p ],  i = p - R + d
template <int K>
__device__ __forceinline__ void kmb_adjoint_taps(const float (&k)[K], int p, int n, int border, float (&w)[K]) {
    constexpr int L = (K - 1) / 2, R = K - 1 - L;
    const bool interior = (border == KM_BORDER_CIRCULAR) || (border == KM_BORDER_CONSTANT) || (p >= K - 1 && p <= n - K);
#pragma unroll
    for (int d = 0; d < K; ++d) w[d] = k[K - 1 - d];  // interior: i + t - L == p  <=>  t = p - i + L = R + L - d
    if (interior) return;
#pragma unroll
    for (int d = 0; d < K; ++d) {
        const int i = p - R + d;
        float acc = 0.f;
        if (i >= 0 && i < n) {
#pragma unroll
            for (int t = 0; t < K; ++t)
                if (km_border_map(i + t - L, n, border) == p) acc += k[t];
        }
        w[d] = acc;
    }
}

#ifndef KMB_BWD_WAVES
#define KMB_BWD_WAVES 1   // minimum waves per SIMD asked of the adjoint's register allocation (8: at most 64 registers)
#endif
template <typename T, int K, bool BWD, int ROWS = KMB_ROWS>
__global__ __launch_bounds__(256, (BWD ? KMB_BWD_WAVES : 1)) void km_blur_reg_kernel(const KmBlurArgs<T> a) {
    constexpr int L = (K - 1) / 2, R = K - 1 - L;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks, a.reverse);
    const uint32_t tbx = bid % a.bx;
    bid /= a.bx;
    const uint32_t tby = bid % a.by;
    const uint32_t bc = bid / a.by;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (wave-uniform: the strip's rows are scalar values)
    const int gx = (int)tbx * 64 + lane;                      // column group (4 px)
    const int r0 = ((int)tby * 4 + wave) * ROWS;              // first output row of this thread's strip
    if (gx >= (int)a.groups_x || r0 >= a.H) return;
    const int H = a.H, W = a.W, border = a.border;
    const int c0 = gx * 4;
    const bool left = (c0 == 0), right = (c0 == W - 4);
    const T* img = a.x + (size_t)bc * H * W;
    T* out = a.y + (size_t)bc * H * W;
    const int b = (int)(bc / a.C);

    float kx[K], ky[K];
    {
        const float* px = a.kx + (size_t)(b % a.Bk) * K;
        const float* py = a.ky + (size_t)(b % a.Bk) * K;
#pragma unroll
        for (int t = 0; t < K; ++t) { kx[t] = px[t]; ky[t] = py[t]; }
    }
    // horizontal taps: forward uses kx as is; the adjoint uses per-output position-dependent taps
    float wx[4][K];
    if (BWD) {
#pragma unroll
        for (int o = 0; o < 4; ++o) kmb_adjoint_taps<K>(kx, c0 + o, W, border, wx[o]);
    }

    // column offsets of the three float4 loads; edge lanes clamp (or wrap, for circular) the outer ones
    int offL = c0 - 4, offR = c0 + 4;
    if (left) offL = (border == KM_BORDER_CIRCULAR) ? W - 4 : c0;
    if (right) offR = (border == KM_BORDER_CIRCULAR) ? 0 : c0;

    float ring[K][4];  // rolling window of row-pass results
    const int n_rows = (r0 + ROWS <= H ? ROWS : H - r0);
    const int total = n_rows + K - 1;
    const bool e0 = (lane == 0), e63 = (lane == 63);

    // The row loads run ONE ROW AHEAD of the arithmetic: a wave walks its strip row by row (load -> row pass -> column pass -> store), and with
    // the wait for a row's load directly in front of its row pass every wave had exactly one 1 KB request in flight - ~8 MB across the chip against
    // the ~16 MB that 8 TB/s times the memory latency ask for.  The request of row it + 1 is issued before row it is processed (8 more registers).
    // (Circular borders load their two neighbour chunks in place, as before.)
    float on4[4] = {0.f, 0.f, 0.f, 0.f}, en4[4] = {0.f, 0.f, 0.f, 0.f};
    int srow_n = -1;
    auto request = [&](int it) {
        const int rin = r0 + it - (BWD ? R : L);
        int sr;
        if (BWD) sr = (border == KM_BORDER_CIRCULAR) ? km_border_map(rin, H, KM_BORDER_CIRCULAR) : ((rin >= 0 && rin < H) ? rin : -1);
        else sr = km_border_map(rin, H, border);
        srow_n = sr;
        if (sr >= 0) {
            const T* rowp = img + (size_t)sr * W;
            km_ld4(rowp + c0, on4);
            // the two chunks no lane holds - left of lane 0, right of lane 63 - come with ONE load (see below)
            if (border != KM_BORDER_CIRCULAR && (e0 || e63)) km_ld4(rowp + (e0 ? offL : offR), en4);
        }
    };
    request(0);

    for (int it0 = 0; it0 < total; it0 += K) {
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int it = it0 + kk;
            if (it < total) {
                // ---- row pass for input row (r0 - L + it) [fwd] / (r0 - R + it) [bwd] ----
                const int srow = srow_n;
                float v[12];
                float o4[4], e4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { o4[q] = on4[q]; e4[q] = en4[q]; }
                if (it + 1 < total) request(it + 1);
                if (srow >= 0) {
                    const T* rowp = img + (size_t)srow * W;
                    float l4[4], r4[4];
                    if (border == KM_BORDER_CIRCULAR) {
                        km_ld4(rowp + offL, l4);
                        km_ld4(rowp + offR, r4);
                    } else {
                        // The neighbouring chunks are what the neighbouring lanes have just loaded: take them from their registers
                        // (one wave-wide DPP shift each) instead of requesting them from memory again.  The texture-address
                        // unit is the busiest unit of this kernel (TA_TA_BUSY 86 % with three 16-byte loads per row,
                        // profiles/r02_pmc_units.json) and it charges ~25 cycles per wave instruction however few lanes
                        // are active, so the two chunks no lane holds - left of lane 0, right of lane 63 - come with ONE load.
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float pl = km_prev64(o4[q]), nr = km_next64(o4[q]);
                            l4[q] = e0 ? e4[q] : pl;
                            r4[q] = e63 ? e4[q] : nr;
                        }
                    }
                    if (border != KM_BORDER_CIRCULAR) {
                        if (BWD || border == KM_BORDER_CONSTANT) {
                            if (left) { l4[0] = l4[1] = l4[2] = l4[3] = 0.f; }
                            if (right) { r4[0] = r4[1] = r4[2] = r4[3] = 0.f; }
                        } else if (border == KM_BORDER_REFLECT) {
                            // positions -4..-1 -> x[4], x[3], x[2], x[1] ; W..W+3 -> x[W-2], x[W-3], x[W-4], x[W-5]
                            if (left) { l4[0] = r4[0]; l4[1] = o4[3]; l4[2] = o4[2]; l4[3] = o4[1]; }
                            if (right) { const float t3 = l4[3]; r4[0] = o4[2]; r4[1] = o4[1]; r4[2] = o4[0]; r4[3] = t3; }
                        } else {  // replicate
                            if (left) { l4[0] = l4[1] = l4[2] = l4[3] = o4[0]; }
                            if (right) { r4[0] = r4[1] = r4[2] = r4[3] = o4[3]; }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] = l4[q]; v[4 + q] = o4[q]; v[8 + q] = r4[q]; }
                } else {
#pragma unroll
                    for (int q = 0; q < 12; ++q) v[q] = 0.f;
                }
                // v[i] holds column c0 - 4 + i
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float acc = 0.f;
                    if (BWD) {
                        // taps wx[o][d] multiply column (c0+o) - R + d
#pragma unroll
                        for (int d = 0; d < K; ++d) acc = km_fma(wx[o][d], v[4 + o - R + d], acc);
                    } else {
#pragma unroll
                        for (int q = 0; q < K; ++q) acc = km_fma(kx[q], v[4 + o - L + q], acc);
                    }
                    ring[kk][o] = km_round_store(acc, (const T*)nullptr);
                }
                // ---- column pass: emits output row r = r0 + it - (K-1) once K rows are in the window ----
                if (it >= K - 1) {
                    const int r = r0 + it - (K - 1);
                    float res[4];
                    if (BWD) {
                        float wy[K];
                        kmb_adjoint_taps<K>(ky, r, H, border, wy);
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            float acc = 0.f;
#pragma unroll
                            for (int d = 0; d < K; ++d) acc = km_fma(wy[d], ring[(kk + 1 + d) % K][o], acc);
                            res[o] = acc;
                        }
                    } else {
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            float acc = 0.f;
#pragma unroll
                            for (int p = 0; p < K; ++p) acc = km_fma(ky[p], ring[(kk + 1 + p) % K][o], acc);
                            res[o] = acc;
                        }
                    }
                    km_st4_pol(out + (size_t)r * W + c0, res, a.stream_out != 0u);
                }
            }
        }
    }
}

template <typename T, int K, int ROWS>
static void km_blur_launch_rows(bool bwd, const KmBlurArgs<T>& a, hipStream_t s) {
    if (bwd)
        hipLaunchKernelGGL((km_blur_reg_kernel<T, K, true, ROWS>), dim3(a.nblocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((km_blur_reg_kernel<T, K, false, ROWS>), dim3(a.nblocks), dim3(256), 0, s, a);
}
template <typename T, int K>
static int km_blur_launch(bool bwd, const KmBlurArgs<T>& a, hipStream_t s) {
    if (a.rows == KMB_ROWS_SMALL) km_blur_launch_rows<T, K, KMB_ROWS_SMALL>(bwd, a, s);
    else if (a.rows == KMB_ROWS_MID) km_blur_launch_rows<T, K, KMB_ROWS_MID>(bwd, a, s);
    else km_blur_launch_rows<T, K, KMB_ROWS>(bwd, a, s);
    return km_check_launch(bwd ? "km_blur_reg_bwd" : "km_blur_reg_fwd");
}

// strip height of a launch (see the table at KMB_ROWS); `force`: KM_BLUR_ROWS / km_config_set("blur_rows") = 8 / 16 / 32
static int km_blur_rows(bool bwd, size_t elem, int K, uint64_t waves_big, int force) {
    if (force == KMB_ROWS_SMALL || force == KMB_ROWS_MID || force == KMB_ROWS) return force;
    if (elem == 2) return KMB_ROWS;
    if (waves_big < KMB_FEW_WAVES_F32 || K == 3) return KMB_ROWS_MID;
    if (K == 5) return bwd ? KMB_ROWS : KMB_ROWS_MID;
    return KMB_ROWS;
}

template <typename T>
static int km_blur_run(bool bwd, const void* x, const void* kx, const void* ky, void* y, int B, int C, int H, int W, int Bk,
                       int K, int border, hipStream_t s) {
    KmBlurArgs<T> a;
    a.x = (const T*)x; a.y = (T*)y; a.kx = (const float*)kx; a.ky = (const float*)ky;
    a.C = C; a.H = H; a.W = W; a.Bk = Bk; a.border = border;
    a.groups_x = (uint32_t)(W / 4);
    a.bx = (a.groups_x + 63) / 64;
    const int force_rows = km_config().blur_rows;  // 8 / 16 / 32: forces the instantiation (tests, A/B timing: KM_BLUR_ROWS / km_config_set)
    const uint64_t waves_big = (uint64_t)a.bx * (uint64_t)((H + KMB_ROWS - 1) / KMB_ROWS) * (uint64_t)B * C;
    const int rows = km_blur_rows(bwd, sizeof(T), K, waves_big, force_rows);
    a.rows = (uint32_t)rows;
    a.by = (uint32_t)((H + 4 * rows - 1) / (4 * rows));
    const uint64_t nb = (uint64_t)a.bx * a.by * (uint64_t)B * C;
    KM_REQUIRE(nb < (1ull << 31), "km_blur: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    a.reverse = km_traversal_next(s);
    a.stream_out = km_stream_stores((uint64_t)B * C * H * W * sizeof(T));
    switch (K) {
        case 3: return km_blur_launch<T, 3>(bwd, a, s);
        case 5: return km_blur_launch<T, 5>(bwd, a, s);
        case 7: return km_blur_launch<T, 7>(bwd, a, s);
        default: return km_blur_launch<T, 9>(bwd, a, s);
    }
}

// 1 if the register-tiled kernel handles this problem
int km_blur_fast_supported(const void* x, const void* y, int H, int W, int kH, int kW, int border, int same, int dtype) {
    if (!same || kH != kW || (kH & 1) == 0 || kH < 3 || kH > 9) return 0;
    if (dtype == KM_F64) return 0;
    if ((W & 3) != 0 || W < 8 || H < kH) return 0;
    if (border == KM_BORDER_REFLECT && (kH - 1) / 2 >= (H < W ? H : W)) return 0;
    const size_t esz = (dtype == KM_F32) ? 4 : 2;
    if (((uintptr_t)x % (4 * esz)) != 0 || ((uintptr_t)y % (4 * esz)) != 0) return 0;
    if ((((size_t)H * W * esz) % (4 * esz)) != 0) return 0;
    return 1;
}

int km_blur_fast_run(bool bwd, const void* x, const void* kx, const void* ky, void* y, int B, int C, int H, int W, int Bk, int K,
                     int border, int dtype, hipStream_t s) {
    switch (dtype) {
        case KM_F32: return km_blur_run<float>(bwd, x, kx, ky, y, B, C, H, W, Bk, K, border, s);
        case KM_BF16: return km_blur_run<km_bf16>(bwd, x, kx, ky, y, B, C, H, W, Bk, K, border, s);
        default: return km_blur_run<km_f16>(bwd, x, kx, ky, y, B, C, H, W, Bk, K, border, s);
    }
}
