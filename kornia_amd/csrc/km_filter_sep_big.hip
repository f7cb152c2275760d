// kornia_amd - separable filter, forward, for LARGE kernels (odd or even, 10 <= k <= 65 per axis): the Gaussian blurs of
// SimCLR / BYOL-style augmentation (kernel ~ 10 % of the image side, e.g. 23 x 23 at 224 x 224).
// Reference: filter2d_separable, kornia/filters/filter.py:155-207 (two pad + conv2d passes, the intermediate stored in
// the input dtype).
//
// The register-tiled kernel (km_blur_fast.hip) stops at k = 9 (its K-row window lives in registers); the small-tile LDS
// kernel (km_filter.hip) issues one LDS read per multiply-add.  Here a 64 x 64 output tile is staged in LDS with its
// halo and both passes run a sliding window in registers: per block of 8 taps a lane loads 12 consecutive values (three
// aligned 16-byte LDS reads in the row pass, 11 conflict-free 4-byte reads in the column pass) and produces 4 outputs x 8
// taps = 32 fused multiply-adds from them - an LDS read per ~3-8 multiply-adds instead of per 1.  The taps are applied
// in increasing order from 0, so the result is bit-identical to oracle/ko_impl.h (ko_filter2d_fwd twice).
// HBM traffic: read x ~1.8x for k = 23 (halo, served mostly by L2) + write y once.
#include "km_regtile.h"

#define KMS_TW 64
#define KMS_TH 64
#define KMS_LDS_LIMIT (64 * 1024)


template <typename T>
struct KmSepBigArgs {
    const T* x;       // (B,C,H,W)
    T* y;             // (B,C,Ho,Wo)
    const float* kx;  // (Bk,kW)
    const float* ky;  // (Bk,kH)
    int C, H, W, Ho, Wo, Bk, kH, kW, border, same, pt, pl;
    int iw_pitch;     // LDS pitch of the staged tile (multiple of 4)
    uint32_t tiles_x, tiles_y, nblocks;
};

__device__ __forceinline__ float kms_round_to(float v, const float*) { return v; }
__device__ __forceinline__ float kms_round_to(float v, const km_bf16*) { return __uint_as_float(((uint32_t)km_f32_to_bf16_bits(v)) << 16); }
__device__ __forceinline__ float kms_round_to(float v, const km_f16*) { KM_OPAQUE(v); return (float)(km_f16)v; }

template <typename T>
__global__ __launch_bounds__(256) void km_filter_sep_big_fwd_kernel(const KmSepBigArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int kW = a.kW, kH = a.kH;
    const int IW = KMS_TW + kW - 1, IH = KMS_TH + kH - 1, IP = a.iw_pitch;
    float* s_in = (float*)smem_raw;       // [IH][IP]
    float* s_tmp = s_in + (size_t)IH * IP;  // [IH][KMS_TW]
    __shared__ __attribute__((aligned(16))) float s_kx[72], s_ky[72];  // taps, zero-padded to a multiple of 8 (k <= 65)

    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t ty = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int b = (int)(bc / a.C);
    const int x0 = (int)tx * KMS_TW, y0 = (int)ty * KMS_TH;
    const T* img = a.x + (size_t)bc * a.H * a.W;
    T* out = a.y + (size_t)bc * a.Ho * a.Wo;
    const int tid = threadIdx.x;
    const float* kxp = a.kx + (size_t)(b % a.Bk) * kW;
    const float* kyp = a.ky + (size_t)(b % a.Bk) * kH;

    if (tid < 72) s_kx[tid] = tid < kW ? kxp[tid] : 0.0f;
    else if (tid >= 128 && tid < 200) s_ky[tid - 128] = (tid - 128) < kH ? kyp[tid - 128] : 0.0f;
    // ---- stage the tile + halo: half a block per row, the column index map is fixed per thread ----
    {
        const int c = tid & 127, rr = tid >> 7;
        int sx = -1;
        if (c < IW) {
            sx = x0 + c - a.pl;
            if (a.same) sx = km_border_map(sx, a.W, a.border);
            else if (sx >= a.W) sx = -1;
        }
        // 4 rows per trip: the loads are issued together (unconditional, clamped addresses), zeros selected afterwards
        for (int r = rr; r < IH; r += 8) {
            float v[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int sy = y0 + r + 2 * u - a.pt;
                if (a.same) sy = km_border_map(sy, a.H, a.border);
                else if (sy >= a.H) sy = -1;
                ok[u] = (r + 2 * u < IH) && sy >= 0 && sx >= 0;
                v[u] = (float)km_ld(img + (size_t)(ok[u] ? sy : 0) * a.W + (ok[u] ? sx : 0));
            }
            if (c < IP) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (r + 2 * u < IH) s_in[(r + 2 * u) * IP + c] = ok[u] ? v[u] : 0.0f;
            }
        }
    }
    __syncthreads();

    // ---- row pass: tmp[r][c] = sum_q kx[q] * in[r][c + q], 4 adjacent outputs per task ----
    for (int task = tid; task < IH * (KMS_TW / 4); task += 256) {
        const int r = task / (KMS_TW / 4), c0 = (task % (KMS_TW / 4)) * 4;
        const float* row = s_in + r * IP + c0;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int q0 = 0;
        for (; q0 + 8 <= kW; q0 += 8) {
            float w[12];
#pragma unroll
            for (int v4 = 0; v4 < 3; ++v4) {
                KM_CHECK_ALIGNED(row + q0 + 4 * v4, 16);
                const float4 t = *reinterpret_cast<const float4*>(row + q0 + 4 * v4);
                w[4 * v4] = t.x; w[4 * v4 + 1] = t.y; w[4 * v4 + 2] = t.z; w[4 * v4 + 3] = t.w;
            }
            float kq[8];
            {
                KM_CHECK_ALIGNED(s_kx + q0, 16);
                const float4 ka = *reinterpret_cast<const float4*>(s_kx + q0), kb = *reinterpret_cast<const float4*>(s_kx + q0 + 4);
                kq[0] = ka.x; kq[1] = ka.y; kq[2] = ka.z; kq[3] = ka.w; kq[4] = kb.x; kq[5] = kb.y; kq[6] = kb.z; kq[7] = kb.w;
            }
#pragma unroll
            for (int qq = 0; qq < 8; ++qq)
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[o] = km_fma(kq[qq], w[o + qq], acc[o]);
        }
        if (q0 < kW) {  // tail block: fewer than 8 taps (the LDS pitch leaves room for the full 12-value window)
            float w[12];
#pragma unroll
            for (int v4 = 0; v4 < 3; ++v4) {
                KM_CHECK_ALIGNED(row + q0 + 4 * v4, 16);
                const float4 t = *reinterpret_cast<const float4*>(row + q0 + 4 * v4);
                w[4 * v4] = t.x; w[4 * v4 + 1] = t.y; w[4 * v4 + 2] = t.z; w[4 * v4 + 3] = t.w;
            }
#pragma unroll
            for (int qq = 0; qq < 7; ++qq) {
                if (q0 + qq < kW) {
                    const float kq = s_kx[q0 + qq];
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[o] = km_fma(kq, w[o + qq], acc[o]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) s_tmp[r * KMS_TW + c0 + o] = kms_round_to(acc[o], (const T*)nullptr);
    }
    __syncthreads();

    // ---- column pass: out[r][c] = sum_p ky[p] * tmp[r + p][c], 4 consecutive rows per task ----
    {
        const int c = tid & (KMS_TW - 1);
        const int ox = x0 + c;
        for (int rg = tid >> 6; rg < KMS_TH / 4; rg += 4) {
            const int rbase = rg * 4;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            int p0 = 0;
            for (; p0 + 8 <= kH; p0 += 8) {
                float w[11];
#pragma unroll
                for (int i = 0; i < 11; ++i) w[i] = s_tmp[(rbase + p0 + i) * KMS_TW + c];
                float kp[8];
                {
                    KM_CHECK_ALIGNED(s_ky + p0, 16);
                    const float4 ka = *reinterpret_cast<const float4*>(s_ky + p0), kb = *reinterpret_cast<const float4*>(s_ky + p0 + 4);
                    kp[0] = ka.x; kp[1] = ka.y; kp[2] = ka.z; kp[3] = ka.w; kp[4] = kb.x; kp[5] = kb.y; kp[6] = kb.z; kp[7] = kb.w;
                }
#pragma unroll
                for (int pp = 0; pp < 8; ++pp)
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[o] = km_fma(kp[pp], w[o + pp], acc[o]);
            }
            if (p0 < kH) {
                float w[11];
#pragma unroll
                for (int i = 0; i < 11; ++i) {
                    const int rr = rbase + p0 + i;
                    w[i] = rr < IH ? s_tmp[rr * KMS_TW + c] : 0.0f;
                }
#pragma unroll
                for (int pp = 0; pp < 7; ++pp) {
                    if (p0 + pp < kH) {
                        const float kp = s_ky[p0 + pp];
#pragma unroll
                        for (int o = 0; o < 4; ++o) acc[o] = km_fma(kp, w[o + pp], acc[o]);
                    }
                }
            }
            if (ox < a.Wo) {
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int oy = y0 + rbase + o;
                    if (oy < a.Ho) km_st(out + (size_t)oy * a.Wo + ox, acc[o]);
                }
            }
        }
    }
}

static size_t kms_lds_bytes(int kH, int kW, int* pitch) {
    const int IW = KMS_TW + kW - 1, IH = KMS_TH + kH - 1;
    const int IP = ((IW + 8 + 3) / 4) * 4;  // room for the 12-value window of the last output group's tail block
    if (pitch) *pitch = IP;
    return ((size_t)IH * IP + (size_t)IH * KMS_TW) * sizeof(float);
}

// 1 if this kernel takes the forward (big kernels only: the register-tiled one is better up to 9, the small-tile LDS
// kernel fine up to ~10)
int km_filter_sep_big_supported(int kH, int kW, int dtype) {
    if (dtype == KM_F64) return 0;
    if (kH < 10 && kW < 10) return 0;
    int pitch = 0;
    const size_t lds = kms_lds_bytes(kH, kW, &pitch);
    if (pitch > 128) return 0;  // staging covers 128 columns per row
    return lds <= KMS_LDS_LIMIT ? 1 : 0;
}

template <typename T>
static int kms_run(const void* x, const void* kx, const void* ky, void* y, int B, int C, int H, int W, int Bk, int kH, int kW, int border,
                   int same, hipStream_t s) {
    KmSepBigArgs<T> a;
    a.x = (const T*)x; a.y = (T*)y; a.kx = (const float*)kx; a.ky = (const float*)ky;
    a.C = C; a.H = H; a.W = W; a.Bk = Bk; a.kH = kH; a.kW = kW; a.border = border; a.same = same;
    a.Ho = same ? H : H - kH + 1;
    a.Wo = same ? W : W - kW + 1;
    a.pt = same ? (kH - 1) / 2 : 0;
    a.pl = same ? (kW - 1) / 2 : 0;
    const size_t lds = kms_lds_bytes(kH, kW, &a.iw_pitch);
    a.tiles_x = (uint32_t)((a.Wo + KMS_TW - 1) / KMS_TW);
    a.tiles_y = (uint32_t)((a.Ho + KMS_TH - 1) / KMS_TH);
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B * C;
    KM_REQUIRE(nb < (1ull << 31), "km_filter2d_sep: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    hipLaunchKernelGGL(km_filter_sep_big_fwd_kernel<T>, dim3(a.nblocks), dim3(256), lds, s, a);
    return km_check_launch("km_filter2d_sep_fwd(big)");
}

int km_filter_sep_big_run(const void* x, const void* kx, const void* ky, void* y, int B, int C, int H, int W, int Bk, int kH, int kW,
                          int border, int same, int dtype, hipStream_t s) {
    switch (dtype) {
        case KM_F32: return kms_run<float>(x, kx, ky, y, B, C, H, W, Bk, kH, kW, border, same, s);
        case KM_BF16: return kms_run<km_bf16>(x, kx, ky, y, B, C, H, W, Bk, kH, kW, border, same, s);
        default: return kms_run<km_f16>(x, kx, ky, y, B, C, H, W, Bk, kH, kW, border, same, s);
    }
}
