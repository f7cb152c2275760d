// kornia_amd - register-tiled full kH x kW correlation for small square odd kernels ('same' padding):
// the filter2d shapes behind box_blur / laplacian / unsharp / non-separable gaussian_blur2d
// (reference: kornia/filters/filter.py:121-150 - F.pad + grouped F.conv2d).
//
// Same organisation as the separable blur (km_blur_fast.hip) and spatial_gradient (km_gradient.hip): a lane owns
// 4 adjacent columns (one 16-byte load per input row + PD scalar halo loads per side, one 16-byte store per
// output row), a wave walks a strip of rows keeping the last K input rows in registers.  Border = index map
// (rows: wave-uniform address select; columns: per-lane halo indices computed once).  The fma chain runs in
// (p, q) order from 0 over the whole kernel - bit-identical to oracle/ko_impl.h ko_filter2d_fwd.
// HBM traffic = read x once + write y once = 2e bytes / element.
#include "km_regtile.h"

#define KMF_ROWS 32



template <typename T>
struct KmF2Args {
    const T* x;
    T* y;
    const float* k;  // (Bk, K, K)
    int C, H, W, Bk, border;
    int flip;        // 1: use the taps rotated by 180 degrees (the adjoint's interior kernel)
    uint32_t tiles_x, tiles_y, nblocks;
};

template <typename T, int K>
__global__ __launch_bounds__(256) void km_filter2d_reg_kernel(const KmF2Args<T> a) {
    constexpr int PD = (K - 1) / 2, NV = 4 + 2 * PD;
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tbx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t tby = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx = (int)tbx * 64 + lane;                  // column group (4 px)
    const int r0 = ((int)tby * 4 + wave) * KMF_ROWS;      // first output row of this wave's strip
    const int H = a.H, W = a.W, border = a.border;
    if (gx * 4 >= W || r0 >= H) return;
    const int c0 = gx * 4;
    const size_t plane = (size_t)H * W;
    const T* img = a.x + (size_t)bc * plane;
    T* out = a.y + (size_t)bc * plane;
    const int b = (int)(bc / a.C);

    // halo columns through the border map (only the edge lanes of a row see anything but c0 - PD + q)
    int hl[PD], hr[PD];
    bool okl[PD], okr[PD];
#pragma unroll
    for (int q = 0; q < PD; ++q) {
        const int il = km_border_map(c0 - PD + q, W, border), ir = km_border_map(c0 + 4 + q, W, border);
        okl[q] = il >= 0; hl[q] = okl[q] ? il : 0;
        okr[q] = ir >= 0; hr[q] = okr[q] ? ir : 0;
    }
    float k[K][K];
    {
        const float* kp = a.k + (size_t)(b % a.Bk) * K * K;
#pragma unroll
        for (int p = 0; p < K; ++p)
#pragma unroll
            for (int q = 0; q < K; ++q) k[p][q] = a.flip ? kp[(K - 1 - p) * K + (K - 1 - q)] : kp[p * K + q];
    }

    float ring[K][NV];  // last K input rows: ring[.][i] = column c0 - PD + i
    const int n_rows = (r0 + KMF_ROWS <= H ? KMF_ROWS : H - r0);
    const int total = n_rows + K - 1;
    for (int it0 = 0; it0 < total; it0 += K) {
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int it = it0 + kk;
            if (it < total) {
                const int srow = km_border_map(r0 - PD + it, H, border);  // wave-uniform
                if (srow >= 0) {
                    const T* rowp = img + (size_t)srow * W;
                    float o4[4];
                    km_ld4(rowp + c0, o4);
#pragma unroll
                    for (int q = 0; q < PD; ++q) {
                        const float vl = (float)km_ld(rowp + hl[q]), vr = (float)km_ld(rowp + hr[q]);
                        ring[kk][q] = okl[q] ? vl : 0.f;
                        ring[kk][PD + 4 + q] = okr[q] ? vr : 0.f;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) ring[kk][PD + q] = o4[q];
                } else {
#pragma unroll
                    for (int q = 0; q < NV; ++q) ring[kk][q] = 0.f;
                }
                if (it >= K - 1) {
                    const int r = r0 + it - (K - 1);
                    float acc[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float s = 0.f;
#pragma unroll
                        for (int p = 0; p < K; ++p)
#pragma unroll
                            for (int q = 0; q < K; ++q) s = km_fma(k[p][q], ring[(kk + 1 + p) % K][c + q], s);
                        acc[c] = s;
                    }
                    km_st4(out + (size_t)r * W + c0, acc);
                }
            }
        }
    }
}

template <typename T>
static int kmf_run(const void* x, const void* k, void* y, int B, int C, int H, int W, int Bk, int K, int border, int flip, hipStream_t s) {
    KmF2Args<T> a;
    a.x = (const T*)x; a.y = (T*)y; a.k = (const float*)k;
    a.C = C; a.H = H; a.W = W; a.Bk = Bk; a.border = border; a.flip = flip;
    a.tiles_x = (uint32_t)((W / 4 + 63) / 64);
    a.tiles_y = (uint32_t)((H + 4 * KMF_ROWS - 1) / (4 * KMF_ROWS));
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B * C;
    KM_REQUIRE(nb < (1ull << 31), "km_filter2d: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    switch (K) {
        case 3: hipLaunchKernelGGL((km_filter2d_reg_kernel<T, 3>), dim3(a.nblocks), dim3(256), 0, s, a); break;
        case 5: hipLaunchKernelGGL((km_filter2d_reg_kernel<T, 5>), dim3(a.nblocks), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((km_filter2d_reg_kernel<T, 7>), dim3(a.nblocks), dim3(256), 0, s, a); break;
    }
    return km_check_launch("km_filter2d_fwd(reg)");
}


// ------------------------------------------------------------------------------------------------
// Gradient wrt the taps:  gk[b % Bk][p][q] += sum_{c,i,j} gy[b,c,i,j] * xpad[b,c,i+p,j+q]   (filter.py autograd of conv2d
// wrt weight).  Same register tiling as the forward: the K-row window of the (border-mapped) input stays in registers,
// every lane accumulates its K*K partial sums over a 32-row x 4-column strip in fp32, then one fp64 wave / block reduction
// and K*K fp64 atomics per block.  One pass over x and gy (2e bytes / element) instead of K*K passes.
template <typename T>
struct KmF2GArgs {
    const T* x;
    const T* gy;
    double* gk;  // (Bk, K, K) fp64 accumulators, pre-zeroed
    int C, H, W, Bk, border;
    uint32_t tiles_x, tiles_y, nblocks;
};

template <typename T, int K>
__global__ __launch_bounds__(256) void km_filter2d_tapgrad_reg_kernel(const KmF2GArgs<T> a) {
    constexpr int PD = (K - 1) / 2, NV = 4 + 2 * PD;
    __shared__ double red[4][K * K];
    uint32_t bid = km_xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tbx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const uint32_t tby = bid % a.tiles_y;
    const uint32_t bc = bid / a.tiles_y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx = (int)tbx * 64 + lane;
    const int r0 = ((int)tby * 4 + wave) * KMF_ROWS;
    const int H = a.H, W = a.W, border = a.border;
    const bool active = (gx * 4 < W) && (r0 < H);
    const int c0 = active ? gx * 4 : 0;
    const size_t plane = (size_t)H * W;
    const T* img = a.x + (size_t)bc * plane;
    const T* gyp = a.gy + (size_t)bc * plane;
    const int b = (int)(bc / a.C);

    float acc[K][K];
#pragma unroll
    for (int p = 0; p < K; ++p)
#pragma unroll
        for (int q = 0; q < K; ++q) acc[p][q] = 0.f;

    if (active) {
        int hl[PD], hr[PD];
        bool okl[PD], okr[PD];
#pragma unroll
        for (int q = 0; q < PD; ++q) {
            const int il = km_border_map(c0 - PD + q, W, border), ir = km_border_map(c0 + 4 + q, W, border);
            okl[q] = il >= 0; hl[q] = okl[q] ? il : 0;
            okr[q] = ir >= 0; hr[q] = okr[q] ? ir : 0;
        }
        float ring[K][NV];
        const int n_rows = (r0 + KMF_ROWS <= H ? KMF_ROWS : H - r0);
        const int total = n_rows + K - 1;
        for (int it0 = 0; it0 < total; it0 += K) {
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const int it = it0 + kk;
                if (it < total) {
                    const int srow = km_border_map(r0 - PD + it, H, border);
                    if (srow >= 0) {
                        const T* rowp = img + (size_t)srow * W;
                        float o4[4];
                        km_ld4(rowp + c0, o4);
#pragma unroll
                        for (int q = 0; q < PD; ++q) {
                            const float vl = (float)km_ld(rowp + hl[q]), vr = (float)km_ld(rowp + hr[q]);
                            ring[kk][q] = okl[q] ? vl : 0.f;
                            ring[kk][PD + 4 + q] = okr[q] ? vr : 0.f;
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) ring[kk][PD + q] = o4[q];
                    } else {
#pragma unroll
                        for (int q = 0; q < NV; ++q) ring[kk][q] = 0.f;
                    }
                    if (it >= K - 1) {
                        const int r = r0 + it - (K - 1);
                        float g4[4];
                        km_ld4(gyp + (size_t)r * W + c0, g4);
#pragma unroll
                        for (int p = 0; p < K; ++p)
#pragma unroll
                            for (int q = 0; q < K; ++q)
#pragma unroll
                                for (int c = 0; c < 4; ++c) acc[p][q] = km_fma(g4[c], ring[(kk + 1 + p) % K][c + q], acc[p][q]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < K; ++p)
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const double sm = km_wave_sum_last((double)acc[p][q]);  // (DPP ladder: valid in lane 63; every thread of the block is here)
            if (lane == 63) red[wave][p * K + q] = sm;
        }
    __syncthreads();
    if (threadIdx.x < K * K) {
        const double sm = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (sm != 0.0) km_atomic_add(a.gk + (size_t)(b % a.Bk) * K * K + threadIdx.x, sm);
    }
}

template <typename T>
static int kmf_tapgrad_run(const void* gy, const void* x, double* gk, int B, int C, int H, int W, int Bk, int K, int border, hipStream_t s) {
    KmF2GArgs<T> a;
    a.x = (const T*)x; a.gy = (const T*)gy; a.gk = gk;
    a.C = C; a.H = H; a.W = W; a.Bk = Bk; a.border = border;
    a.tiles_x = (uint32_t)((W / 4 + 63) / 64);
    a.tiles_y = (uint32_t)((H + 4 * KMF_ROWS - 1) / (4 * KMF_ROWS));
    const uint64_t nb = (uint64_t)a.tiles_x * a.tiles_y * (uint64_t)B * C;
    KM_REQUIRE(nb < (1ull << 31), "km_filter2d_bwd_kernel: grid too large");
    a.nblocks = (uint32_t)nb;
    if (nb == 0) return 0;
    switch (K) {
        case 3: hipLaunchKernelGGL((km_filter2d_tapgrad_reg_kernel<T, 3>), dim3(a.nblocks), dim3(256), 0, s, a); break;
        case 5: hipLaunchKernelGGL((km_filter2d_tapgrad_reg_kernel<T, 5>), dim3(a.nblocks), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((km_filter2d_tapgrad_reg_kernel<T, 7>), dim3(a.nblocks), dim3(256), 0, s, a); break;
    }
    return km_check_launch("km_filter2d_bwd_kernel(reg)");
}

int km_filter2d_fast_tapgrad_run(const void* gy, const void* x, double* gk, int B, int C, int H, int W, int Bk, int K, int border, int dtype,
                                 hipStream_t s) {
    switch (dtype) {
        case KM_F32: return kmf_tapgrad_run<float>(gy, x, gk, B, C, H, W, Bk, K, border, s);
        case KM_BF16: return kmf_tapgrad_run<km_bf16>(gy, x, gk, B, C, H, W, Bk, K, border, s);
        default: return kmf_tapgrad_run<km_f16>(gy, x, gk, B, C, H, W, Bk, K, border, s);
    }
}

// 1 if the register-tiled kernel handles this problem ('same', square odd 3/5/7, W % 4 == 0, 16-byte aligned, not fp64)
int km_filter2d_fast_supported(const void* x, const void* y, int H, int W, int kH, int kW, int border, int same, int dtype) {
    if (!same || kH != kW || !(kH == 3 || kH == 5 || kH == 7)) return 0;
    if (dtype == KM_F64) return 0;
    if ((W & 3) != 0 || W < 8 || H < 1) return 0;
    if (border == KM_BORDER_REFLECT && (kH - 1) / 2 >= (H < W ? H : W)) return 0;
    const size_t esz = (dtype == KM_F32) ? 4 : 2;
    if (((uintptr_t)x % (4 * esz)) != 0 || ((uintptr_t)y % (4 * esz)) != 0) return 0;
    return 1;
}

// flip = 0: y = correlate(x, k) with `border`;  flip = 1: the same with the taps rotated by 180 degrees - with
// border = constant that is the exact adjoint of the constant-padded correlation, and the interior of every other one.
int km_filter2d_fast_run(const void* x, const void* k, void* y, int B, int C, int H, int W, int Bk, int K, int border, int flip, int dtype,
                         hipStream_t s) {
    switch (dtype) {
        case KM_F32: return kmf_run<float>(x, k, y, B, C, H, W, Bk, K, border, flip, s);
        case KM_BF16: return kmf_run<km_bf16>(x, k, y, B, C, H, W, Bk, K, border, flip, s);
        default: return kmf_run<km_f16>(x, k, y, B, C, H, W, Bk, K, border, flip, s);
    }
}
