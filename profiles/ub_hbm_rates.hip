// scratch micro-benchmark: what a streaming kernel reaches on this box (HBM3E, MI355X) for the access shapes of the hot kernels.
//   hipcc --offload-arch=gfx950 -O3 profiles/ub_hbm_rates.hip -o scratch/ub/hbm_rates && scratch/ub/hbm_rates
// 805 MB tensors (256x3x512x512 fp32), every variant timed over 10 launches with HIP events.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float v4 __attribute__((ext_vector_type(4)));

// mode: 0 plain, 1 nontemporal loads, 2 nontemporal stores, 3 both
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void k_copy(const v4* __restrict__ src, v4* __restrict__ dst, size_t n4) {
    size_t i = ((size_t)blockIdx.x * UNROLL) * 256 + threadIdx.x;
    v4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const size_t k = i + (size_t)u * 256;
        if (k < n4) v[u] = (MODE & 1) ? __builtin_nontemporal_load(src + k) : src[k];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const size_t k = i + (size_t)u * 256;
        if (k < n4) {
            if (MODE & 2) __builtin_nontemporal_store(v[u], dst + k);
            else dst[k] = v[u];
        }
    }
}

template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void k_read(const v4* __restrict__ src, float* __restrict__ out, size_t n4) {
    size_t i = ((size_t)blockIdx.x * UNROLL) * 256 + threadIdx.x;
    float s = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const size_t k = i + (size_t)u * 256;
        if (k < n4) {
            const v4 v = (MODE & 1) ? __builtin_nontemporal_load(src + k) : src[k];
            s += v.x + v.y + v.z + v.w;
        }
    }
    if (s == 123.456f) out[0] = s;
}

template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void k_write(v4* __restrict__ dst, size_t n4, float val) {
    size_t i = ((size_t)blockIdx.x * UNROLL) * 256 + threadIdx.x;
    const v4 v = {val, val, val, val};
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const size_t k = i + (size_t)u * 256;
        if (k < n4) {
            if (MODE & 2) __builtin_nontemporal_store(v, dst + k);
            else dst[k] = v;
        }
    }
}

// two reads + one write (the shape of a 3e backward)
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void k_r2w1(const v4* __restrict__ a, const v4* __restrict__ b, v4* __restrict__ dst, size_t n4) {
    size_t i = ((size_t)blockIdx.x * UNROLL) * 256 + threadIdx.x;
    v4 va[UNROLL], vb[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const size_t k = i + (size_t)u * 256;
        if (k < n4) { va[u] = a[k]; vb[u] = b[k]; }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const size_t k = i + (size_t)u * 256;
        if (k < n4) {
            const v4 r = va[u] + vb[u];
            if (MODE & 2) __builtin_nontemporal_store(r, dst + k);
            else dst[k] = r;
        }
    }
}

// dword-per-lane copy (the store shape of the gather forward)
__global__ __launch_bounds__(256) void k_copy1(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    size_t i = ((size_t)blockIdx.x * 4) * 256 + threadIdx.x;
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const size_t k = i + (size_t)u * 256; if (k < n) v[u] = src[k]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const size_t k = i + (size_t)u * 256; if (k < n) dst[k] = v[u]; }
}

// tile-ordered copy of (planes x 512 x 512): a 256-thread block copies a TW x TH tile of one plane, dword per lane (VEC = 1) or
// 16 bytes per lane (VEC = 4); blocks in tile-row-major order inside a plane, optionally remapped so that an XCD owns a contiguous range
__device__ inline unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned q = nblocks / 8, r = nblocks % 8, xcd = bid % 8, k = bid / 8;
    return xcd * q + (xcd < r ? xcd : r) + k;
}
template <int TW, int TH, int VEC, int REMAP, int PLANES_PER_BLOCK>
__global__ __launch_bounds__(256) void k_copy_tiled(const float* __restrict__ src, float* __restrict__ dst, int S, unsigned nblocks) {
    const unsigned bid = REMAP ? xcd_remap(blockIdx.x, nblocks) : blockIdx.x;
    const unsigned tiles_x = S / TW, tiles_y = S / TH;
    const unsigned tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, pl = bid / (tiles_x * tiles_y);
    constexpr int CPR = TW / VEC;              // threads per tile row
    constexpr int RPP = 256 / CPR;             // rows per pass
    const int c = (threadIdx.x % CPR) * VEC, r0 = threadIdx.x / CPR;
#pragma unroll
    for (int p = 0; p < PLANES_PER_BLOCK; ++p) {
        const size_t base = ((size_t)(pl * PLANES_PER_BLOCK + p) * S + (size_t)ty * TH) * S + (size_t)tx * TW + c;
#pragma unroll
        for (int r = r0; r < TH; r += RPP) {
            if (VEC == 4) *(v4*)(dst + base + (size_t)r * S) = *(const v4*)(src + base + (size_t)r * S);
            else dst[base + (size_t)r * S] = src[base + (size_t)r * S];
        }
    }
}
// forward-like access shapes on a 64x16 tile (3 planes per block, XCD remap):  LD: 0 dword per pixel, 1 two overlapping dwordx2
// (rows r and r+1, columns c..c+1: the bilinear footprint under the identity), 2 one 16-byte load per 4 pixels;  ST4: lane owns 4
// adjacent pixels and stores 16 bytes, else one dword per pixel (lane = column)
struct __attribute__((packed, aligned(4))) f2u { float x, y; };
template <int LD, int ST4>
__global__ __launch_bounds__(256) void k_fwdlike(const float* __restrict__ src, float* __restrict__ dst, int S, unsigned nblocks) {
    const unsigned bid = xcd_remap(blockIdx.x, nblocks);
    const unsigned tiles_x = S / 64, tiles_y = S / 16;
    const unsigned tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, pl = bid / (tiles_x * tiles_y);
    const int Sm = S - 2;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const float* sp = src + (size_t)(pl * 3 + p) * S * S;
        float* dp = dst + (size_t)(pl * 3 + p) * S * S;
        if (ST4) {
            const int c = (threadIdx.x & 15) * 4 + tx * 64, r = (threadIdx.x >> 4) + ty * 16;  // 16 lanes x 4 px per row, 16 rows
            v4 o;
            if (LD == 2) o = *(const v4*)(sp + (size_t)r * S + c);
            else {
                float t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (LD == 0) t[k] = sp[(size_t)r * S + c + k];
                    else {
                        const int cc = min(c + k, Sm), rr = min(r, Sm);
                        const f2u a = *(const f2u*)(sp + (size_t)rr * S + cc), b = *(const f2u*)(sp + (size_t)(rr + 1) * S + cc);
                        t[k] = a.x * 0.25f + a.y * 0.25f + b.x * 0.25f + b.y * 0.25f;
                    }
                }
                o = v4{t[0], t[1], t[2], t[3]};
            }
            *(v4*)(dp + (size_t)r * S + c) = o;
        } else {
            const int c = (threadIdx.x & 63) + tx * 64;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = (threadIdx.x >> 6) + 4 * k + ty * 16;
                float t;
                if (LD == 0) t = sp[(size_t)r * S + c];
                else if (LD == 1) {
                    const int cc = min(c, Sm), rr = min(r, Sm);
                    const f2u a = *(const f2u*)(sp + (size_t)rr * S + cc), b = *(const f2u*)(sp + (size_t)(rr + 1) * S + cc);
                    t = a.x * 0.25f + a.y * 0.25f + b.x * 0.25f + b.y * 0.25f;
                } else if (LD == 3) {  // four dword loads
                    const int cc = min(c, Sm), rr = min(r, Sm);
                    const float a0 = sp[(size_t)rr * S + cc], a1 = sp[(size_t)rr * S + cc + 1], b0 = sp[(size_t)(rr + 1) * S + cc], b1 = sp[(size_t)(rr + 1) * S + cc + 1];
                    t = a0 * 0.25f + a1 * 0.25f + b0 * 0.25f + b1 * 0.25f;
                } else if (LD == 4) {  // two ALIGNED 8-byte loads (even columns: not the footprint, the cost of the instruction)
                    const int cc = min(c, Sm) & ~1, rr = min(r, Sm);
                    const float2 a = *(const float2*)(sp + (size_t)rr * S + cc), b = *(const float2*)(sp + (size_t)(rr + 1) * S + cc);
                    t = a.x * 0.25f + a.y * 0.25f + b.x * 0.25f + b.y * 0.25f;
                } else if (LD == 5) {  // two dword loads (rows r, r+1), the x neighbour from the next lane
                    const int cc = min(c, Sm), rr = min(r, Sm);
                    const float a0 = sp[(size_t)rr * S + cc], b0 = sp[(size_t)(rr + 1) * S + cc];
                    const float a1 = __shfl_down(a0, 1, 64), b1 = __shfl_down(b0, 1, 64);
                    t = a0 * 0.25f + a1 * 0.25f + b0 * 0.25f + b1 * 0.25f;
                } else t = sp[(size_t)r * S + c];
                dp[(size_t)r * S + c] = t;
            }
        }
    }
}
template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

#define RUN(name, bytes, ...)                                                                     \
    do {                                                                                          \
        const float ms = timeit([&]() { __VA_ARGS__; });                                          \
        printf("%-52s %7.3f ms  %7.1f GB/s\n", name, ms, (double)(bytes) / ms / 1e6); fflush(stdout);            \
    } while (0)

#define TILED(TW, TH, VEC, REMAP, PPB)                                                                                        \
    do {                                                                                                                      \
        const unsigned nb = (unsigned)((512 / TW) * (512 / TH) * (768 / PPB));                                                \
        RUN("tiled copy " #TW "x" #TH " vec" #VEC " remap" #REMAP " planes/block " #PPB, 2 * bytes,                          \
            hipLaunchKernelGGL((k_copy_tiled<TW, TH, VEC, REMAP, PPB>), dim3(nb), dim3(256), 0, 0, a, c, 512, nb));           \
    } while (0)

#define FWDLIKE(LD, ST4)                                                                                                  \
    do {                                                                                                                  \
        const unsigned nb = (unsigned)(8 * 32 * 256);                                                                     \
        RUN("fwd-like 64x16: loads " #LD " (0 dword,1 2xdwordx2 footprint,2 16B), st4 " #ST4, 2 * bytes,                \
            hipLaunchKernelGGL((k_fwdlike<LD, ST4>), dim3(nb), dim3(256), 0, 0, a, c, 512, nb));                          \
    } while (0)


int main() {
    const size_t n = (size_t)256 * 3 * 512 * 512, n4 = n / 4, bytes = n * 4;
    float *a, *b, *c, *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&o, 64);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes); hipMemset(c, 0, bytes);
    const v4* a4 = (const v4*)a; const v4* b4 = (const v4*)b; v4* c4 = (v4*)c;
#define GRID(U) dim3((unsigned)((n4 + 256 * (U) - 1) / (256 * (U))))
    RUN("copy x4, unroll 1", 2 * bytes, hipLaunchKernelGGL((k_copy<0, 1>), GRID(1), dim3(256), 0, 0, a4, c4, n4));
    RUN("copy x4, unroll 2", 2 * bytes, hipLaunchKernelGGL((k_copy<0, 2>), GRID(2), dim3(256), 0, 0, a4, c4, n4));
    RUN("copy x4, unroll 4", 2 * bytes, hipLaunchKernelGGL((k_copy<0, 4>), GRID(4), dim3(256), 0, 0, a4, c4, n4));
    RUN("copy x4, unroll 8", 2 * bytes, hipLaunchKernelGGL((k_copy<0, 8>), GRID(8), dim3(256), 0, 0, a4, c4, n4));
    RUN("copy x4, unroll 4, nt loads", 2 * bytes, hipLaunchKernelGGL((k_copy<1, 4>), GRID(4), dim3(256), 0, 0, a4, c4, n4));
    RUN("copy x4, unroll 4, nt stores", 2 * bytes, hipLaunchKernelGGL((k_copy<2, 4>), GRID(4), dim3(256), 0, 0, a4, c4, n4));
    RUN("copy x4, unroll 4, nt both", 2 * bytes, hipLaunchKernelGGL((k_copy<3, 4>), GRID(4), dim3(256), 0, 0, a4, c4, n4));
    RUN("copy dword per lane, unroll 4", 2 * bytes, hipLaunchKernelGGL(k_copy1, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, a, c, n));
    RUN("read x4, unroll 4", bytes, hipLaunchKernelGGL((k_read<0, 4>), GRID(4), dim3(256), 0, 0, a4, o, n4));
    RUN("read x4, unroll 8", bytes, hipLaunchKernelGGL((k_read<0, 8>), GRID(8), dim3(256), 0, 0, a4, o, n4));
    RUN("read x4, unroll 4, nt", bytes, hipLaunchKernelGGL((k_read<1, 4>), GRID(4), dim3(256), 0, 0, a4, o, n4));
    RUN("write x4, unroll 4", bytes, hipLaunchKernelGGL((k_write<0, 4>), GRID(4), dim3(256), 0, 0, c4, n4, 1.0f));
    RUN("write x4, unroll 4, nt", bytes, hipLaunchKernelGGL((k_write<2, 4>), GRID(4), dim3(256), 0, 0, c4, n4, 1.0f));
    RUN("2 reads + 1 write x4, unroll 2", 3 * bytes, hipLaunchKernelGGL((k_r2w1<0, 2>), GRID(2), dim3(256), 0, 0, a4, b4, c4, n4));
    RUN("2 reads + 1 write x4, unroll 2, nt stores", 3 * bytes, hipLaunchKernelGGL((k_r2w1<2, 2>), GRID(2), dim3(256), 0, 0, a4, b4, c4, n4));
    TILED(64, 16, 1, 0, 1); TILED(64, 16, 1, 1, 1); TILED(64, 16, 1, 1, 3);
    TILED(64, 64, 1, 1, 3); TILED(32, 32, 1, 1, 3);
    TILED(128, 8, 1, 1, 3); TILED(256, 4, 1, 1, 3); TILED(256, 16, 1, 1, 3);
    TILED(64, 16, 4, 1, 3); TILED(256, 4, 4, 1, 3); TILED(256, 32, 4, 1, 1); TILED(256, 32, 4, 0, 1); TILED(512, 8, 4, 1, 3); TILED(512, 16, 4, 1, 1);
    FWDLIKE(0, 0); FWDLIKE(1, 0); FWDLIKE(3, 0); FWDLIKE(4, 0); FWDLIKE(5, 0);
    RUN("hipMemcpyAsync D2D", 2 * bytes, hipMemcpyAsync(c, a, bytes, hipMemcpyDeviceToDevice, 0));
    RUN("hipMemsetAsync", bytes, hipMemsetAsync(c, 0, bytes, 0));
    return 0;
}
