// TEST INFRASTRUCTURE ONLY - never shipped, never imported by kornia_amd/.
//
// A host stand-in for <hip/hip_runtime.h>: with this directory first on the include path, the UNMODIFIED kernel sources
// under kornia_amd/csrc/*.hip compile for x86 (ROCm's clang++, -ffp-contract=off like the device build) into
// tests/_build/libkornia_amd_emu.so, which exports the same C ABI (include/kornia_amd.h).  A launch runs every workgroup
// in turn; the work-items of a group are cooperative fibers on one OS thread (emu_runtime.cpp), so __syncthreads, LDS,
// wave shuffles / ballot / readfirstlane and atomics behave like the hardware's (and a barrier or wave operation that
// not all live lanes reach is reported instead of hanging).  It exists so that the CPU test tier can check the arithmetic
// of the shipped kernels bit for bit against the oracle without a GPU; it says nothing about speed and it is not a
// fallback: the Python package refuses host tensors and never loads this library.
#pragma once

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

using std::max;
using std::min;

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_idx3 {
    unsigned x, y, z;
};
extern emu_idx3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
struct hipDeviceProp_t {
    char gcnArchName[64];
    int multiProcessorCount;
};
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipGetDevice(int* dev);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int dev);
// (streams and events: the host build runs every launch at once on the calling thread; km_side_fork never hands out a side stream there)
typedef void* hipEvent_t;
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 1; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return 1; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 1; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 1; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    memset(p, v, n);
    return hipSuccess;
}

// ---- alignment of vector accesses (kornia_amd/csrc/km_common.h leaves this empty on the device) --------------------
namespace emu { void misaligned(const void* p, int bytes, const char* file, int line); }
#define KM_CHECK_ALIGNED(p, bytes) \
    do { if (((uintptr_t)(p)) % (bytes)) emu::misaligned((p), (bytes), __FILE__, __LINE__); } while (0)

// ---- vector types -------------------------------------------------------------------------------
struct __attribute__((aligned(8))) float2 { float x, y; };
struct __attribute__((aligned(16))) float4 { float x, y, z, w; };
struct __attribute__((aligned(8))) int2 { int x, y; };
struct __attribute__((aligned(16))) int4 { int x, y, z, w; };
struct __attribute__((aligned(8))) uint2 { unsigned x, y; };
struct __attribute__((aligned(16))) uint4 { unsigned x, y, z, w; };
struct __attribute__((aligned(16))) double2 { double x, y; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline int2 make_int2(int x, int y) { return int2{x, y}; }
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

// ---- bit casts / scalar intrinsics --------------------------------------------------------------
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float __frcp_rn(float x) { return 1.0f / x; }
// v_rcp_f32 is accurate to 1 ulp, not correctly rounded: the stand-in returns the rounded reciprocal moved by one ulp
// up, down or not at all depending on the operand's bits, so code that claims an exact result after refining it
// (kml_rcp_refined / kml_div_by, km_lean.h) is tested against every error the instruction is allowed to make
inline float emu_rcpf(float x) {
    float r = 1.0f / x;
    if (r != r || r - r != 0.0f || r == 0.0f) return r;  // NaN, inf, zero
    unsigned u;
    memcpy(&u, &x, 4);
    u = (u ^ (u >> 7) ^ (u >> 13)) * 2654435761u;
    const unsigned k = (u >> 29) % 3u;
    if (k == 1) r = nextafterf(r, 3.0e38f);
    if (k == 2) r = nextafterf(r, -3.0e38f);
    return r;
}
#define __builtin_amdgcn_rcpf(x) emu_rcpf(x)
// v_mul_i32_i24: product of the sign-extended low 24 bits (callers guarantee that both operands fit)
inline int __mul24(int a, int b) { return (int)((unsigned)((a << 8) >> 8) * (unsigned)((b << 8) >> 8)); }
// v_cvt_rpi_i32_f32: floor(v + 0.5) with the conversion's hardware semantics (saturation, NaN -> 0)
inline int emu_cvt_rpi_i32_f32(float v) {
    if (v != v) return 0;
    const float f = floorf(v + 0.5f);
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return -2147483647 - 1;
    return (int)f;
}
// v_cvt_i32_f32: truncation toward zero, saturating, NaN -> 0
inline int emu_cvt_i32_f32(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return -2147483647 - 1;
    return (int)v;
}
#define KM_F2I(v) emu_cvt_i32_f32(v)
#define KM_CVT_RPI(v) emu_cvt_rpi_i32_f32(v)
#define KM_LDS_BARRIER() __syncthreads()
#define KM_SCHED_FENCE() ((void)0)
#define KM_OPAQUE(v) ((void)0)
// LDS-DMA (km_common.h KM_GLDS16 / KM_GLDS4): the lane's piece lands at lds_wave_base + lane * size - on the device at SOME time between
// the request and the wave's `s_waitcnt vmcnt(0)` (KM_VMCNT0).  Two models bracket that (emu_set_glds / KM_EMU_GLDS):
//   immediate (default)  the copy happens at the request: the earliest the hardware could land it - a request into a buffer that another
//                        wave is still reading shows as a wrong result;
//   deferred             the copy happens at the requesting lane's KM_VMCNT0(), in shuffled order, from the source as it is THEN: the latest
//                        the hardware may land it - a missing wait (or a missing barrier behind it) leaves stale LDS and shows as a wrong result;
//                        a request that is never waited for never lands.
// (What neither model checks: that the device's inline assembly is well-formed and that M0-relative LDS addresses above 64 KB work - the -m gpu run.)
namespace emu { void glds_issue(void* dst, const void* src, int bytes); void glds_wait(); }
#define KM_GLDS16(gsrc, lds_wave_base) emu::glds_issue((char*)(lds_wave_base) + 16 * emu::lane_id(), (const void*)(gsrc), 16)
#define KM_GLDS4(gsrc, lds_wave_base) emu::glds_issue((char*)(lds_wave_base) + 4 * emu::lane_id(), (const void*)(gsrc), 4)
#define KM_VMCNT0() emu::glds_wait()
#define KM_TID_PINNED 1
inline int km_tid_pinned() { return (int)threadIdx.x; }
inline float emu_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
#define __builtin_amdgcn_fmed3f(a, b, c) emu_fmed3f((a), (b), (c))

// ---- atomics (one OS thread: plain read-modify-write) ----------------------------------------------
template <typename T> inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }
// integer atomics wrap around (the hardware adds modulo 2^32; signed overflow would be undefined behaviour on the host)
inline int atomicAdd(int* p, int v) { const int o = *p; *p = (int)((unsigned)o + (unsigned)v); return o; }
template <typename T> inline T unsafeAtomicAdd(T* p, T v) { return atomicAdd(p, v); }
template <typename T> inline T atomicMax(T* p, T v) { const T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T* p, T v) { const T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicOr(T* p, T v) { const T o = *p; *p = o | v; return o; }

// ---- barriers and wave-level operations (emu_runtime.cpp) ---------------------------------------------
namespace emu {
void syncthreads();
int syncthreads_or(int pred);
// publish an 8-byte value for this lane, wait until every live lane of the wave did, then read lane `src`'s value
// (src < 0: the first live lane).  ok = false when that lane has exited or does not exist.
uint64_t wave_exchange(uint64_t mine, int src, bool* ok);
uint64_t wave_ballot(bool pred);
int lane_id();
char* dyn_smem();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
}  // namespace emu

inline void __syncthreads() { emu::syncthreads(); }
template <typename T> inline T __shfl_down(T v, unsigned off, int width = 64) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    const int lane = emu::lane_id(), src = lane + (int)off;
    bool ok = false;
    const uint64_t got = emu::wave_exchange(bits, ((lane % width) + (int)off < width && src < 64) ? src : lane, &ok);
    T r;
    memcpy(&r, &got, sizeof(T));
    return ok ? r : v;
}
// ds_bpermute-style read of lane `src` (0 .. 63): every live lane must reach it; a lane that has exited gives the caller's own value
template <typename T> inline T __shfl(T v, int src, int width = 64) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    bool ok = false;
    const uint64_t got = emu::wave_exchange(bits, src & (width - 1), &ok);
    T r;
    memcpy(&r, &got, sizeof(T));
    return ok ? r : v;
}
// DPP row_shl:1 (km_common.h km_next16): lane i of a row of 16 reads lane i + 1, the last lane of the row its own value
#define KM_NEXT16 1
inline uint32_t km_next16(uint32_t v) { return __shfl_down(v, 1, 16); }
inline float km_next16(float v) { return __shfl_down(v, 1, 16); }
inline float km_prev16(float v) {  // row_shr:1
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(float));
    const int lane = emu::lane_id();
    bool ok = false;
    const uint64_t got = emu::wave_exchange(bits, (lane & 15) ? lane - 1 : lane, &ok);
    float r;
    memcpy(&r, &got, sizeof(float));
    return ok ? r : v;
}
inline float km_quad_xor1(float v) { return __shfl(v, emu::lane_id() ^ 1, 64); }  // quad_perm [1,0,3,2]
inline float km_quad_xor2(float v) { return __shfl(v, emu::lane_id() ^ 2, 64); }  // quad_perm [2,3,0,1]
inline uint32_t km_next64(uint32_t v) { return __shfl_down(v, 1, 64); }  // wave_shl:1
inline float km_next64(float v) { return __shfl_down(v, 1, 64); }
inline float km_prev64(float v) {  // wave_shr:1
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(float));
    const int lane = emu::lane_id();
    bool ok = false;
    const uint64_t got = emu::wave_exchange(bits, lane ? lane - 1 : lane, &ok);
    float r;
    memcpy(&r, &got, sizeof(float));
    return ok ? r : v;
}
inline uint32_t km_wave_umax_last(uint32_t v) {  // (km_common.h: valid in lane 63; here every lane gets the maximum)
    for (unsigned off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_down(v, off, 64); v = o > v ? o : v; }
    return __shfl(v, 0, 64);
}
inline double km_wave_sum_last(double v) {  // (km_common.h: valid in lane 63; here every lane gets the sum)
    for (unsigned off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return __shfl(v, 0, 64);
}
inline int emu_readfirstlane(int v) {
    bool ok = false;
    return (int)(uint32_t)emu::wave_exchange((uint64_t)(uint32_t)v, -1, &ok);
}
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane(v)
#define __builtin_amdgcn_s_setprio(p) ((void)0)  // (issue priority of a wave: no meaning for results)
inline unsigned long long __ballot(int pred) { return emu::wave_ballot(pred != 0); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __all(int pred) { return emu::wave_ballot(pred == 0) == 0ull; }  // no live lane with a false predicate
inline int __any(int pred) { return emu::wave_ballot(pred != 0) != 0ull; }
inline int __syncthreads_or(int pred) { return emu::syncthreads_or(pred); }
inline int __syncthreads_and(int pred) { return !emu::syncthreads_or(!pred); }

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (size_t)(shmem), [&]() { kern(__VA_ARGS__); })
