// kornia_amd - helpers shared by the register-tiled kernels (a lane owns 4 adjacent pixels of a row): 16-byte row
// loads / stores for every storage dtype, and the border index map of F.pad (kornia/filters/filter.py:139).
#pragma once

#include "km_common.h"

// border codes of the C ABI (km_filter2d_fwd & co.)
enum { KM_BORDER_CONSTANT = 0, KM_BORDER_REFLECT = 1, KM_BORDER_REPLICATE = 2, KM_BORDER_CIRCULAR = 3 };

// source index of the unpadded coordinate s (may be < 0 or >= n), or -1 where the padding is the constant zero
__device__ __forceinline__ int km_border_map(int s, int n, int border) {
    if (s >= 0 && s < n) return s;
    switch (border) {
        case KM_BORDER_REFLECT:
            if (s < 0) s = -s;
            if (s >= n) s = 2 * (n - 1) - s;
            return (s >= 0 && s < n) ? s : -1;
        case KM_BORDER_REPLICATE: return s < 0 ? 0 : n - 1;
        case KM_BORDER_CIRCULAR: { int r = s % n; return r < 0 ? r + n : r; }
        default: return -1;
    }
}

// four adjacent pixels with one 16-byte (fp32) / 8-byte (bf16, f16) access; p must be aligned to that size
__device__ __forceinline__ void km_ld4(const float* p, float (&o)[4]) {
    KM_CHECK_ALIGNED(p, 16);
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void km_ld4(const km_bf16* p, float (&o)[4]) {
    KM_CHECK_ALIGNED(p, 8);
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
__device__ __forceinline__ void km_ld4(const km_f16* p, float (&o)[4]) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    KM_CHECK_ALIGNED(p, 8);
    const h4 v = *reinterpret_cast<const h4*>(p);
    o[0] = (float)v.x; o[1] = (float)v.y; o[2] = (float)v.z; o[3] = (float)v.w;
}
__device__ __forceinline__ void km_st4(float* p, const float (&o)[4]) {
    KM_CHECK_ALIGNED(p, 16);
#ifdef KM_NT_ST
    typedef float km_f4v __attribute__((ext_vector_type(4)));
    km_f4v v; v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
    __builtin_nontemporal_store(v, reinterpret_cast<km_f4v*>(p));
#else
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
#endif
}
__device__ __forceinline__ void km_st4(km_bf16* p, const float (&o)[4]) {
    uint2 v;
    v.x = km_f32x2_to_bf16x2_bits(o[0], o[1]);
    v.y = km_f32x2_to_bf16x2_bits(o[2], o[3]);
    KM_CHECK_ALIGNED(p, 8);
#ifdef KM_NT_ST
    typedef uint32_t km_u2v __attribute__((ext_vector_type(2)));
    km_u2v vv; vv.x = v.x; vv.y = v.y;
    __builtin_nontemporal_store(vv, reinterpret_cast<km_u2v*>(p));
#else
    *reinterpret_cast<uint2*>(p) = v;
#endif
}
// fp32 result -> f16, never fused with the arithmetic that produced it (km_common.h, km_st(km_f16*, float))
__device__ __forceinline__ _Float16 km_f16_of(float v) { KM_OPAQUE(v); return (_Float16)v; }
__device__ __forceinline__ void km_st4(km_f16* p, const float (&o)[4]) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 v;
    v.x = km_f16_of(o[0]); v.y = km_f16_of(o[1]); v.z = km_f16_of(o[2]); v.w = km_f16_of(o[3]);
    KM_CHECK_ALIGNED(p, 8);
#ifdef KM_NT_ST
    __builtin_nontemporal_store(v, reinterpret_cast<h4*>(p));
#else
    *reinterpret_cast<h4*>(p) = v;
#endif
}
// km_st4 with the store policy chosen per launch (km_stream_stores): streaming or plain
__device__ __forceinline__ void km_st4_pol(float* p, const float (&o)[4], bool stream) {
    if (stream) { km_st4(p, o); return; }
    KM_CHECK_ALIGNED(p, 16);
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ void km_st4_pol(km_bf16* p, const float (&o)[4], bool stream) {
    if (stream) { km_st4(p, o); return; }
    uint2 v;
    v.x = km_f32x2_to_bf16x2_bits(o[0], o[1]);
    v.y = km_f32x2_to_bf16x2_bits(o[2], o[3]);
    KM_CHECK_ALIGNED(p, 8);
    *reinterpret_cast<uint2*>(p) = v;
}
__device__ __forceinline__ void km_st4_pol(km_f16* p, const float (&o)[4], bool stream) {
    if (stream) { km_st4(p, o); return; }
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 v;
    v.x = km_f16_of(o[0]); v.y = km_f16_of(o[1]); v.z = km_f16_of(o[2]); v.w = km_f16_of(o[3]);
    KM_CHECK_ALIGNED(p, 8);
    *reinterpret_cast<h4*>(p) = v;
}
// two adjacent pixels with one 8-byte / 4-byte store
__device__ __forceinline__ void km_st2(float* p, float a, float b) {
    KM_CHECK_ALIGNED(p, 8);
#ifdef KM_NT_ST
    typedef float km_f2v __attribute__((ext_vector_type(2)));
    km_f2v v; v.x = a; v.y = b;
    __builtin_nontemporal_store(v, reinterpret_cast<km_f2v*>(p));
#else
    *reinterpret_cast<float2*>(p) = make_float2(a, b);
#endif
}
__device__ __forceinline__ void km_st2(km_bf16* p, float a, float b) {
    KM_CHECK_ALIGNED(p, 4);
#ifdef KM_NT_ST
    __builtin_nontemporal_store(km_f32x2_to_bf16x2_bits(a, b), reinterpret_cast<uint32_t*>(p));
#else
    *reinterpret_cast<uint32_t*>(p) = km_f32x2_to_bf16x2_bits(a, b);
#endif
}
__device__ __forceinline__ void km_st2(km_f16* p, float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v;
    v.x = km_f16_of(a); v.y = km_f16_of(b);
    KM_CHECK_ALIGNED(p, 4);
#ifdef KM_NT_ST
    __builtin_nontemporal_store(v, reinterpret_cast<h2*>(p));
#else
    *reinterpret_cast<h2*>(p) = v;
#endif
}
