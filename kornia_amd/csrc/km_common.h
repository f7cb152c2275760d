// kornia_amd - shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
//
// Numerics contract (see DESIGN.md "Parity"): the whole library is compiled with
// -ffp-contract=off, so a multiply-add is fused ONLY where the source says km_fma().  The
// coordinate pipeline reproduces the reference's per-op rounding sequence,
// which is what makes the fp32 results bit-identical to the CPU oracle in oracle/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define KM_ABI_VERSION 3  // include/kornia_amd.h, "Versioning": a library of version N exports the symbol sets of every version <= N

// dtype codes of the C ABI (include/kornia_amd.h)
enum { KM_F32 = 0, KM_F64 = 1, KM_BF16 = 2, KM_F16 = 3 };

// ---- error reporting (no exceptions cross the ABI) ------------------------------------------
void km_set_error(const char* fmt, ...);
int km_check_launch(const char* what);

#define KM_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            km_set_error(__VA_ARGS__); \
            return -1;                 \
        }                              \
    } while (0)

// Vector accesses below assume natural alignment (the launchers check pointers and row pitches before choosing such a path).
// Nothing on the device; the host build of the kernels used by the CPU test tier (tests/emu) turns it into a run-time check.
#ifndef KM_CHECK_ALIGNED
#define KM_CHECK_ALIGNED(p, bytes) ((void)0)
#endif

// float -> int32 conversion of a value that may be anything: v_cvt_i32_f32 saturates and turns NaN into 0.  (In C++ the cast
// is undefined for such values; the host build of the kernels substitutes a function with the instruction's semantics.)
#ifndef KM_F2I
#define KM_F2I(v) ((int)(v))
#endif

// floor(v + 0.5) as an int32 in ONE instruction (v_cvt_rpi_i32_f32; saturating, NaN -> 0): the fixed-point quantisation of the
// tile-owner backward kernels.  (The host build of the kernels supplies a function with the instruction's semantics.)
#ifndef KM_CVT_RPI
__device__ __forceinline__ int km_cvt_rpi(float v) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
#define KM_CVT_RPI(v) km_cvt_rpi(v)
#endif

// Value held by the NEXT lane of the same row of 16 lanes (DPP row_shl:1 - one VALU move, no LDS traffic); the last lane of a
// row gets its own value back.  (The host build of the kernels defines KM_NEXT16 and supplies the same function.)
#ifndef KM_NEXT16
__device__ __forceinline__ uint32_t km_next16(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x101, 0xf, 0xf, false);
}
__device__ __forceinline__ float km_next16(float v) { return __uint_as_float(km_next16(__float_as_uint(v))); }
// ... and by the PREVIOUS lane of the row (row_shr:1); the first lane of a row gets its own value back
__device__ __forceinline__ float km_prev16(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v), 0x111, 0xf, 0xf, false));
}
// The same across the whole wave (gfx9 DPP wave_shl:1 / wave_shr:1, verified on gfx950: lane 63 / lane 0 - and a lane whose
// neighbour has exited - get their own value back)
__device__ __forceinline__ uint32_t km_next64(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ float km_next64(float v) { return __uint_as_float(km_next64(__float_as_uint(v))); }
__device__ __forceinline__ float km_prev64(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v), 0x138, 0xf, 0xf, false));
}
// Value held by lane ^ 1 / lane ^ 2 of the same QUAD of lanes (DPP quad_perm [1,0,3,2] / [2,3,0,1]: one VALU move each)
__device__ __forceinline__ float km_quad_xor1(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v), 0xB1, 0xf, 0xf, false));
}
__device__ __forceinline__ float km_quad_xor2(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v), 0x4E, 0xf, 0xf, false));
}
// Maximum of an unsigned value over the wave, valid in LANE 63 only (other lanes hold partial maxima): an inclusive scan inside each row
// of 16 lanes (row_shr:1/2/4/8; a lane without a source keeps its own value), then lane 15 of a row into the next row (row_bcast:15,
// rows 1 and 3) and lane 31 into rows 2 and 3 (row_bcast:31) - ten VALU instructions, no LDS traffic (a __shfl_down ladder is six
// dependent ds_bpermute round trips).  All 64 lanes must be active.
__device__ __forceinline__ uint32_t km_wave_umax_last(uint32_t v) {
    int x = (int)v;
#define KM_UMAX_STEP(ctrl, rows) x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(x, x, ctrl, rows, 0xf, false));
    KM_UMAX_STEP(0x111, 0xf) KM_UMAX_STEP(0x112, 0xf) KM_UMAX_STEP(0x114, 0xf) KM_UMAX_STEP(0x118, 0xf)
    KM_UMAX_STEP(0x142, 0xa) KM_UMAX_STEP(0x143, 0xc)
#undef KM_UMAX_STEP
    return (uint32_t)x;
}
// Sum of an fp64 value over the wave, valid in LANE 63 only: the same ladder on the two halves of the value (a lane without a source, a row
// outside the step's row mask adds 0.0) - 12 DPP moves and 6 fp64 adds, no LDS traffic, where a __shfl_down ladder is 12 dependent
// ds_bpermute round trips (nine such sums per wave were 18 % of the matrix-gradient kernel: profiles/r06/run13_*).  All 64 lanes must be active.
__device__ __forceinline__ double km_wave_sum_last(double v) {
#define KM_DSUM_STEP(ctrl, rows)                                                                                  \
    {                                                                                                             \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rows, 0xf, false);                \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rows, 0xf, false);                \
        v += __hiloint2double(hi_, lo_);                                                                          \
    }
    KM_DSUM_STEP(0x111, 0xf) KM_DSUM_STEP(0x112, 0xf) KM_DSUM_STEP(0x114, 0xf) KM_DSUM_STEP(0x118, 0xf)
    KM_DSUM_STEP(0x142, 0xa) KM_DSUM_STEP(0x143, 0xc)
#undef KM_DSUM_STEP
    return v;
}
#endif

// 4 x 4 transpose inside a quad of lanes: lane q (= lane & 3) enters with a[r] = element (row r, column q) of a 4 x 4 block and leaves with
// a[k] = element (row q, column k) - its own ROW of the block, i.e. four horizontally adjacent pixels for ONE 16-byte store where the
// lane = column layout of the sampling kernels would issue four 4-byte ones.  Two butterfly stages (lane bit 0 <-> register bit 0, lane bit 1
// <-> register bit 1): 4 quad-permute moves + 12 selects, no LDS.  All four lanes of the quad must be active.
__device__ __forceinline__ void km_quad_transpose4(float (&a)[4], int q) {
    const bool b0 = (q & 1) != 0, b1 = (q & 2) != 0;
#pragma unroll
    for (int p = 0; p < 4; p += 2) {  // register pairs (0,1), (2,3) with lane ^ 1
        const float r = km_quad_xor1(b0 ? a[p] : a[p + 1]);
        a[p] = b0 ? r : a[p];
        a[p + 1] = b0 ? a[p + 1] : r;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {     // register pairs (0,2), (1,3) with lane ^ 2
        const float r = km_quad_xor2(b1 ? a[p] : a[p + 2]);
        a[p] = b1 ? r : a[p];
        a[p + 2] = b1 ? a[p + 2] : r;
    }
}
// four horizontally adjacent fp32 pixels with one 16-byte store (p 16-byte aligned), streaming or plain (km_st_c's policy)
template <bool STREAM>
__device__ __forceinline__ void km_st4_c(float* p, const float (&v)[4]) {
    typedef float km_f4v_ __attribute__((ext_vector_type(4)));
    km_f4v_ vv; vv.x = v[0]; vv.y = v[1]; vv.z = v[2]; vv.w = v[3];
    if constexpr (STREAM) __builtin_nontemporal_store(vv, reinterpret_cast<km_f4v_*>(p));
    else *reinterpret_cast<km_f4v_*>(p) = vv;
}

// ---- storage types ----------------------------------------------------------------------------
struct km_bf16 {
    uint16_t bits;
};
typedef _Float16 km_f16;

template <typename T>
struct KmTraits;
template <>
struct KmTraits<float> {
    typedef float R;  // compute type
    static constexpr int code = KM_F32;
};
template <>
struct KmTraits<double> {
    typedef double R;
    static constexpr int code = KM_F64;
};
template <>
struct KmTraits<km_bf16> {
    typedef float R;
    static constexpr int code = KM_BF16;
};
template <>
struct KmTraits<km_f16> {
    typedef float R;
    static constexpr int code = KM_F16;
};

__device__ __forceinline__ float km_ld(const float* p) { return *p; }
__device__ __forceinline__ double km_ld(const double* p) { return *p; }
__device__ __forceinline__ float km_ld(const km_bf16* p) { return __uint_as_float(((uint32_t)p->bits) << 16); }
__device__ __forceinline__ float km_ld(const km_f16* p) { return (float)(*p); }

// float -> bfloat16, round to nearest even, NaN stays a quiet NaN.  gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two
// values per instruction: what `(__bf16)f` compiles to in the device pass); everywhere else - the host pass, the host build of the
// kernels under tests/emu - the same rounding in integer arithmetic (5 instructions per value).  Measured on the bf16 blur of
// BASELINE config 3 (256x3x224^2): see DESIGN.md.
__device__ __forceinline__ uint16_t km_f32_to_bf16_bits_sw(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                            // round to nearest even
    return (uint16_t)(u >> 16);
}
#if defined(__gfx950__)
__device__ __forceinline__ uint16_t km_f32_to_bf16_bits(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
// (lo, hi) -> one 32-bit word of two bfloat16, lo in the low half
__device__ __forceinline__ uint32_t km_f32x2_to_bf16x2_bits(float lo, float hi) {
    typedef __bf16 km_bf2v __attribute__((ext_vector_type(2)));
    typedef float km_f2v_ __attribute__((ext_vector_type(2)));
    km_f2v_ v; v.x = lo; v.y = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, km_bf2v));
}
#else
__device__ __forceinline__ uint16_t km_f32_to_bf16_bits(float f) { return km_f32_to_bf16_bits_sw(f); }
__device__ __forceinline__ uint32_t km_f32x2_to_bf16x2_bits(float lo, float hi) {
    return (uint32_t)km_f32_to_bf16_bits_sw(lo) | ((uint32_t)km_f32_to_bf16_bits_sw(hi) << 16);
}
#endif
// An opaque copy of a per-lane value: what is computed from it stays where it is written (not hoisted out of a loop, not merged with
// the same computation elsewhere).  No code is emitted (the host build of the kernels defines it away).
#ifndef KM_OPAQUE
#define KM_OPAQUE(v) asm volatile("" : "+v"(v))
#endif
// Outputs are written once and read by the NEXT kernel, after ~0.8 GB of other traffic at the hot sizes: streaming (non-temporal)
// stores keep them from displacing the lines the running kernel still needs in L2.  Measured on MI355X, config 2, same box: step
// 1.798 -> 1.744 ms (forward 0.384 -> 0.370, blur 0.332 -> 0.319, blur adjoint 0.332 -> 0.323, scatter 0.398 -> 0.391 ms).
// Non-temporal LOADS measured slower (blur 0.332 -> 0.35 ms; grad_out in the scatter, which reads it once: step 1.64 -> 1.67 ms).  -DKM_NO_NT_ST builds with plain stores (A/B).
#ifndef KM_NO_NT_ST
#define KM_NT_ST 1
#endif
#ifdef KM_NT_ST
__device__ __forceinline__ void km_st(float* p, float v) { __builtin_nontemporal_store(v, p); }
#else
__device__ __forceinline__ void km_st(float* p, float v) { *p = v; }
#endif
// The three kernels of the hot step choose per launch: an output that fits in the 256 MB Infinity Cache with room to spare is what the next
// kernel will find there, and streaming it out costs that (config 2 at B = 16, 50 MB tensors: step 0.152 ms with plain stores, 0.166 ms with
// streaming ones; B = 64, 201 MB: 0.486 / 0.478; B = 256, 805 MB: 1.74 / 1.69).  stream = km_stream_stores(bytes of the output), wave-uniform.
#define KM_STREAM_MIN_BYTES (128ull << 20)
static inline uint32_t km_stream_stores(uint64_t out_bytes) {
#ifdef KM_NT_ST
    return out_bytes >= KM_STREAM_MIN_BYTES ? 1u : 0u;
#else
    (void)out_bytes;
    return 0u;
#endif
}
// The policy as a COMPILE-TIME parameter.  (Round 2's form - a bool argument, `if (stream) nontemporal_store else plain store` - compiled to
// plain stores in every instantiation: the two stores of the if / else are merged into one before the constant reaches them, and the merge
// drops the non-temporal mark; found in round 3 by reading the ISA of the forward - its hot path had not streamed a byte since.)
#ifndef KM_FWD_PLAIN_ST
#define KM_FWD_PLAIN_ST 0   // 1: the forward's compile-time policy is forced to plain stores (A/B against round 2's accidental behaviour)
#endif
template <bool STREAM>
__device__ __forceinline__ void km_st_c(float* p, float v) {
    if constexpr (STREAM && !KM_FWD_PLAIN_ST) __builtin_nontemporal_store(v, p);
    else *p = v;
}
__device__ __forceinline__ void km_st(double* p, double v) { *p = v; }
__device__ __forceinline__ void km_st(km_bf16* p, float v) { p->bits = km_f32_to_bf16_bits(v); }
// (the fp32 result is made opaque before it is converted: left visible, the compiler fuses the last fma of an accumulation chain with the
// conversion into v_fma_mixlo_f16, which rounds the exact product-sum to f16 ONCE - a result that is not "the fp32 result rounded to the
// storage type" at ties, and differed between kernels that got the fusion and kernels that did not: 1 f16 ulp at ~1e-5 of the pixels)
__device__ __forceinline__ void km_st(km_f16* p, float v) { KM_OPAQUE(v); *p = (km_f16)v; }
template <bool STREAM>
__device__ __forceinline__ void km_st_c(double* p, double v) { *p = v; }
template <bool STREAM>
__device__ __forceinline__ void km_st_c(km_bf16* p, float v) { km_st(p, v); }  // (2-byte stores: no streaming form worth having)
template <bool STREAM>
__device__ __forceinline__ void km_st_c(km_f16* p, float v) { km_st(p, v); }

// a compute-precision value as the storage dtype T would hold it (used where the reference materialises an intermediate)
__device__ __forceinline__ float km_round_as(float v, const float*) { return v; }
__device__ __forceinline__ double km_round_as(double v, const double*) { return v; }
__device__ __forceinline__ float km_round_as(float v, const km_bf16*) { return __uint_as_float(((uint32_t)km_f32_to_bf16_bits(v)) << 16); }
__device__ __forceinline__ float km_round_as(float v, const km_f16*) { KM_OPAQUE(v); return (float)(km_f16)v; }

// two horizontally adjacent pixels with ONE load (8 bytes for fp32 / fp64 pairs, 4 bytes for 16-bit types).
// The address is only element-aligned: gfx950 global loads handle that in hardware; the packed structs
// keep the compiler from assuming more.  Gather loads are TA-bound on this chip (a dword-per-lane wave load
// occupies the texture-address path as long as a 2-dword one), so halving the load count matters.
struct __attribute__((packed, aligned(4))) km_f32x2_u { float x, y; };
struct __attribute__((packed, aligned(8))) km_f64x2_u { double x, y; };
struct __attribute__((packed, aligned(2))) km_u16x2_u { uint16_t x, y; };
__device__ __forceinline__ void km_ld2(const float* p, float& a, float& b) {
    const km_f32x2_u v = *reinterpret_cast<const km_f32x2_u*>(p);
    a = v.x; b = v.y;
}
__device__ __forceinline__ void km_ld2(const double* p, double& a, double& b) {
    const km_f64x2_u v = *reinterpret_cast<const km_f64x2_u*>(p);
    a = v.x; b = v.y;
}
__device__ __forceinline__ void km_ld2(const km_bf16* p, float& a, float& b) {
    const km_u16x2_u v = *reinterpret_cast<const km_u16x2_u*>(p);
    a = __uint_as_float(((uint32_t)v.x) << 16); b = __uint_as_float(((uint32_t)v.y) << 16);
}
struct __attribute__((packed, aligned(2))) km_h16x2_u { km_f16 x, y; };
__device__ __forceinline__ void km_ld2(const km_f16* p, float& a, float& b) {
    const km_h16x2_u v = *reinterpret_cast<const km_h16x2_u*>(p);  // one 4-byte request, like the bf16 pair
    a = (float)v.x; b = (float)v.y;
}

// four horizontally adjacent pixels with one load (element-aligned address), for the bicubic taps
struct __attribute__((packed, aligned(4))) km_f32x4_u { float x, y, z, w; };
struct __attribute__((packed, aligned(8))) km_f64x4_u { double x, y, z, w; };
struct __attribute__((packed, aligned(2))) km_u16x4_u { uint16_t x, y, z, w; };
__device__ __forceinline__ void km_ld4u(const float* p, float (&o)[4]) {
    const km_f32x4_u v = *reinterpret_cast<const km_f32x4_u*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void km_ld4u(const double* p, double (&o)[4]) {
    const km_f64x4_u v = *reinterpret_cast<const km_f64x4_u*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void km_ld4u(const km_bf16* p, float (&o)[4]) {
    const km_u16x4_u v = *reinterpret_cast<const km_u16x4_u*>(p);
    o[0] = __uint_as_float(((uint32_t)v.x) << 16); o[1] = __uint_as_float(((uint32_t)v.y) << 16);
    o[2] = __uint_as_float(((uint32_t)v.z) << 16); o[3] = __uint_as_float(((uint32_t)v.w) << 16);
}
__device__ __forceinline__ void km_ld4u(const km_f16* p, float (&o)[4]) {
    o[0] = (float)p[0]; o[1] = (float)p[1]; o[2] = (float)p[2]; o[3] = (float)p[3];
}

// base + 32-bit element offset with the byte offset kept in 32 bits: with a wave-uniform base the compiler can
// use the  global_load vdst, voffset, s[base:base+1]  form (no 64-bit VALU address arithmetic per lane).
// Callers guarantee  plane elements * sizeof(T) < 2^32.
template <typename T>
__device__ __forceinline__ const T* km_at(const T* base, uint32_t elem_off) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + elem_off * (uint32_t)sizeof(T));
}

template <typename T>
__device__ __forceinline__ T* km_at_mut(T* base, uint32_t elem_off) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + elem_off * (uint32_t)sizeof(T));
}

// ---- explicitly fused / explicitly rounded arithmetic ---------------------------------------
__device__ __host__ __forceinline__ float km_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __host__ __forceinline__ double km_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float km_floor(float x) { return __builtin_floorf(x); }
__device__ __forceinline__ double km_floor(double x) { return __builtin_floor(x); }
__device__ __forceinline__ float km_fabs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double km_fabs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float km_rint(float x) { return __builtin_rintf(x); }  // ties-to-even (nearbyint)
__device__ __forceinline__ double km_rint(double x) { return __builtin_rint(x); }
__device__ __forceinline__ float km_fmod(float a, float b) { return fmodf(a, b); }
__device__ __forceinline__ double km_fmod(double a, double b) { return fmod(a, b); }
__device__ __forceinline__ float km_sqrt(float x) { return __builtin_sqrtf(x); }  // IEEE (built with -fhip-fp32-correctly-rounded-divide-sqrt)
__device__ __forceinline__ double km_sqrt(double x) { return __builtin_sqrt(x); }

// fp32/fp64 global atomic add without return value (hardware global_atomic_add_f32/_f64 on gfx950)
__device__ __forceinline__ void km_atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void km_atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// ---- XCD-aware block index remap ---------------------------------------------------------------
// MI355X dispatches workgroup b to XCD (b % 8); each XCD has a private 4 MiB L2.  Remap the linear
// block id so that each XCD works on one contiguous range of logical tiles (= whole images), which
// keeps the halo rows that neighbouring tiles share inside one L2.  Bijective for any grid size.
// reverse != 0: every XCD walks its range backwards (see km_traversal_next).
__device__ __forceinline__ uint32_t km_xcd_remap(uint32_t bid, uint32_t nblocks, uint32_t reverse = 0u) {
    const uint32_t NX = 8;
    const uint32_t q = nblocks / NX, r = nblocks % NX;
    const uint32_t xcd = bid % NX, k = bid / NX;
    // XCD x owns q (+1 if x < r) logical blocks, laid out back to back
    const uint32_t start = xcd * q + (xcd < r ? xcd : r);
    const uint32_t count = q + (xcd < r ? 1u : 0u);
    return start + (reverse ? count - 1u - k : k);
}

// LDS-DMA (gfx950 global_load_lds_dwordx4 / _dword): every lane names a 16-byte (4-byte) piece of global memory; the wave's pieces land in
// LDS at lds_wave_base + lane * 16 (* 4) - straight from the memory pipe, no vector register, no ds_write.  Issued as inline assembly
// (M0 carries the LDS byte address: saved, set, restored in the one statement), so the COMPILER DOES NOT KNOW the operation exists: it
// neither waits for it nor orders LDS reads behind it - the kernel does, with `KM_VMCNT0()` (the operation counts in vmcnt like any load,
// and loads complete in order) and a barrier before another wave reads the piece.  lds_wave_base: a pointer into LDS, the same in every
// lane of the wave.  (The host build of the kernels supplies functions with the same effect.)
#ifndef KM_GLDS16
__device__ __forceinline__ uint32_t km_lds_addr(const void* p) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p);
}
__device__ __forceinline__ void km_glds16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void km_glds4(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
#define KM_GLDS16(gsrc, lds_wave_base) km_glds16((const void*)(gsrc), km_lds_addr(lds_wave_base))
#define KM_GLDS4(gsrc, lds_wave_base) km_glds4((const void*)(gsrc), km_lds_addr(lds_wave_base))
#define KM_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// The thread index as a value the optimiser cannot hoist out of a loop (an empty volatile statement "writes" it): in a kernel that walks
// several tiles, everything that depends on the thread index alone - lane / wave / column / chunk indices, address offsets - is
// loop-invariant, gets hoisted and stays live across the whole body.  (The host build of the kernels: the plain index.)
#ifndef KM_TID_PINNED
__device__ __forceinline__ int km_tid_pinned() {
    int t = (int)threadIdx.x;
#ifndef KM_TID_UNPINNED
    asm volatile("" : "+v"(t));
#endif
    return t;
}
#endif

// Scheduling fence: the compiler does not move instructions across it (no code is emitted).  Used between independent unrolled
// bodies whose interleaving would raise the register count (the host build of the kernels defines it away).
#ifndef KM_SCHED_FENCE
#define KM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// Direction of the next launch of a streaming kernel on stream s (host side, km_runtime.hip).  The kernels of the hot step each stream
// ~0.8 GB in and out, three times the 256 MB Infinity Cache, so a consumer that walks the batch in the producer's order finds
// nothing of what the producer touched last.  Consecutive launches ON ONE STREAM therefore alternate direction: the consumer starts
// where the producer (or the previous reader of the same tensor) ended.  Measured on MI355X, config 2: step 1.864 -> 1.837 ms.
// The parity is per (device, stream); KM_TRAVERSAL=fixed / km_set_traversal(1) keeps every launch forward (A/B).
uint32_t km_traversal_next(hipStream_t s);

// Launch policy, read once from the environment when the library is first used (km_runtime.hip); see profiles/README.md
struct KmConfig {
    int traversal_fixed;    // KM_TRAVERSAL=fixed
    int warp_fwd_algo;      // KM_WARP_FWD_ALGO: 0 default, 1 generic, 3 box (km_warp_fwd_box_kernel), 4 rows (the gather kernel)
    int warp_gm_algo;       // KM_WARP_GM_ALGO: 0 default (box form where it applies), 1 generic, 2 lds, 3 rows (the gather kernel)
    int warp_bwd_generic;   // KM_WARP_BWD_ALGO=generic
    int warp_bwd_fused;     // KM_WARP_BWD_FUSED=0 turns the one-read backward off (two launches)
    int sep_lds;            // KM_SEP_ALGO=lds
    int sg_generic;         // KM_SG_ALGO=generic
    int pyrdown_separable;  // KM_PYRDOWN_ALGO=separable
    int blur_rows;          // KM_BLUR_ROWS=8 / 16 / 32 (0: by storage type, direction, kernel size and size of the launch: km_blur_rows)
    int warp_bwd_no_scan;   // KM_WARP_BWD_SCAN=0: the one-read backward does not look for non-finite gradients at output pixels that sample entirely outside the source
};
const KmConfig& km_config();
int km_device_cus();
// a side stream forked from `s` (work on it runs beside what `s` is given next) and its join; nullptr: launch on `s` itself (km_runtime.hip)
hipStream_t km_side_fork(hipStream_t s);
int km_side_join(hipStream_t s);

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): a __syncthreads() also waits for every global
// load and store of the wave (vmcnt(0)), which is exactly what a kernel that keeps the NEXT tile's loads in flight across the barrier
// must not do (km_warp_bwd_fused.hip).  Global memory written before it is NOT made visible by it.  (The host build of the kernels
// maps it to its ordinary barrier.)
#ifndef KM_LDS_BARRIER
#define KM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// wave-level sum (64 lanes) in double precision
__device__ __forceinline__ double km_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
