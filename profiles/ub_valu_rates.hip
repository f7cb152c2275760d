// scratch micro-benchmark: issue rate of the VALU / LDS instruction classes the warp kernels are made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 profiles/ub_valu_rates.hip -o scratch/ub/valu_rates && scratch/ub/valu_rates
// Every kernel runs ITER iterations of 16 independent copies of one instruction per wave; the grid puts W waves on
// every SIMD (256 CUs x 4 SIMDs).  Reported: SIMD cycles per wave-instruction = time * f / (ITER * 16 * W), with f from
// the s_memtime / wall_clock64 ratio measured in the same run.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ITER 2000

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int OP>
__global__ __launch_bounds__(256) void k_rate(float* out, float seed, unsigned long long* clk) {
    float a[16];
    int ia[16];
    float2 p[16];
    __shared__ float lds[4096 + 64];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        a[k] = seed + (float)threadIdx.x * 1e-3f + (float)k;
        ia[k] = (int)threadIdx.x + k * 3 + (int)seed;
        p[k] = make_float2(a[k], a[k] + 0.5f);
    }
    for (int e = threadIdx.x; e < 4096 + 64; e += 256) lds[e] = (float)e;
    __syncthreads();
    const float b = seed * 0.999f + 0.5f, c = seed * 1e-3f;
    const float2 pb = make_float2(b, b), pc = make_float2(c, c);
    const int laddr = (int)(threadIdx.x * 4 + 4 * (int)seed) * 4;  // byte address, 16-B multiples of lane
    const int laddr_u = (int)(threadIdx.x * 3 + (int)seed) * 4;    // 12-byte stride: only 4-byte aligned
    const int laddr4 = (int)((threadIdx.x & 63) + 4 * (int)seed) * 4;   // 4-byte lane stride (waves of a block share the words)
    const int laddr8 = (int)((threadIdx.x & 63) + 4 * (int)seed) * 8;   // 8-byte lane stride
    unsigned long long t0 = 0, t1 = 0, w0 = 0, w1 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
    for (int it = 0; it < ITER; ++it) {
        if (OP == 0) {
#define S(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
            REP16(S)
#undef S
        } else if (OP == 1) {
#define S(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(pb), "v"(pc));
            REP16(S)
#undef S
        } else if (OP == 2) {
#define S(k) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
            REP16(S)
#undef S
        } else if (OP == 3) {
#define S(k) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[k]) : "v"(pb));
            REP16(S)
#undef S
        } else if (OP == 4) {
#define S(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
            REP16(S)
#undef S
        } else if (OP == 5) {
#define S(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ia[k]) : "v"(ia[(k + 1) & 15]));
            REP16(S)
#undef S
        } else if (OP == 6) {
#define S(k) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(ia[k]) : "v"(a[k]));
            REP16(S)
#undef S
        } else if (OP == 7) {
#define S(k) asm volatile("v_floor_f32 %0, %0" : "+v"(a[k]));
            REP16(S)
#undef S
        } else if (OP == 8) {
#define S(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : "vcc");
            REP16(S)
#undef S
        } else if (OP == 9) {
#define S(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[k]) : "v"(ia[(k + 1) & 15]));
            REP16(S)
#undef S
        } else if (OP == 10) {
#define S(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(ia[k]) : "v"(ia[(k + 1) & 15]));
            REP16(S)
#undef S
        } else if (OP == 11) {
#define S(k) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[k]) : "v"(b) : "vcc");
            REP16(S)
#undef S
        } else if (OP == 12) {
#define S(k) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            REP16(S)
#undef S
        } else if (OP == 13) {
#define S(k) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            REP16(S)
#undef S
        } else if (OP == 14) {  // LDS: ds_read_b32
#define S(k) asm volatile("ds_read_b32 %0, %1 offset:" #k "*4" : "=v"(a[k]) : "v"(laddr));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == 15) {  // LDS: ds_read2_b32 (two dwords, adjacent)
#define S(k) asm volatile("ds_read2_b32 %0, %1 offset0:" #k " offset1:" #k "+1" : "=v"(p[k]) : "v"(laddr_u));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == 16) {  // LDS: ds_read_b64 at 4-byte aligned (unaligned) addresses
#define S(k) asm volatile("ds_read_b64 %0, %1 offset:" #k "*4" : "=v"(p[k]) : "v"(laddr_u));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == 17) {  // LDS: ds_read_b64, 8-byte aligned
#define S(k) asm volatile("ds_read_b64 %0, %1 offset:" #k "*8" : "=v"(p[k]) : "v"(laddr));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == 18) {  // ds_add_u32 (no return), consecutive lanes -> consecutive dwords
#define S(k) asm volatile("ds_add_u32 %0, %1 offset:" #k "*4" : : "v"(laddr), "v"(ia[k]));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == 24) {  // ds_add_u32, lane stride 4 bytes: conflict-free (OP 18's 16-byte lane stride is a 4-way bank conflict)
#define S(k) asm volatile("ds_add_u32 %0, %1 offset:" #k "*256" : : "v"(laddr4), "v"(ia[k]));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == 25) {  // ds_read_b32, lane stride 4 bytes: conflict-free
#define S(k) asm volatile("ds_read_b32 %0, %1 offset:" #k "*256" : "=v"(a[k]) : "v"(laddr4));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == 26) {  // ds_add_u64, lane stride 8 bytes
#define S(k) asm volatile("ds_add_u64 %0, %1 offset:" #k "*512" : : "v"(laddr8), "v"(p[k]));
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == 19) {  // v_cvt_rpi_i32_f32
#define S(k) asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(ia[k]) : "v"(a[k]));
            REP16(S)
#undef S
        } else if (OP == 20) {  // v_mul_f32
#define S(k) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
            REP16(S)
#undef S
        } else if (OP == 21) {  // v_pk_mul_f32
#define S(k) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[k]) : "v"(pb));
            REP16(S)
#undef S
        } else if (OP == 22) {  // 64-bit address add
            unsigned long long* q = (unsigned long long*)p;
#define S(k) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q[k]) : "v"(q[(k + 1) & 15]));
            REP16(S)
#undef S
        } else if (OP == 23) {  // dependent chain of fma (latency)
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
            REP16(S)
#undef S
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        t1 = __builtin_readcyclecounter(); w1 = wall_clock64();
        clk[0] = t1 - t0; clk[1] = w1 - w0;
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k] + (float)ia[k] + p[k].x + p[k].y;
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int OP>
static void run(const char* name, int waves_per_simd, float* out, unsigned long long* clk) {
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double n = (double)ITER * 16.0;
    // per-wave view: cycles between start and end of block 0 / instructions issued by one wave
    printf("%-28s W=%d  %8.3f ms   wave0: %7.2f cyc/instr (cycle counter), %7.2f ticks(100MHz)/1000instr   all: %6.2f ns*SIMD/instr\n", name, waves_per_simd, ms,
           (double)h[0] / n, (double)h[1] / n * 1000.0, (double)ms * 1e6 / (n * waves_per_simd));
}

int main(int argc, char** argv) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 4096); hipMalloc(&clk, 64);
    const int ws[3] = {1, 2, 4};
    for (int wi = 0; wi < 3; ++wi) {
        const int W = ws[wi];
        run<0>("v_fma_f32", W, out, clk);
        run<1>("v_pk_fma_f32", W, out, clk);
        run<2>("v_add_f32", W, out, clk);
        run<3>("v_pk_add_f32", W, out, clk);
        run<20>("v_mul_f32", W, out, clk);
        run<21>("v_pk_mul_f32", W, out, clk);
        run<4>("v_rcp_f32", W, out, clk);
        run<5>("v_mul_lo_u32", W, out, clk);
        run<10>("v_mad_u32_u24", W, out, clk);
        run<9>("v_add_u32", W, out, clk);
        run<22>("v_lshl_add_u64", W, out, clk);
        run<6>("v_cvt_i32_f32", W, out, clk);
        run<19>("v_cvt_rpi_i32_f32", W, out, clk);
        run<7>("v_floor_f32", W, out, clk);
        run<8>("v_cmp+v_cndmask (2 instr)", W, out, clk);
        run<11>("v_div_scale_f32", W, out, clk);
        run<12>("v_div_fixup_f32", W, out, clk);
        run<13>("v_med3_f32", W, out, clk);
        run<23>("v_fma_f32 dependent", W, out, clk);
        run<14>("ds_read_b32 (16B lane stride)", W, out, clk);
        run<15>("ds_read2_b32 (12B stride)", W, out, clk);
        run<16>("ds_read_b64 unaligned", W, out, clk);
        run<17>("ds_read_b64 aligned", W, out, clk);
        run<18>("ds_add_u32 (16B lane stride)", W, out, clk);
        run<24>("ds_add_u32 conflict-free", W, out, clk);
        run<25>("ds_read_b32 conflict-free", W, out, clk);
        run<26>("ds_add_u64 (8B lane stride)", W, out, clk);
    }
    return 0;
}
