// TEST INFRASTRUCTURE ONLY: two toy kernels for the emulator's own tests (tests/test_emulated_kernels.py).
// `handoff` passes a value from every work-item to the one 64 places further through LDS; without the barrier the result
// depends on which wave runs first - correct by luck in launch order, wrong when the waves are resumed in another order.
#include <hip/hip_runtime.h>

template <bool BARRIER>
__global__ void handoff_kernel(int* out) {
    __shared__ int slot[256];
    slot[threadIdx.x] = (int)threadIdx.x + 1000 * (int)blockIdx.x;
    if (BARRIER) __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = slot[(threadIdx.x + 192) % 256];
}

// a wave operation that only half of a wave's lanes reach while the others wait at the workgroup barrier: the emulator has no
// execution mask, so it reports this (stricter than the hardware) instead of hanging - kernels keep wave operations in
// wave-uniform control flow
__global__ void divergent_wave_op_kernel(int* out) {
    int v = (int)threadIdx.x;
    if ((threadIdx.x & 63) < 32) v = __shfl_down(v, 1, 64);
    __syncthreads();
    out[threadIdx.x] = v;
}

// LDS-DMA: lane 0 requests two pieces and waits (KM_VMCNT0) in between - under the deferred model the first has landed when it is read,
// the second (never waited for) has not; under the immediate model both have
__global__ void glds_kernel(const float* src, float* out) {
    __shared__ __attribute__((aligned(16))) float piece[2][64];
    piece[0][threadIdx.x] = 0.f; piece[1][threadIdx.x] = 0.f;
    __syncthreads();
    KM_GLDS4(src + threadIdx.x, &piece[0][0]);
    KM_VMCNT0();
    KM_GLDS4(src + threadIdx.x, &piece[1][0]);
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = piece[0][5]; out[1] = piece[1][5]; }
}

extern "C" {
int selftest_glds(float* out) {
    static float src[64];
    for (int i = 0; i < 64; ++i) src[i] = 1.0f;
    hipLaunchKernelGGL(glds_kernel, dim3(1), dim3(64), 0, nullptr, (const float*)src, out);
    return hipGetLastError();
}
int selftest_handoff(int* out, int blocks, int with_barrier) {
    if (with_barrier)
        hipLaunchKernelGGL((handoff_kernel<true>), dim3(blocks), dim3(256), 0, nullptr, out);
    else
        hipLaunchKernelGGL((handoff_kernel<false>), dim3(blocks), dim3(256), 0, nullptr, out);
    return hipGetLastError();
}
int selftest_divergent_wave_op(int* out) {
    hipLaunchKernelGGL(divergent_wave_op_kernel, dim3(1), dim3(256), 0, nullptr, out);
    return hipGetLastError();
}
}
