// kornia_amd - C-ABI runtime glue: error reporting, version, device query.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <string.h>

#include <atomic>
#include <mutex>

#include "km_common.h"

static thread_local char g_km_error[512] = "";

void km_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_km_error, sizeof(g_km_error), fmt, ap);
    va_end(ap);
}

// ---- launch policy ---------------------------------------------------------------------------------------------------------------
// Read ONCE, when the library is first used (the A/B switches of profiles/README.md); launchers read the struct, never the
// environment.  km_config_set changes an entry explicitly (tests, A/B timing of one process) - it is not meant to race with launches.
static KmConfig g_km_config;
static std::once_flag g_km_config_once;
static int km_env_first(const char* name, char c0, char c1 = 0) {
    const char* e = getenv(name);
    if (!e) return 0;
    if (e[0] == c0) return 1;
    if (c1 && e[0] == c1) return 2;
    return 0;
}
static void km_config_init() {
    KmConfig& c = g_km_config;
    c.traversal_fixed = km_env_first("KM_TRAVERSAL", 'f');
    c.warp_fwd_algo = km_env_first("KM_WARP_FWD_ALGO", 'g');        // 1 generic, 3 box, 4 rows (the gather kernel)
    if (!c.warp_fwd_algo) c.warp_fwd_algo = km_env_first("KM_WARP_FWD_ALGO", 'b') ? 3 : (km_env_first("KM_WARP_FWD_ALGO", 'r') ? 4 : 0);
    c.warp_gm_algo = km_env_first("KM_WARP_GM_ALGO", 'g', 'l');     // 1 generic, 2 lds, 3 rows (the gather kernel where the box form would run)
    if (!c.warp_gm_algo && km_env_first("KM_WARP_GM_ALGO", 'r')) c.warp_gm_algo = 3;
    c.warp_bwd_generic = km_env_first("KM_WARP_BWD_ALGO", 'g');
    c.warp_bwd_fused = km_env_first("KM_WARP_BWD_FUSED", '0') ? 0 : 1;
    c.sep_lds = km_env_first("KM_SEP_ALGO", 'l');
    c.sg_generic = km_env_first("KM_SG_ALGO", 'g');
    c.pyrdown_separable = km_env_first("KM_PYRDOWN_ALGO", 's');
    const char* br = getenv("KM_BLUR_ROWS");
    c.blur_rows = br ? atoi(br) : 0;
    c.warp_bwd_no_scan = km_env_first("KM_WARP_BWD_SCAN", '0');
}
const KmConfig& km_config() {
    std::call_once(g_km_config_once, km_config_init);
    return g_km_config;
}

// Direction of the next launch of a streaming kernel ON THIS STREAM: the parity is per (device, stream), so what another thread,
// stream or device launches never changes the order a stream's own kernels see (re-entrant: the table is lock-free, an entry is
// claimed once and then only touched through its own atomic counter).  Beyond KM_STREAM_SLOTS live streams the direction stays fixed.
#define KM_STREAM_SLOTS 256
static std::atomic<uint64_t> g_km_stream_key[KM_STREAM_SLOTS];  // 0 = free
static std::atomic<uint32_t> g_km_stream_parity[KM_STREAM_SLOTS];
uint32_t km_traversal_next(hipStream_t s) {
    if (km_config().traversal_fixed) return 0u;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t key = ((uint64_t)(uintptr_t)s) ^ ((uint64_t)(dev + 1) << 56) ^ 0x8000000000000000ull;  // never 0
    uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 56);
    for (int probe = 0; probe < KM_STREAM_SLOTS; ++probe, h = (h + 1u) % KM_STREAM_SLOTS) {
        uint64_t k = g_km_stream_key[h].load(std::memory_order_acquire);
        if (k == 0) {
            uint64_t expect = 0;
            if (g_km_stream_key[h].compare_exchange_strong(expect, key, std::memory_order_acq_rel)) k = key;
            else k = expect;
        }
        if (k == key) return g_km_stream_parity[h].fetch_add(1u, std::memory_order_relaxed) & 1u;
    }
    return 0u;
}

// compute units of the current device (cached per device ordinal; 0 on the host build of the kernels)
int km_device_cus() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
        v = p.multiProcessorCount > 0 ? p.multiProcessorCount : -1;
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v > 0 ? v : 0;
}

// ---- a forked side stream per device ---------------------------------------------------------------------------------------------------
// km_side_fork(s): work launched on the returned stream starts after everything already queued on `s` and runs BESIDE what `s` is given next;
// km_side_join(s) makes `s` wait for it.  One non-blocking stream and two events per device, created on first use; the record / wait pairs of
// concurrent host threads are serialised by a mutex (a later thread's record only ever makes an earlier thread's wait cover more).  Returns
// nullptr - the caller then launches on `s` itself - while `s` is being captured into a graph, on the host build of the kernels (no device)
// and when a stream or an event cannot be created.
struct KmSide {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int state = 0;  // 0 untried, 1 ready, -1 unavailable
    std::mutex mu;
};
static KmSide g_km_side[64];
hipStream_t km_side_fork(hipStream_t s) {
    int dev = 0;
    if (km_device_cus() <= 0 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    KmSide& k = g_km_side[dev];
    std::lock_guard<std::mutex> lock(k.mu);
    if (k.state == 0) {
        k.state = (hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&k.fork, hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&k.join, hipEventDisableTiming) == hipSuccess) ? 1 : -1;
        if (k.state < 0) (void)hipGetLastError();
    }
    if (k.state < 0) return nullptr;
    if (hipEventRecord(k.fork, s) != hipSuccess || hipStreamWaitEvent(k.stream, k.fork, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return k.stream;
}
// after the side stream's work has been launched: `s` waits for it
int km_side_join(hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    KmSide& k = g_km_side[dev];
    std::lock_guard<std::mutex> lock(k.mu);
    if (k.state != 1) return -1;
    hipError_t e = hipEventRecord(k.join, k.stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(s, k.join, 0);
    if (e != hipSuccess) { km_set_error("km_side_join: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

int km_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        km_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// ---- km_stream_copy: the copy the hot kernels are compared with (bench.py: roofline.measured_streaming_copy_GBps) --------------------
// ONE 16-byte piece per thread, one workgroup per 4 KB, no loop: the shape that streams fastest on this part (profiles/r02_hbm_shapes.txt: 6.2 TB/s;
// two / four / eight pieces per thread 5.8 / 5.7 / 4.3, a grid-stride loop over ~8 resident workgroups per CU 5.2 - round 6, run 3)
template <bool NT>
__global__ __launch_bounds__(256) void km_stream_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16) {
    typedef float km_f4v __attribute__((ext_vector_type(4)));
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n16) {
        if (NT) {
            const km_f4v v = __builtin_nontemporal_load(reinterpret_cast<const km_f4v*>(src) + i);
            __builtin_nontemporal_store(v, reinterpret_cast<km_f4v*>(dst) + i);
        } else {
            dst[i] = src[i];
        }
    }
}

extern "C" {

int km_abi_version(void) { return KM_ABI_VERSION; }

int km_stream_copy(const void* src, void* dst, long long bytes, int nontemporal, void* stream) {
    KM_REQUIRE(bytes >= 0 && bytes % 16 == 0, "km_stream_copy: bytes must be a non-negative multiple of 16");
    if (bytes == 0) return 0;
    KM_REQUIRE(src && dst && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "km_stream_copy: null or unaligned pointer");
    const size_t n16 = (size_t)bytes / 16;
    const size_t blocks = (n16 + 255) / 256;
    KM_REQUIRE(blocks < (1ull << 31), "km_stream_copy: at most 8 TiB per call");
    hipStream_t s = (hipStream_t)stream;
    if (nontemporal) hipLaunchKernelGGL((km_stream_copy_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)src, (float4*)dst, n16);
    else hipLaunchKernelGGL((km_stream_copy_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)src, (float4*)dst, n16);
    return km_check_launch("km_stream_copy");
}

int km_set_traversal(int mode) {
    (void)km_config();
    const int prev = g_km_config.traversal_fixed;
    g_km_config.traversal_fixed = mode ? 1 : 0;
    return prev;
}

// Explicit setter of a launch-policy entry (the names of KmConfig's fields); returns the previous value, -1 for an unknown key.
// For tests and A/B timing: call it between launches, not concurrently with them.
static int* km_config_field(const char* key) {
    struct { const char* name; int* field; } table[] = {
        {"traversal_fixed", &g_km_config.traversal_fixed}, {"warp_fwd_algo", &g_km_config.warp_fwd_algo},
        {"warp_gm_algo", &g_km_config.warp_gm_algo},       {"warp_bwd_generic", &g_km_config.warp_bwd_generic},
        {"warp_bwd_fused", &g_km_config.warp_bwd_fused},   {"sep_lds", &g_km_config.sep_lds},
        {"sg_generic", &g_km_config.sg_generic},           {"pyrdown_separable", &g_km_config.pyrdown_separable},
        {"blur_rows", &g_km_config.blur_rows},                 {"warp_bwd_no_scan", &g_km_config.warp_bwd_no_scan},
    };
    if (key)
        for (auto& e : table)
            if (strcmp(e.name, key) == 0) return e.field;
    return nullptr;
}
int km_config_set(const char* key, int value) {
    (void)km_config();
    int* f = km_config_field(key);
    if (!f) { km_set_error("km_config_set: unknown key '%s'", key ? key : "(null)"); return -1; }
    const int prev = *f;
    *f = value;
    return prev;
}
// current value of a launch-policy entry (-1: unknown key)
int km_config_get(const char* key) {
    (void)km_config();
    const int* f = km_config_field(key);
    if (!f) { km_set_error("km_config_get: unknown key '%s'", key ? key : "(null)"); return -1; }
    return *f;
}

const char* km_last_error(void) { return g_km_error; }

// Fills name (up to n bytes) with the gcnArchName of the current device; returns CU count or <0.
int km_device_info(char* name, int n) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        km_set_error("hipGetDevice: %s", hipGetErrorString(e));
        return -1;
    }
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) {
        km_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
        return -1;
    }
    if (name && n > 0) snprintf(name, (size_t)n, "%s", p.gcnArchName);
    return p.multiProcessorCount;
}

}  // extern "C"
