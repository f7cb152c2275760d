// kornia_amd - C-ABI runtime glue: error reporting, version, device query.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "km_common.h"

static thread_local char g_km_error[512] = "";

void km_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_km_error, sizeof(g_km_error), fmt, ap);
    va_end(ap);
}

static std::atomic<int> g_km_traversal_mode{-1};  // -1: not set (KM_TRAVERSAL=fixed in the environment selects 1), 0: alternate, 1: fixed
static int km_traversal_mode() {
    int m = g_km_traversal_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("KM_TRAVERSAL");
        m = (e && e[0] == 'f') ? 1 : 0;
        g_km_traversal_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
uint32_t km_traversal_next() {
    static std::atomic<uint32_t> n{0};
    return km_traversal_mode() ? 0u : (n.fetch_add(1u, std::memory_order_relaxed) & 1u);
}

int km_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        km_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" {

int km_abi_version(void) { return KM_ABI_VERSION; }

int km_set_traversal(int mode) {
    const int prev = km_traversal_mode();
    g_km_traversal_mode.store(mode ? 1 : 0, std::memory_order_relaxed);
    return prev;
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
