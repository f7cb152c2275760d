// kornia_amd - C-ABI runtime glue: error reporting, version, device query.
#include <stdarg.h>
#include <stdio.h>

#include "km_common.h"

static thread_local char g_km_error[512] = "";

void km_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_km_error, sizeof(g_km_error), fmt, ap);
    va_end(ap);
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
