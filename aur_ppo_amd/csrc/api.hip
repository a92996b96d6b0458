// Error reporting and device probing for libaurppo_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

void aurppo_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* aurppo_last_error(void) { return g_err; }

extern "C" int aurppo_version(void) { return AURPPO_VERSION; }

extern "C" int aurppo_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        aurppo_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return AURPPO_EHIP;
    }
    return n;
}
