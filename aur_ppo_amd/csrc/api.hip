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

namespace {
int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
AurppoKnobs parse_knobs() {
    AurppoKnobs k;
    k.k7_variant = env_int("AURPPO_K7_VARIANT", 3);   // default: k_mlp_step3 (see DESIGN 4.3d for the measurements behind it)
    k.k7_spare_cus = env_int("AURPPO_MLP_SPARE_CUS", 8);
    k.static_tiles = env_int("AURPPO_STATIC_TILES", 0);
    k.k2_one_stream = env_int("AURPPO_K2_ONE_STREAM", 0);
    k.k2_link_wgs = env_int("AURPPO_K2_LINK_WGS", 48);
    k.k2_resolve_wgs = env_int("AURPPO_K2_RESOLVE_WGS", 256);
    k.k2_post_stream = env_int("AURPPO_K2_POST_STREAM", 1);
    k.k2_accept = env_int("AURPPO_K2_ACCEPT", 3);
    k.k2_accept3_wgs = env_int("AURPPO_K2_ACCEPT3_WGS", 6);
    k.k2_starve = env_int("AURPPO_TEST_K2_STARVE", 0);
    k.gather_unroll = env_int("AURPPO_GATHER_UNROLL", 0);
    k.gather_rows = env_int("AURPPO_GATHER_ROWS", 0);
    return k;
}
}  // namespace

const AurppoKnobs& aurppo_knobs() {
    static const bool live = env_int("AURPPO_TEST_KNOBS", 0) == 1;
    static thread_local AurppoKnobs k = parse_knobs();
    if (live) k = parse_knobs();
    return k;
}

extern "C" const char* aurppo_last_error(void) { return g_err; }

extern "C" int aurppo_version(void) { return AURPPO_VERSION; }

extern "C" int aurppo_k7_variant(void) {
    const int v = aurppo_knobs().k7_variant;
    return (v == 3 || v == 4) ? v : 2;
}

extern "C" int aurppo_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        aurppo_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return AURPPO_EHIP;
    }
    return n;
}
