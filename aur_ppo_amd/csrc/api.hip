// Error reporting and device probing for libaurppo_hip.so.
#include <stdarg.h>
#include <atomic>
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
    k.k7_variant = env_int("AURPPO_K7_VARIANT", 3) == 2 ? 2 : 3;   // normalised HERE, once: 2 = k_mlp_step2, anything else = the default k_mlp_step3
    k.k7w_variant = env_int("AURPPO_K7W_VARIANT", 3) == 2 ? 2 : 3;
    k.k7_spare_cus = env_int("AURPPO_MLP_SPARE_CUS", 8);
    k.static_tiles = env_int("AURPPO_STATIC_TILES", 0);
    k.k2_one_stream = env_int("AURPPO_K2_ONE_STREAM", 0);
    k.k2_link_wgs = env_int("AURPPO_K2_LINK_WGS", 48);
    k.k2_resolve_wgs = env_int("AURPPO_K2_RESOLVE_WGS", 256);
    k.k2_post_stream = env_int("AURPPO_K2_POST_STREAM", 1);
    k.k2_accept = env_int("AURPPO_K2_ACCEPT", 3) == 1 ? 1 : 3;       // 1 = k_fy_accept, anything else = the default k_fy_accept3
    k.k2_accept3_wgs = env_int("AURPPO_K2_ACCEPT3_WGS", 6);
    k.k2_starve = env_int("AURPPO_TEST_K2_STARVE", 0);
    k.gather_unroll = env_int("AURPPO_GATHER_UNROLL", 0);
    k.gather_rows = env_int("AURPPO_GATHER_ROWS", 0);
    return k;
}
}  // namespace

namespace {
std::atomic<unsigned> g_knob_gen{0};
}

const AurppoKnobs& aurppo_knobs() {
    static const bool live = env_int("AURPPO_TEST_KNOBS", 0) == 1;
    static thread_local AurppoKnobs k = parse_knobs();
    static thread_local unsigned seen = g_knob_gen.load(std::memory_order_relaxed);
    const unsigned gen = g_knob_gen.load(std::memory_order_relaxed);
    if (live || gen != seen) {
        k = parse_knobs();
        seen = gen;
    }
    return k;
}

// Re-read the environment knobs once, now (every thread picks the new values up at its next call).  bench.py times the
// plain-fp32 K7 beside the default one AFTER its timed region this way, so that the timed region itself runs on the
// product's parse-once configuration.
extern "C" int aurppo_reload_knobs(void) {
    g_knob_gen.fetch_add(1u, std::memory_order_relaxed);
    (void)aurppo_knobs();
    return AURPPO_OK;
}

extern "C" const char* aurppo_last_error(void) { return g_err; }

extern "C" int aurppo_version(void) { return AURPPO_VERSION; }

extern "C" int aurppo_k7_variant(void) {
    return aurppo_knobs().k7_variant;
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
