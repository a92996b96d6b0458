// Shared helpers for libaurppo_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/aurppo.h"

void aurppo_set_error(const char* fmt, ...);

#define AURPPO_HIP_TRY(expr)                                                   \
    do {                                                                       \
        hipError_t e__ = (expr);                                               \
        if (e__ != hipSuccess) {                                               \
            aurppo_set_error("%s failed: %s", #expr, hipGetErrorString(e__));  \
            return AURPPO_EHIP;                                                \
        }                                                                      \
    } while (0)

#define AURPPO_LAUNCH_CHECK(name)                                              \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            aurppo_set_error("launch of %s failed: %s", name, hipGetErrorString(e__)); \
            return AURPPO_EHIP;                                                \
        }                                                                      \
    } while (0)

#define AURPPO_REQUIRE(cond, code, ...)                                        \
    do {                                                                       \
        if (!(cond)) {                                                         \
            aurppo_set_error(__VA_ARGS__);                                     \
            return (code);                                                     \
        }                                                                      \
    } while (0)

constexpr int kWave = 64;

// Per-device launch state (function attributes are per device; one process may drive several): slot of the calling
// thread's current device in small static tables.
constexpr int kMaxDevices = 16;
static inline int aurppo_device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
    return d;
}
// Diagnostic knobs.  Every environment variable the library looks at is parsed in ONE place (api.hip) into this struct,
// once per process -- or on every call when AURPPO_TEST_KNOBS=1 (tests/conftest.py), so that one test process can flip
// them.  None of them changes results beyond summation order; the defaults are the product configuration.
struct AurppoKnobs {
    int k7_variant;        // AURPPO_K7_VARIANT: 3 (default) = 3 x bf16-split MFMA k_mlp_step3 (mlp3.hip), 2 = f32 MFMA k_mlp_step2; always one of the two (api.hip normalises)
    int k7w_variant;       // AURPPO_K7W_VARIANT: 3 (default) = k_mlpw3_step (bf16x3 MFMA) for the shapes wider than 64, 2 = the fp32-MFMA k_mlpw_step
    int k7_spare_cus;      // AURPPO_MLP_SPARE_CUS: CUs K7 / K7w leave to the side stream's shuffle kernels (default 8)
    int static_tiles;      // AURPPO_STATIC_TILES=1: K7 / K7w deal tiles by static stride instead of through the counter, which
                           // fixes the order of every sum (bit-reproducible gradients; tests/test_determinism.py)
    int k2_one_stream;     // AURPPO_K2_ONE_STREAM=1: twist and resolves on the caller's stream (everything serial)
    int k2_link_wgs;       // AURPPO_K2_LINK_WGS: workgroups of k_fy_link (default 48; 0 = one per 256 positions)
    int k2_resolve_wgs;    // AURPPO_K2_RESOLVE_WGS: workgroups of k_fy_resolve (default 256; 0 = one per 256 positions)
    int k2_post_stream;    // AURPPO_K2_POST_STREAM: 1 (default since round 3) = link + resolve on a third stream of the handle; 0 = behind accept / fill
    int k2_starve;         // AURPPO_TEST_K2_STARVE (tests): the twist keeps this per cent of one shuffle's draws in stock, so shuffles run dry
    int k2_accept3_wgs;    // AURPPO_K2_ACCEPT3_WGS: workgroups of k_fy_accept3's relay (default 6)
    int k2_accept;         // AURPPO_K2_ACCEPT: 3 (default) = k_fy_accept3 (relay between workgroups), 1 = k_fy_accept (one workgroup, rounds); always one of the two
    int gather_unroll;     // AURPPO_GATHER_UNROLL (0 = by row width)
    int gather_rows;       // AURPPO_GATHER_ROWS (0 = by row width)
};
const AurppoKnobs& aurppo_knobs();

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// Sum over a workgroup of NW waves; result valid in thread 0.  `scratch` holds NW doubles per
// concurrent value; callers separate uses with __syncthreads().
template <int NW>
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    v = wave_sum(v);
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < NW; ++w) r += scratch[w];
    }
    return r;
}

static inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
