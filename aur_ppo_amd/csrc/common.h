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
// Diagnostic knobs (AURPPO_MLP_VARIANT ...) are read once per process; with AURPPO_TEST_KNOBS=1 (tests/conftest.py)
// they are re-read on every call so that one test process can run both variants.
static inline bool aurppo_live_knobs() {
    static const bool live = [] { const char* e = getenv("AURPPO_TEST_KNOBS"); return e && *e == '1'; }();
    return live;
}

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
