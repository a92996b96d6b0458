// K1: GAE(lambda) / discounted-return backward scan over the time-major (T, N) rollout buffer.
//
// Reference arithmetic: ppo.run_gae (src/ppo.py:125-142), ppo.normal_advantage
// (src/ppo.py:145-157), robot_ppo.run_gae (src/robot_ppo.py:224-244).
//
// Design (gfx950).  The recurrence X[t] = d[t] + c[t] * X[t+1] is sequential in t but every d[t],
// c[t] is independent of it.  A 256-thread workgroup owns TILE_N adjacent envs:
//   phase 1 (all 256 lanes): coalesced row-segment loads of r, V, done for a 128-step slab,
//            d/c computed with the reference's exact fp32 association, staged to LDS;
//   phase 2 (TILE_N lanes):  the dependent chain -- one multiply + one add per step, operands
//            prefetched from LDS eight steps at a time, so the chain is ~2 dependent VALU ops/step;
//   phase 3 (all lanes):     A and R written back as coalesced row segments.
// Keeping the chain in program order (and building with -ffp-contract=off) makes the result
// bit-identical to the reference's CPU loop; re-associating the scan would not be.
// HBM traffic = 12 B read + 8 B written per (t, n) cell: the algorithmic minimum.
#include "common.h"

namespace {

constexpr int kSlab = 128;  // timesteps staged in LDS per pass

template <int TILE_N>
__global__ __launch_bounds__(256) void k_gae(const float* __restrict__ rewards, const float* __restrict__ values,
                                             const float* __restrict__ terminals,
                                             const float* __restrict__ next_value,
                                             const float* __restrict__ next_done, float* __restrict__ adv,
                                             float* __restrict__ ret, const float* __restrict__ log_probs,
                                             float4* __restrict__ rec, int T, int N, float g, float gl, int mode) {
    constexpr int ROWS = 256 / TILE_N;
    __shared__ float s_d[kSlab][TILE_N];  // d[t], then X[t]
    __shared__ float s_c[kSlab][TILE_N];  // c[t]
    __shared__ float s_v[kSlab][TILE_N];  // V[t]
    const int e = threadIdx.x % TILE_N;
    const int tr = threadIdx.x / TILE_N;
    const int n = blockIdx.x * TILE_N + e;
    const bool valid = n < N;
    const bool normal = (mode == AURPPO_NORMAL_ADV);
    float carry = 0.0f;
    if (tr == 0 && valid && normal) carry = next_value[n];

    for (int t_hi = T; t_hi > 0; t_hi -= kSlab) {
        const int t_lo = t_hi > kSlab ? t_hi - kSlab : 0;
        const int len = t_hi - t_lo;
        // ---- phase 1: d, c for the slab.  All of a thread's loads are issued before the first use (a full slab is
        // kSlab / ROWS passes of 3-4 loads, unrolled): one memory round trip instead of one per pass.
        if (valid) {
            constexpr int kPass = kSlab / ROWS;
            float rt[kPass], vt[kPass], nv[kPass], nd[kPass];
#pragma unroll
            for (int u = 0; u < kPass; ++u) {
                const int tt = tr + u * ROWS;
                const int t = t_lo + (tt < len ? tt : len - 1);      // clamped: padding passes re-read the last row
                const size_t at = (size_t)t * N + n;
                rt[u] = rewards[at];
                vt[u] = values[at];
                const bool last = t == T - 1;
                nv[u] = last ? next_value[n] : values[at + N];
                nd[u] = last ? next_done[n] : terminals[at + N];
            }
#pragma unroll
            for (int u = 0; u < kPass; ++u) {
                const int tt = tr + u * ROWS;
                if (tt >= len) continue;
                const int t = t_lo + tt;
                const float nnt = 1.0f - nd[u];
                float d_, c_;
                if (normal) {
                    d_ = rt[u];
                    c_ = g * nnt;
                } else {
                    d_ = (rt[u] + (g * nv[u]) * nnt) - vt[u];
                    c_ = gl * nnt;
                    if (mode == AURPPO_GAE_SKIP_LAST && t == T - 1) {
                        d_ = 0.0f;
                        c_ = 0.0f;
                    }
                }
                s_d[tt][e] = d_;
                s_c[tt][e] = c_;
                s_v[tt][e] = vt[u];
            }
        }
        __syncthreads();
        // ---- phase 2: the dependent chain, in reference order
        if (tr == 0 && valid) {
            float x = carry;
            int tt = len - 1;
            for (; tt >= 7; tt -= 8) {
                float d8[8], c8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    d8[k] = s_d[tt - k][e];
                    c8[k] = s_c[tt - k][e];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    x = d8[k] + c8[k] * x;
                    s_d[tt - k][e] = x;
                }
            }
            for (; tt >= 0; --tt) {
                x = s_d[tt][e] + s_c[tt][e] * x;
                s_d[tt][e] = x;
            }
            carry = x;
        }
        __syncthreads();
        // ---- phase 3: write back
        if (valid) {
            for (int tt = tr; tt < len; tt += ROWS) {
                const size_t at = (size_t)(t_lo + tt) * N + n;
                const float x = s_d[tt][e];
                const float vt = s_v[tt][e];
                const float a_ = normal ? x - vt : x;
                const float r_ = normal ? x : x + vt;
                adv[at] = a_;
                ret[at] = r_;
                // optional per-sample record {old_logp, A, R, V}: one 16-B request per sample for the
                // minibatch gather instead of four divergent 4-B ones
                if (rec) rec[at] = make_float4(log_probs[at], a_, r_, vt);
            }
        }
        __syncthreads();
    }
}

}  // namespace

static int gae_launch(const float* rewards, const float* values, const float* terminals, const float* next_value,
                      const float* next_done, float* advantages, float* returns, const float* log_probs, float* rec,
                      int T, int N, double gamma, double lam, int mode, void* stream) {
    AURPPO_REQUIRE(rewards && values && terminals && next_value && next_done && advantages && returns, AURPPO_EINVAL,
                   "aurppo_gae_f32: null pointer");
    AURPPO_REQUIRE(!rec || aligned_to(rec, 16), AURPPO_EINVAL, "aurppo_gae_pack_f32: rec not 16-byte aligned");
    AURPPO_REQUIRE(mode >= 0 && mode <= 2, AURPPO_EINVAL, "aurppo_gae_f32: bad mode %d", mode);
    AURPPO_REQUIRE(T > 0 && N > 0, AURPPO_ESHAPE, "aurppo_gae_f32: T=%d N=%d must be positive", T, N);
    const float g = (float)gamma;
    const float gl = (float)(gamma * lam);  // folded in fp64 first, as Python does (src/ppo.py:140)
    hipStream_t s = (hipStream_t)stream;
    // Narrow tiles put a workgroup on every CU at N=4096; widen them only when N alone fills the chip.
    if (N >= 16384) {
        hipLaunchKernelGGL(k_gae<64>, dim3((N + 63) / 64), dim3(256), 0, s, rewards, values, terminals, next_value,
                           next_done, advantages, returns, log_probs, reinterpret_cast<float4*>(rec), T, N, g, gl, mode);
    } else if (N >= 8192) {
        hipLaunchKernelGGL(k_gae<32>, dim3((N + 31) / 32), dim3(256), 0, s, rewards, values, terminals, next_value,
                           next_done, advantages, returns, log_probs, reinterpret_cast<float4*>(rec), T, N, g, gl, mode);
    } else {
        hipLaunchKernelGGL(k_gae<16>, dim3((N + 15) / 16), dim3(256), 0, s, rewards, values, terminals, next_value,
                           next_done, advantages, returns, log_probs, reinterpret_cast<float4*>(rec), T, N, g, gl, mode);
    }
    AURPPO_LAUNCH_CHECK("k_gae");
    return AURPPO_OK;
}

extern "C" int aurppo_gae_f32(const float* rewards, const float* values, const float* terminals,
                              const float* next_value, const float* next_done, float* advantages, float* returns,
                              int T, int N, double gamma, double lam, int mode, void* stream) {
    return gae_launch(rewards, values, terminals, next_value, next_done, advantages, returns, nullptr, nullptr, T, N,
                      gamma, lam, mode, stream);
}

extern "C" int aurppo_gae_pack_f32(const float* rewards, const float* values, const float* terminals,
                                   const float* next_value, const float* next_done, const float* log_probs,
                                   float* advantages, float* returns, float* rec, int T, int N, double gamma,
                                   double lam, int mode, void* stream) {
    AURPPO_REQUIRE(log_probs && rec, AURPPO_EINVAL, "aurppo_gae_pack_f32: null pointer");
    return gae_launch(rewards, values, terminals, next_value, next_done, advantages, returns, log_probs, rec, T, N,
                      gamma, lam, mode, stream);
}
