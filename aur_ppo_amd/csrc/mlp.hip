// K7 host side + the small kernels around the fused PPO minibatch step for the reference's MLP actor-critic
// (src/models/actor_critic.py:8-51 over src/nets/nets.py:19-53: two Tanh hidden layers of 64, Gaussian head with a
// state-independent log-std or Categorical head): gather (a7) + policy/value forward (a8) + advantage normalisation and
// clipped-surrogate loss (a9/a10) + back-propagation, producing the flat parameter-gradient bucket and the 9 loss scalars.
//
// Why: rocprof of the per-op path (profiles/r01) shows ~100 small kernels per minibatch moving ~2 GB through HBM for
// 11 GFLOP of work.  A minibatch reads each sample's observation row and its packed record ONCE (through the
// permutation, so K3's copy disappears too) and writes only per-workgroup gradient slabs; every activation lives in
// LDS / registers.  The step kernels themselves: mlp2.hip (k_mlp_step2, fp32 MFMA) and mlp3.hip (k_mlp_step3, the same
// step on bf16 MFMAs over three-way bf16 splits of every fp32 operand).  Here:
//   * k_adv_stats_idx  -- advantage partial sums of a minibatch + operand-order copies of the weights the step streams;
//   * k_mlp_reduce     -- sums the slabs in fixed order (deterministic), folds the loss scalars, leaves the clip's partial sums;
//   * k_adam_chain     -- clip + Adam, refreshes the operand-order copies, prepares the next minibatch's statistics;
//   * k_mlp_act (K8)   -- the rollout step (forward only);
//   * k_pack_rec64     -- record + action row in one 64-B line per sample.
// (Round 1-2's one-tile-set kernel k_mlp_step -- 4 waves, 221 us -- was kept for A/B until round 3 and is gone.)
#include <stdlib.h>

#include "adam_math.h"
#include "bf16x3.h"
#include "mlp_common.h"

using namespace aurppo_mlp;

namespace {

// Also lays out W1 of both nets in the B-operand order of k_mlp_step2's layer-1 MFMA chain (w1op != nullptr):
// w1op[(w * 32 + m) * 64 + lane] = W1[net = w >> 1][(w & 1) * 32 + (lane & 31)][2m + (lane >> 5)], zero beyond D, so
// that wave w reads its slice with 32 fully coalesced loads per tile instead of holding it in registers.
__global__ __launch_bounds__(256) void k_adv_stats_idx(const float4* __restrict__ rec, int rec_stride,
                                                       const int32_t* __restrict__ idx,
                                                       int M, double (*__restrict__ stats)[2], const float* __restrict__ params,
                                                       int w1_actor, int w1_critic, int D, float* __restrict__ w1op,
                                                       unsigned* __restrict__ tile_counter) {
    __shared__ double sc[2][kThreads / kWave];
    if (w1op) {
        if (blockIdx.x == 0 && threadIdx.x < 2) tile_counter[threadIdx.x] = 0u;
        for (int e = blockIdx.x * kThreads + threadIdx.x; e < 4 * 32 * 64; e += gridDim.x * kThreads) {
            const int lane = e & 63, m = (e >> 6) & 31, w = e >> 11;
            const int row = (w & 1) * 32 + (lane & 31), k = 2 * m + (lane >> 5);
            w1op[e] = k < D ? params[((w >> 1) ? w1_critic : w1_actor) + row * D + k] : 0.0f;
        }
    }
    double s = 0.0, q = 0.0;
    adv_partial_sums(rec, rec_stride, idx, M, blockIdx.x * kThreads + threadIdx.x, gridDim.x * kThreads, s, q);
    const double bs = block_sum<kThreads / kWave>(s, sc[0]);
    const double bq = block_sum<kThreads / kWave>(q, sc[1]);
    if (threadIdx.x == 0) {
        stats[blockIdx.x][0] = bs;
        stats[blockIdx.x][1] = bq;
    }
}

// ---- K8: forward-only sibling of K7 for the rollout step (src/ppo.py:103-108: policy.evaluate(next_obs) under
// no_grad, then buffer.values / actions / log_probs [step] = ...).  One workgroup per 32-row tile: both nets'
// forward pass as in K7, then 32 lanes sample the action from caller-supplied noise (a ~ mu + sigma*eps for the
// Gaussian head, inverse CDF of softmax(logits) at u for the Categorical head), form its log-prob and write
// action / log-prob / value straight into the rollout buffer rows.  With noise == nullptr only the value is
// produced (the bootstrap policy.value(next_obs), src/ppo.py:161).
struct ActArgs {
    const float* obs;     // (N, D)
    const float* noise;   // (N, A) standard normal | (N,) uniform [0,1) | nullptr
    const float* params;
    float* actions;       // (N, A) | (N,)
    float* logp;          // (N,)
    float* value;         // (N,)
    int N, D, A, continuous;
    MlpLayout L;
};

__global__ __launch_bounds__(256, 1) void k_mlp_act(const ActArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sX = lds;                    // [R][LD]
    float* sH1 = sX + R * LD;           // [2][R][LD]
    float* sH2 = sH1 + 2 * R * LD;      // [2][R][LD]
    float* sW1 = sH2 + 2 * R * LD;      // [2][H][LD]
    float* sW2 = sW1 + 2 * H * LD;      // [2][H][LD]
    float* sW3 = sW2 + 2 * H * LD;      // [2][AP][LD]
    float* sOut = sW3 + 2 * AP * LD;    // [2][R][LDO]
    float* sB1 = sOut + 2 * R * LDO;    // [2][H]
    float* sB2 = sB1 + 2 * H;
    float* sB3 = sB2 + 2 * H;           // [2][AP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int net = wave >> 1, cb = wave & 1;
    const int D = a.D, A = a.A;
    const int out_dim[2] = {A, 1};
    const int row0 = blockIdx.x * R;
    // Staging: every element's address comes from shifts of a flat index over rows padded to 64 columns (no division,
    // no zero-fill pass), and all of a thread's loads are independent, so they go out together -- the kernel is a
    // latency chain (launch, staging, three layers, sampling) and this was its longest link.
    float* sLs = sB3 + 2 * AP;          // [AP] log-std
    float* sSd = sLs + AP;              // [AP] exp(log-std)
#pragma unroll
    for (int u = 0; u < R * H / kThreads; ++u) {
        const int e = tid + u * kThreads, r = e >> 6, c = e & 63;
        // branch-free: a padding lane reads a clamped (valid) element and masks it -- a load under a divergent
        // condition makes the compiler wait for everything in flight
        const int rr = row0 + r < a.N ? row0 + r : a.N - 1, cc = c < D ? c : D - 1;
        const float x = a.obs[(size_t)rr * D + cc];
        sX[r * LD + c] = (c < D && row0 + r < a.N) ? x : 0.0f;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
        for (int u = 0; u < H * H / kThreads; ++u) {
            const int e = tid + u * kThreads, o = e >> 6, c = e & 63;
            const float w1v = a.params[a.L.w1[n] + o * D + (c < D ? c : D - 1)];
            const float w2v = a.params[a.L.w2[n] + e];
            sW1[(n * H + o) * LD + c] = c < D ? w1v : 0.0f;
            sW2[(n * H + o) * LD + c] = w2v;
        }
#pragma unroll
        for (int u = 0; u < AP * H / kThreads; ++u) {
            const int e = tid + u * kThreads, o = e >> 6, i = e & 63;
            sW3[(n * AP + o) * LD + i] = o < out_dim[n] ? a.params[a.L.w3[n] + o * H + i] : 0.0f;
        }
        if (tid < H) {
            sB1[n * H + tid] = a.params[a.L.b1[n] + tid];
            sB2[n * H + tid] = a.params[a.L.b2[n] + tid];
        }
        if (tid < AP) sB3[n * AP + tid] = tid < out_dim[n] ? a.params[a.L.b3[n] + tid] : 0.0f;
    }
    if (tid < AP) {
        const float ls = (a.continuous && tid < A) ? a.params[a.L.logstd + tid] : 0.0f;
        sLs[tid] = ls;
        sSd[tid] = expf(ls);
    }
    __syncthreads();
    {
        f32x16 acc = zero16();
        const float* W = sW1 + (net * H + cb * 32) * LD;
        mma32<H>(acc, [&](int i, int k) { return sX[i * LD + k]; }, [&](int k, int j) { return W[j * LD + k]; });
        const int col = cb * 32 + (lane & 31);
        const float bias = sB1[net * H + col];
#pragma unroll
        for (int e = 0; e < 16; ++e) sH1[(net * R + acc_row(e, lane)) * LD + col] = tanh_fast(acc[e] + bias);
    }
    __syncthreads();
    {
        f32x16 acc = zero16();
        const float* W = sW2 + (net * H + cb * 32) * LD;
        const float* Hin = sH1 + net * R * LD;
        mma32<H>(acc, [&](int i, int k) { return Hin[i * LD + k]; }, [&](int k, int j) { return W[j * LD + k]; });
        const int col = cb * 32 + (lane & 31);
        const float bias = sB2[net * H + col];
#pragma unroll
        for (int e = 0; e < 16; ++e) sH2[(net * R + acc_row(e, lane)) * LD + col] = tanh_fast(acc[e] + bias);
    }
    __syncthreads();
    {
        const float* W = sW3 + net * AP * LD;
        const float* Hin = sH2 + (net * R + cb * 16) * LD;
        const f32x4 acc = mma16<H>([&](int i, int k) { return Hin[i * LD + k]; },
                                   [&](int k, int j) { return W[j * LD + k]; });
        const int col = lane & 15;
        const float bias = sB3[net * AP + col];
#pragma unroll
        for (int e = 0; e < 4; ++e) sOut[(net * R + cb * 16 + 4 * (lane >> 4) + e) * LDO + col] = acc[e] + bias;
    }
    __syncthreads();
    if (a.noise && a.continuous) {
        // 8 lanes per row, action dims lj and lj + 8 per lane; the log-prob terms are formed as evaluate() forms them
        // and folded with DPP moves
        const int r = tid >> 3, lj = tid & 7, n = row0 + r;
        const bool live = n < a.N;
        const float* mu = sOut + r * LDO;
        float lp = 0.0f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = lj + 8 * q;
            if (live && k < A) {
                const float ls = sLs[k], sd = sSd[k], m = mu[k];
                const float act = m + sd * a.noise[(size_t)n * A + k];
                a.actions[(size_t)n * A + k] = act;
                const float z = act - m;                                   // as evaluate() forms it: (a - mu)
                lp += (-(z * z) / (2.0f * (sd * sd)) - ls) - 0.9189385332046727f;
            }
        }
        lp = sum8(lp);
        if (live && lj == 0) {
            a.logp[n] = lp;
            a.value[n] = sOut[(R + r) * LDO];
        }
        return;
    }
    if (tid < R && row0 + tid < a.N) {
        const int n = row0 + tid;
        const float* mu = sOut + tid * LDO;
        a.value[n] = sOut[(R + tid) * LDO];
        if (a.noise) {
            if (a.continuous) {
                float lp = 0.0f;
                for (int k = 0; k < A; ++k) {
                    const float ls = a.params[a.L.logstd + k];
                    const float eps = a.noise[(size_t)n * A + k];
                    const float act = mu[k] + expf(ls) * eps;
                    a.actions[(size_t)n * A + k] = act;
                    const float z = act - mu[k];                       // as evaluate() forms it: (a - mu)
                    const float sd = expf(ls);
                    lp += (-(z * z) / (2.0f * (sd * sd)) - ls) - 0.9189385332046727f;
                }
                a.logp[n] = lp;
            } else {
                float mx = mu[0];
                for (int k = 1; k < A; ++k) mx = fmaxf(mx, mu[k]);
                float se = 0.0f;
                for (int k = 0; k < A; ++k) se += expf(mu[k] - mx);
                const float lse = mx + logf(se);
                const float u = a.noise[n];
                float cdf = 0.0f;
                int pick = A - 1;
                for (int k = 0; k < A; ++k) {
                    cdf += expf(mu[k] - lse);
                    if (u < cdf) {
                        pick = k;
                        break;
                    }
                }
                a.actions[n] = (float)pick;
                a.logp[n] = mu[pick] - lse;
            }
        }
    }
}

constexpr size_t act_lds_bytes() {
    return sizeof(float) * (size_t)(R * LD + 2 * 2 * R * LD + 2 * 2 * H * LD + 2 * AP * LD + 2 * R * LDO + 4 * H + 2 * AP + 2 * AP);
}

// grads[p] = sum over slabs, fixed order (deterministic); block 0 also folds the loss scalars.
// 64 parameters x 16 slab groups per 1024-thread workgroup: every wave-instruction reads one coalesced
// 256-B slab row, 16 rows per parameter are in flight at once, groups are combined through LDS in order.
constexpr int kRedGroups = 16;
// sq_part != nullptr (chained minibatch step): also the clip's partial sums of squares, one per workgroup; step_dev != nullptr: the Adam
// step count is advanced and the tile counter of the next K7 launch is cleared.
template <int HALVES>   // 16 slab rows per thread and half: 256 slabs (K7, K7w) or 512 (K7w with two workgroups per CU)
__global__ __launch_bounds__(1024) void k_mlp_reduce(const float* __restrict__ slabs, const double* __restrict__ loss_part,
                                                     int n_slabs, int n_params, PpoHyper h, float* __restrict__ grads,
                                                     float* __restrict__ out_scalars, double* __restrict__ sq_part,
                                                     float* __restrict__ step_dev, unsigned* __restrict__ tile_counter,
                                                     double beta1, double beta2, double* __restrict__ bc_out) {
    __shared__ float s_part[kRedGroups][64];
    const int pi = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + pi;
    const bool stepper = step_dev && threadIdx.x == 0 && blockIdx.x == 0;
    const float t_new = stepper ? *step_dev + 1.0f : 0.0f;      // the Adam step this minibatch is
    float acc = 0.0f;
    if (p < n_params) {
        // n_slabs <= kMaxGrid = 16 groups x 16: a thread's rows are all fetched before the first add (one memory round
        // trip, not four); rows past n_slabs read as zero; the order of the adds is fixed
        static_assert(kMaxGrid <= 16 * kRedGroups, "k_mlp_reduce: 16 rows per thread cover the grid");
        float x[16 * HALVES];
#pragma unroll
        for (int j = 0; j < 16 * HALVES; ++j) {
            const int b = grp + j * kRedGroups;
            x[j] = b < n_slabs ? slabs[(size_t)b * n_params + p] : 0.0f;
        }
        if (stepper && bc_out) {
            // Adam's bias corrections for the optimizer launch that follows, while the slab rows are on their way
            bc_out[0] = 1.0 - pow(beta1, (double)t_new);
            bc_out[1] = 1.0 - pow(beta2, (double)t_new);
        }
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            float a4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a4[k] = ((x[16 * h + k] + x[16 * h + k + 4]) + x[16 * h + k + 8]) + x[16 * h + k + 12];
            const float ah = (a4[0] + a4[1]) + (a4[2] + a4[3]);
            acc = h == 0 ? ah : acc + ah;
        }
    }
    s_part[grp][pi] = acc;
    __syncthreads();
    if (grp == 0) {   // wave 0
        float t = 0.0f;
        if (p < n_params) {
#pragma unroll
            for (int g = 0; g < kRedGroups; ++g) t += s_part[g][pi];
            grads[p] = t;
        }
        if (sq_part) {
            const double q = wave_sum((double)t * (double)t);
            if (pi == 0) sq_part[blockIdx.x] = q;
        }
        if (stepper) {
            *step_dev = t_new;   // the Adam kernel (a later launch) reads the new step count
            tile_counter[0] = 0u;
            tile_counter[1] = 0u;
        }
    }
    __shared__ double r[6];
    if (blockIdx.x == 0 && threadIdx.x < 6 * kWave) {
        // one wave per loss quantity: lanes stride over the workgroups' partials, then a shuffle reduce
        const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
        double s = 0.0;
        for (int b = l; b < n_slabs; b += kWave) s += loss_part[(size_t)b * 8 + q];
        s = wave_sum(s);
        if (l == 0) r[q] = s;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double M = (double)h.M;
        const float pg = (float)(r[0] / M), vl = 0.5f * (float)(r[1] / M), ent = (float)(r[2] / M);
        out_scalars[AURPPO_S_PG] = pg;
        out_scalars[AURPPO_S_VL] = vl;
        out_scalars[AURPPO_S_ENT] = ent;
        out_scalars[AURPPO_S_OLD_KL] = (float)(r[3] / M);
        out_scalars[AURPPO_S_KL] = (float)(r[4] / M);
        out_scalars[AURPPO_S_CLIPFRAC] = (float)(r[5] / M);
        out_scalars[AURPPO_S_LOSS] = (pg - h.ent_coef * ent) + vl * h.vf_coef;
        out_scalars[AURPPO_S_ADV_MEAN] = (float)loss_part[6];
        out_scalars[AURPPO_S_ADV_STD] = (float)loss_part[7];
    }
}

// Where an updated weight is also kept in MFMA-operand order for the next K7 launch (W1 in k_mlp_step2's fp32 B-operand order;
// W1 / W2 as bf16 planes for k_mlp_step3).  Offsets past the bucket switch a copy off.
struct OperandCopies {
    int w1_actor, w1_critic, D;
    float* w1op;
    int w2_actor, w2_critic;
    unsigned short* wop3;
    aurppo_mlp::WideCopies wide;     // K7w's copies (wide.wop == nullptr: none)
};
__device__ __forceinline__ void refresh_operand_copies(const OperandCopies& oc, int i, float pn) {
    if (oc.wide.wop) {
        // K7w (k_mlpw_prep's layout, source-first): forward copy B[k][j] = W[j][k], backward copy (layers >= 1) B[k][j] = W[k][j];
        // entry ((nl * 2 + dir) * 16 + blk) * 1024 + lane * 16 + m
        const aurppo_mlp::WideCopies& wc = oc.wide;
        for (int n = 0; n < 2; ++n)
            for (int l = 0; l < wc.NL; ++l) {
                const int in_dim = l == 0 ? wc.D : wc.Hd;
                const int e = i - wc.w[n][l];
                if (e < 0 || e >= wc.Hd * in_dim) continue;
                const int row = e / in_dim, col = e - row * in_dim;
                const int nl = n * 3 + l;
                if (wc.wop3) {
                    // k_mlpw3_step's copies (mlp_wide.hip, w3::wop_index): slot (net, layer, direction) x column block x k-step x
                    // plane x lane x 8; forward B[k][n = out] = W[out][k], backward (layers >= 1) B[k = out][n = in] = W[out][in]
                    unsigned p0, p1, p2;
                    bf3::split3(pn, 0.0f, p0, p1, p2);
                    for (int dir = 0; dir < (l > 0 ? 2 : 1); ++dir) {
                        const int nn = dir == 0 ? row : col, kk = dir == 0 ? col : row;
                        const int at = (nl * 2 + dir) * (4 * 8 * 3 * 512) + (nn >> 5) * (8 * 3 * 512) +
                                       (((kk >> 4) * 3) * 64 + (nn & 31) + 32 * ((kk >> 3) & 1)) * 8 + (kk & 7);
                        wc.wop3[at] = (unsigned short)p0;
                        wc.wop3[at + 512] = (unsigned short)p1;
                        wc.wop3[at + 1024] = (unsigned short)p2;
                    }
                    continue;
                }
                {
                    const int blk = (row >> 5) * 4 + (col >> 5), lane = (row & 31) + 32 * (col & 1), m = (col & 31) >> 1;
                    wc.wop[(((nl * 2 + 0) * 16 + blk) * 64 + lane) * 16 + m] = pn;
                }
                if (l > 0) {
                    const int blk = (col >> 5) * 4 + (row >> 5), lane = (col & 31) + 32 * (row & 1), m = (row & 31) >> 1;
                    wc.wop[(((nl * 2 + 1) * 16 + blk) * 64 + lane) * 16 + m] = pn;
                }
            }
        return;
    }
    const int D = oc.D;
    const int ea = i - oc.w1_actor, ec = i - oc.w1_critic;
    const int e = (ea >= 0 && ea < H * D) ? ea : ((ec >= 0 && ec < H * D) ? ec : -1);
    if (e >= 0 && oc.w1op) {
        const int net = (ea >= 0 && ea < H * D) ? 0 : 1;
        const int row = e / D, k = e - row * D;
        oc.w1op[((net * 2 + (row >> 5)) * 32 + (k >> 1)) * kWave + (row & 31) + 32 * (k & 1)] = pn;
    }
    if (oc.wop3) {      // k_mlp_step3's copies: the bf16 planes of the new value wherever the weight appears as an operand
        const int e2a = i - oc.w2_actor, e2c = i - oc.w2_critic;
        const int e2 = (e2a >= 0 && e2a < H * H) ? e2a : ((e2c >= 0 && e2c < H * H) ? e2c : -1);
        if (e >= 0 || e2 >= 0) {
            const bool is_w2 = e < 0;
            const int net = is_w2 ? ((e2a >= 0 && e2a < H * H) ? 0 : 1) : ((ea >= 0 && ea < H * D) ? 0 : 1);
            const int row = is_w2 ? e2 / H : e / D;
            const int col = is_w2 ? e2 % H : e - (e / D) * D;
            int at[2];
            const int n_at = bf3::wop3_places(net, is_w2 ? 1 : 0, row, col, at);
            unsigned p0, p1, p2;
            bf3::split3(pn, 0.0f, p0, p1, p2);
            for (int q = 0; q < n_at; ++q) {
                oc.wop3[at[q]] = (unsigned short)p0;
                oc.wop3[at[q] + bf3::kWopBlock] = (unsigned short)p1;
                oc.wop3[at[q] + 2 * bf3::kWopBlock] = (unsigned short)p2;
            }
        }
    }
}

// Chained minibatch step, third launch: global-norm clip + Adam over the bucket (K6b's arithmetic); the W1
// elements it has just updated are dropped into the operand-order copy the next K7 launch streams; and the
// workgroups past `nb_upd` form the next minibatch's advantage partial sums -- so nothing is left to prepare
// before that launch.
__global__ __launch_bounds__(kThreads) void k_adam_chain(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, int n, const double* __restrict__ part,
                                                         int n_part, float max_norm, const float* __restrict__ lr_dev,
                                                         const float* __restrict__ step, double beta1, double beta2,
                                                         double eps, float* __restrict__ out_norm, float gscale, int nb_upd,
                                                         OperandCopies oc, const float4* __restrict__ rec, int rec_stride,
                                                         const int32_t* __restrict__ next_idx, int next_M,
                                                         double (*__restrict__ stats)[2], const double* __restrict__ bc) {
    __shared__ double sc[2][kThreads / kWave];
    __shared__ float s_coef;
    if ((int)blockIdx.x < nb_upd) {
        __shared__ double s_own;
        // a thread's first four elements are fetched before the clip preamble, whose partial sums, block reductions
        // and pow() calls they then overlap
        constexpr int kPre = 4;
        float p0[kPre], g0[kPre], m0[kPre], v0[kPre];
#pragma unroll
        for (int k = 0; k < kPre; ++k) {
            const int i = (blockIdx.x + k * nb_upd) * kThreads + threadIdx.x;
            const bool ok = i < n;
            p0[k] = ok ? p[i] : 0.0f;
            g0[k] = ok ? g[i] : 0.0f;
            m0[k] = ok ? m[i] : 0.0f;
            v0[k] = ok ? v[i] : 0.0f;
        }
        if (!part) {
            // no partial sums were left by k_mlp_reduce (the gradient went through an all-reduce since): every
            // workgroup forms the norm of the scaled gradient itself, in the same order -- 17 k floats, L2-resident
            // 16-B loads, all of a thread's in flight before the first add (n = 17 k: 17 per thread)
            double q = 0.0;
            const int n4 = (reinterpret_cast<uintptr_t>(g) & 15) == 0 ? n / 4 : 0;
            const float4* g4 = reinterpret_cast<const float4*>(g);
            constexpr int kU = 8;
            for (int i0 = threadIdx.x; i0 < n4; i0 += kU * kThreads) {
                float4 x[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int i = i0 + u * kThreads;
                    x[u] = i < n4 ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const double a0 = (double)(x[u].x * gscale), a1 = (double)(x[u].y * gscale);
                    const double a2 = (double)(x[u].z * gscale), a3 = (double)(x[u].w * gscale);
                    q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                }
            }
            for (int i = 4 * n4 + threadIdx.x; i < n; i += kThreads) {
                const double x = (double)(g[i] * gscale);
                q += x * x;
            }
            const double t = block_sum<kThreads / kWave>(q, sc[1]);
            if (threadIdx.x == 0) s_own = t;
            __syncthreads();
        }
        AdamScalars a = adam_scalars<kThreads / kWave>(part ? part : &s_own, part ? n_part : 1, max_norm, lr_dev, step, beta1,
                                                       beta2, eps, out_norm, blockIdx.x == 0, sc[0], &s_coef, bc);
        a.gscale = gscale;
        int k = 0;
        for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += nb_upd * kThreads, ++k) {
            // in-kernel norm: g is still being read by the other workgroups, so the clipped value is not stored back
            float pn;
            if (k < kPre) {
                float pk = p0[0], gk = g0[0], mk = m0[0], vk = v0[0];
#pragma unroll
                for (int j = 1; j < kPre; ++j)
                    if (k == j) { pk = p0[j]; gk = g0[j]; mk = m0[j]; vk = v0[j]; }
                pn = adam_update_pre(p, g, m, v, i, pk, gk, mk, vk, a, part != nullptr);
            } else {
                pn = adam_update(p, g, m, v, i, true, a, part != nullptr);
            }
            refresh_operand_copies(oc, i, pn);
        }
    } else {
        const int nsb = gridDim.x - nb_upd, b = blockIdx.x - nb_upd;
        double s = 0.0, q = 0.0;
        adv_partial_sums(rec, rec_stride, next_idx, next_M, b * kThreads + threadIdx.x, nsb * kThreads, s, q);
        const double bs = block_sum<kThreads / kWave>(s, sc[0]);
        const double bq = block_sum<kThreads / kWave>(q, sc[1]);
        if (threadIdx.x == 0) {
            stats[b][0] = bs;
            stats[b][1] = bq;
        }
    }
}

}  // namespace

extern "C" size_t aurppo_mlp_workspace_bytes(int n_params) {
    return sizeof(double) * 2 * kStatBlocks + sizeof(double) * 8 * kMaxGrid + sizeof(float) * (size_t)kMaxGrid * (size_t)n_params + 64 +
           sizeof(unsigned long long) * 44 * kMaxGrid + sizeof(float) * 4 * 32 * 64 + 64 +
           ((sizeof(double) * (size_t)((n_params + 63) / 64) + 63) / 64) * 64 + 64 +
           mlp_step3_wop_bytes() + 64;
}

namespace {
struct ChainArgs {   // the optimizer half of aurppo_mlp_ppo_minibatch_f32
    float* params_rw;
    float* exp_avg;
    float* exp_avg_sq;
    double max_norm;
    const float* lr_dev;
    float* step_dev;
    double beta1, beta2, eps;
    float* out_norm;
    const int32_t* next_idx;
    int next_M;
    int chained;
    int grad_only;   // stop after k_mlp_reduce (the caller all-reduces the gradient, then calls the apply half)
};
struct WsView {   // carve-up of the caller's workspace (aurppo_mlp_workspace_bytes)
    double* stats;               // (kStatBlocks, 2) advantage partial sums of the prepared minibatch
    double* loss_part;           // (kMaxGrid, 8)
    float* slabs;                // (kMaxGrid, n_params)
    unsigned long long* stamps;  // diagnostic build
    float* w1op;                 // W1 in B-operand order
    unsigned* tile_counter;
    double* sq_part;             // clip partial sums, one per k_mlp_reduce workgroup
    unsigned short* wop3;        // k_mlp_step3: bf16 planes of W1 / W2 in operand order
};
WsView ws_view(void* workspace, int n_params) {
    WsView v;
    char* w = reinterpret_cast<char*>(workspace);
    v.stats = reinterpret_cast<double*>(w);
    v.loss_part = v.stats + 2 * kStatBlocks;
    v.slabs = reinterpret_cast<float*>(v.loss_part + 8 * kMaxGrid);
    v.stamps = reinterpret_cast<unsigned long long*>(w + ((sizeof(double) * (2 * kStatBlocks + 8 * kMaxGrid) +
                                                           sizeof(float) * (size_t)kMaxGrid * (size_t)n_params + 63) / 64) * 64);
    v.w1op = reinterpret_cast<float*>(v.stamps + 44 * kMaxGrid);
    v.tile_counter = reinterpret_cast<unsigned*>(v.w1op + 4 * 32 * 64);
    v.sq_part = reinterpret_cast<double*>(v.tile_counter + 16);
    v.wop3 = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(v.sq_part) +
                                               ((sizeof(double) * (size_t)((n_params + 63) / 64) + 63) / 64) * 64);
    return v;
}
int stat_blocks_for(int M) {
    int sb = (M + kThreads * 4 - 1) / (kThreads * 4);
    return sb > kStatBlocks ? kStatBlocks : sb;
}
}  // namespace

static int mlp_step_impl(const float* obs, const float* actions, const float* rec, const int32_t* idx, int M, int D,
                         int A, int continuous, int hidden, const float* params, const int* layout_h, int n_params, float* grads,
                         double clip, double ent_coef, double vf_coef, int norm_adv, int vloss_mode, float* out_scalars,
                         void* workspace, void* stream, void* ev_begin, void* ev_end, const ChainArgs* chain = nullptr) {
    AURPPO_REQUIRE(obs && rec && idx && params && layout_h && grads && out_scalars && workspace, AURPPO_EINVAL,
                   "aurppo_mlp_ppo_step_f32: null pointer");
    // actions == NULL: packed records -- rec is (B, 16), floats 4.. of a record hold the sample's action row
    AURPPO_REQUIRE(actions || (continuous ? A : 1) <= 12, AURPPO_ESHAPE,
                   "aurppo_mlp_ppo_step_f32: packed records hold at most 12 action floats (action_dim=%d)", A);
    AURPPO_REQUIRE(hidden == H, AURPPO_ESHAPE, "aurppo_mlp_ppo_step_f32: hidden_dim=%d (only %d is built)", hidden, H);
    AURPPO_REQUIRE(D >= 1 && D <= H, AURPPO_ESHAPE, "aurppo_mlp_ppo_step_f32: state_dim=%d must be 1..%d", D, H);
    AURPPO_REQUIRE(A >= 1 && A <= AP && (continuous || A >= 2), AURPPO_ESHAPE,
                   "aurppo_mlp_ppo_step_f32: action_dim=%d must be 1..%d (>= 2 logits for a Categorical head)", A, AP);
    AURPPO_REQUIRE(M > 0 && n_params > 0, AURPPO_ESHAPE, "aurppo_mlp_ppo_step_f32: M=%d n_params=%d", M, n_params);
    AURPPO_REQUIRE(vloss_mode >= 0 && vloss_mode <= 2, AURPPO_EINVAL, "aurppo_mlp_ppo_step_f32: bad vloss_mode %d", vloss_mode);
    AURPPO_REQUIRE(aligned_to(workspace, 16) && aligned_to(rec, 16), AURPPO_EINVAL,
                   "aurppo_mlp_ppo_step_f32: workspace / rec not 16-byte aligned");
    MlpArgs a;
    a.obs = obs; a.actions = actions; a.rec = reinterpret_cast<const float4*>(rec); a.idx = idx; a.params = params;
    a.rec_stride = actions ? 1 : 4;
    a.D = D; a.A = A;
    a.continuous = continuous ? 1 : 0;
    // layout_h: w1a,b1a,w2a,b2a,w3a,b3a, w1c,b1c,w2c,b2c,w3c,b3c, logstd
    for (int n = 0; n < 2; ++n) {
        a.L.w1[n] = layout_h[6 * n + 0]; a.L.b1[n] = layout_h[6 * n + 1]; a.L.w2[n] = layout_h[6 * n + 2];
        a.L.b2[n] = layout_h[6 * n + 3]; a.L.w3[n] = layout_h[6 * n + 4]; a.L.b3[n] = layout_h[6 * n + 5];
    }
    a.L.logstd = continuous ? layout_h[12] : 0;
    a.L.n_params = n_params;
    for (int k = 0; k < (continuous ? 13 : 12); ++k)
        AURPPO_REQUIRE(layout_h[k] >= 0 && layout_h[k] < n_params, AURPPO_ESHAPE, "aurppo_mlp_ppo_step_f32: layout[%d]=%d", k, layout_h[k]);
    a.h = make_hyper(M, clip, ent_coef, vf_coef, norm_adv, vloss_mode);
    const WsView wv = ws_view(workspace, n_params);
    double* stats = wv.stats;
    a.stats = wv.stats;
    a.loss_part = wv.loss_part;
    a.slabs = wv.slabs;
    a.stamps = wv.stamps;
    hipStream_t s = (hipStream_t)stream;
    const int sb = stat_blocks_for(M);
    a.n_stat_blocks = sb;
    const AurppoKnobs& knobs = aurppo_knobs();
    // 2: k_mlp_step2 (f32 MFMA); 3: k_mlp_step3 (3 x bf16-split MFMA); normalised in api.hip::parse_knobs
    const int variant = knobs.k7_variant;
    a.w1op = wv.w1op;
    a.tile_counter = wv.tile_counter;
    a.static_tiles = knobs.static_tiles ? 1 : 0;
    a.wop3 = wv.wop3;
    double* sq_part = wv.sq_part;
    if (!(chain && chain->chained)) {   // otherwise the previous chained call has prepared all of this
        hipLaunchKernelGGL(k_adv_stats_idx, dim3(sb), dim3(kThreads), 0, s, a.rec, a.rec_stride, idx, M,
                           reinterpret_cast<double (*)[2]>(stats), params, a.L.w1[0], a.L.w1[1], D, a.w1op, a.tile_counter);
        AURPPO_LAUNCH_CHECK("k_adv_stats_idx");
        if (variant == 3) {
            const int rc = launch_mlp3_prep(params, a.L, D, wv.wop3, s);
            if (rc != AURPPO_OK) return rc;
        }
    }
    const int n_tiles = (M + R - 1) / R;
    // One persistent workgroup per CU, minus one CU per XCD (AURPPO_MLP_SPARE_CUS, default 8; workgroups are dealt
    // round-robin over the 8 XCDs): the single-workgroup shuffle kernels of the side stream then have a CU of
    // their own.  The kernel hands tiles out dynamically, but it fills a CU's register file, so a workgroup whose CU
    // is taken starts late; 8 spare CUs measured 3.86-3.93 ms per update against 4.09-4.10 ms with none.
    static int cus_of[kMaxDevices] = {0};
    const int dslot = aurppo_device_slot();
    if (!cus_of[dslot]) {
        hipDeviceProp_t prop;
        AURPPO_HIP_TRY(hipGetDeviceProperties(&prop, dslot));
        cus_of[dslot] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : kMaxGrid;
    }
    const int cus = cus_of[dslot];
    const int spare = knobs.k7_spare_cus >= 0 ? knobs.k7_spare_cus : 8;
    int grid = cus - spare;
    if (grid > kMaxGrid) grid = kMaxGrid;
    if (grid < 1) grid = 1;
    // A minibatch that every CU could take in ONE round of tiles (two per workgroup) but the spare-CU grid could not -- BASELINE
    // config 4's shard, 512 envs x 128 steps / 4 = 512 tiles against 2 x 248 -- gets all the CUs: the stragglers' second round
    // was a quarter of that launch (stamps at M = 16 384: tile loop 12.0 us median, 22.2 us for the sets that drew a second tile)
    if (n_tiles > 2 * grid && n_tiles <= 2 * cus && cus <= kMaxGrid) grid = cus;
    if (grid > (n_tiles + 1) / 2) grid = (n_tiles + 1) / 2;     // two tile sets per workgroup
    if (ev_begin) AURPPO_HIP_TRY(hipEventRecord((hipEvent_t)ev_begin, s));
    {
        const int rc = variant == 3 ? launch_mlp_step3(a, grid, s) : launch_mlp_step2(a, grid, s);
        if (rc != AURPPO_OK) return rc;
    }
    if (ev_end) AURPPO_HIP_TRY(hipEventRecord((hipEvent_t)ev_end, s));
    const int n_red = (n_params + 63) / 64;
    // (words 8..11 of the counter block: Adam's two bias corrections, formed by the reduce for the optimizer launch behind it)
    double* const bc = (chain && !chain->grad_only) ? reinterpret_cast<double*>(a.tile_counter + 8) : nullptr;
    hipLaunchKernelGGL(k_mlp_reduce<1>, dim3(n_red), dim3(1024), 0, s, a.slabs, a.loss_part, grid, n_params, a.h, grads,
                       out_scalars, (chain && !chain->grad_only) ? sq_part : nullptr, chain ? chain->step_dev : nullptr,
                       a.tile_counter, bc ? chain->beta1 : 0.0, bc ? chain->beta2 : 0.0, bc);
    AURPPO_LAUNCH_CHECK("k_mlp_reduce");
    if (chain && !chain->grad_only) {
        int nb_upd = (n_params + kThreads * 4 - 1) / (kThreads * 4);
        if (nb_upd > 64) nb_upd = 64;
        const int nsb = chain->next_idx ? stat_blocks_for(chain->next_M) : 0;
        OperandCopies oc = {a.L.w1[0], a.L.w1[1], D, a.w1op, a.L.w2[0], a.L.w2[1],
                            variant == 3 ? wv.wop3 : (unsigned short*)nullptr, {}};
        oc.wide.wop = nullptr;
        hipLaunchKernelGGL(k_adam_chain, dim3(nb_upd + nsb), dim3(kThreads), 0, s, chain->params_rw, grads, chain->exp_avg,
                           chain->exp_avg_sq, n_params, sq_part, n_red, (float)chain->max_norm, chain->lr_dev,
                           chain->step_dev, chain->beta1, chain->beta2, chain->eps, chain->out_norm, 1.0f, nb_upd, oc,
                           a.rec, a.rec_stride, chain->next_idx, chain->next_M, reinterpret_cast<double (*)[2]>(stats), bc);
        AURPPO_LAUNCH_CHECK("k_adam_chain");
    }
    return AURPPO_OK;
}

namespace {
OperandCopies no_operand_copies(int n_params) {   // every offset past the bucket: no copy is refreshed
    OperandCopies oc = {n_params, n_params, 1, nullptr, n_params, n_params, nullptr, {}};
    oc.wide.wop = nullptr;
    return oc;
}
}  // namespace

int aurppo_mlp::launch_mlp_reduce(const float* slabs, const double* loss_part, int n_slabs, int n_params, const PpoHyper& h,
                                  float* grads, float* out_scalars, hipStream_t s, double* sq_part, float* step_dev,
                                  unsigned* unused_counter, double beta1, double beta2, double* bc_out) {
    // (k_mlp_reduce clears K7's tile counter when it advances the step: the caller names a word it may clear instead)
    AURPPO_REQUIRE(!step_dev || unused_counter, AURPPO_EINVAL, "launch_mlp_reduce: step_dev without a scratch counter");
    AURPPO_REQUIRE(n_slabs >= 1 && n_slabs <= 2 * kMaxGrid, AURPPO_ESHAPE, "launch_mlp_reduce: n_slabs=%d", n_slabs);
    if (n_slabs <= kMaxGrid)
        hipLaunchKernelGGL(k_mlp_reduce<1>, dim3((n_params + 63) / 64), dim3(1024), 0, s, slabs, loss_part, n_slabs, n_params, h,
                           grads, out_scalars, sq_part, step_dev, unused_counter, beta1, beta2, bc_out);
    else
        hipLaunchKernelGGL(k_mlp_reduce<2>, dim3((n_params + 63) / 64), dim3(1024), 0, s, slabs, loss_part, n_slabs, n_params, h,
                           grads, out_scalars, sq_part, step_dev, unused_counter, beta1, beta2, bc_out);
    AURPPO_LAUNCH_CHECK("k_mlp_reduce");
    return AURPPO_OK;
}

int aurppo_mlp::launch_adam_tail(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int n_params,
                                 const double* sq_part, double max_norm, const float* lr_dev, const float* step_dev,
                                 double beta1, double beta2, double eps, float* out_norm, hipStream_t s, const WideCopies* wide,
                                 const float4* rec, int rec_stride, const int32_t* next_idx, int next_M, double* stats,
                                 const double* bc) {
    int nb_upd = (n_params + kThreads * 4 - 1) / (kThreads * 4);
    if (nb_upd > 64) nb_upd = 64;
    // no K7 operand copies (offsets past the bucket); K7w's if the caller names them; statistics of a next minibatch if it names one
    OperandCopies oc = no_operand_copies(n_params);
    if (wide) oc.wide = *wide;
    const int nsb = (next_idx && stats) ? stat_blocks_for(next_M) : 0;
    hipLaunchKernelGGL(k_adam_chain, dim3(nb_upd + nsb), dim3(kThreads), 0, s, params, grads, exp_avg, exp_avg_sq, n_params, sq_part,
                       (n_params + 63) / 64, (float)max_norm, lr_dev, step_dev, beta1, beta2, eps, out_norm, 1.0f, nb_upd, oc, rec,
                       rec_stride, nsb ? next_idx : (const int32_t*)nullptr, nsb ? next_M : 0, reinterpret_cast<double (*)[2]>(stats), bc);
    AURPPO_LAUNCH_CHECK("k_adam_chain");
    return AURPPO_OK;
}

extern "C" int aurppo_mlp_ppo_step_f32(const float* obs, const float* actions, const float* rec, const int32_t* idx,
                                       int M, int D, int A, int continuous, int hidden, const float* params, const int* layout_h,
                                       int n_params, float* grads, double clip, double ent_coef, double vf_coef,
                                       int norm_adv, int vloss_mode, float* out_scalars, void* workspace, void* stream) {
    return mlp_step_impl(obs, actions, rec, idx, M, D, A, continuous, hidden, params, layout_h, n_params, grads, clip, ent_coef,
                         vf_coef, norm_adv, vloss_mode, out_scalars, workspace, stream, nullptr, nullptr);
}

extern "C" int aurppo_mlp_ppo_step_ev_f32(const float* obs, const float* actions, const float* rec, const int32_t* idx,
                                          int M, int D, int A, int continuous, int hidden, const float* params, const int* layout_h,
                                          int n_params, float* grads, double clip, double ent_coef, double vf_coef,
                                          int norm_adv, int vloss_mode, float* out_scalars, void* workspace,
                                          void* stream, void* ev_begin, void* ev_end) {
    return mlp_step_impl(obs, actions, rec, idx, M, D, A, continuous, hidden, params, layout_h, n_params, grads, clip, ent_coef,
                         vf_coef, norm_adv, vloss_mode, out_scalars, workspace, stream, ev_begin, ev_end);
}

extern "C" int aurppo_mlp_ppo_minibatch_f32(const float* obs, const float* actions, const float* rec, const int32_t* idx, int M,
                                            int D, int A, int continuous, int hidden, float* params, const int* layout_h,
                                            int n_params, float* grads, double clip, double ent_coef, double vf_coef,
                                            int norm_adv, int vloss_mode, float* out_scalars, float* exp_avg,
                                            float* exp_avg_sq, double max_norm, const float* lr_dev, float* step_dev,
                                            double beta1, double beta2, double eps, float* out_norm, const int32_t* next_idx,
                                            int next_M, int chained, void* workspace, void* stream) {
    AURPPO_REQUIRE(exp_avg && exp_avg_sq && lr_dev && step_dev && out_norm, AURPPO_EINVAL,
                   "aurppo_mlp_ppo_minibatch_f32: null optimizer pointer");
    AURPPO_REQUIRE(!next_idx || next_M > 0, AURPPO_ESHAPE, "aurppo_mlp_ppo_minibatch_f32: next_M=%d", next_M);
    ChainArgs c;
    c.params_rw = params; c.exp_avg = exp_avg; c.exp_avg_sq = exp_avg_sq; c.max_norm = max_norm; c.lr_dev = lr_dev;
    c.step_dev = step_dev; c.beta1 = beta1; c.beta2 = beta2; c.eps = eps; c.out_norm = out_norm;
    c.next_idx = next_idx; c.next_M = next_idx ? next_M : 0; c.chained = chained ? 1 : 0; c.grad_only = 0;
    return mlp_step_impl(obs, actions, rec, idx, M, D, A, continuous, hidden, params, layout_h, n_params, grads, clip, ent_coef,
                         vf_coef, norm_adv, vloss_mode, out_scalars, workspace, stream, nullptr, nullptr, &c);
}

extern "C" int aurppo_mlp_ppo_grad_f32(const float* obs, const float* actions, const float* rec, const int32_t* idx, int M, int D,
                                       int A, int continuous, int hidden, const float* params, const int* layout_h, int n_params,
                                       float* grads, double clip, double ent_coef, double vf_coef, int norm_adv, int vloss_mode,
                                       float* out_scalars, float* step_dev, int chained, void* workspace, void* stream) {
    AURPPO_REQUIRE(step_dev, AURPPO_EINVAL, "aurppo_mlp_ppo_grad_f32: null step_dev");
    ChainArgs c = {};
    c.step_dev = step_dev; c.chained = chained ? 1 : 0; c.grad_only = 1;
    return mlp_step_impl(obs, actions, rec, idx, M, D, A, continuous, hidden, params, layout_h, n_params, grads, clip, ent_coef,
                         vf_coef, norm_adv, vloss_mode, out_scalars, workspace, stream, nullptr, nullptr, &c);
}

static int mlp_apply_impl(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const int* layout_h, int n_params, int D,
                          double grad_scale, const double* sq_part, int n_part, double max_norm, const float* lr_dev,
                          const float* step_dev, double beta1, double beta2, double eps, float* out_norm, const float* rec,
                          int rec_floats, const int32_t* next_idx, int next_M, void* workspace, void* stream) {
    AURPPO_REQUIRE(params && grads && exp_avg && exp_avg_sq && layout_h && lr_dev && step_dev && out_norm && workspace,
                   AURPPO_EINVAL, "aurppo_mlp_ppo_apply_f32: null pointer");
    AURPPO_REQUIRE(n_params > 0 && D >= 1 && D <= H, AURPPO_ESHAPE, "aurppo_mlp_ppo_apply_f32: n_params=%d D=%d",
                   n_params, D);
    AURPPO_REQUIRE(!next_idx || (next_M > 0 && rec && (rec_floats == 4 || rec_floats == 16)), AURPPO_ESHAPE,
                   "aurppo_mlp_ppo_apply_f32: next_M=%d rec_floats=%d", next_M, rec_floats);
    AURPPO_REQUIRE(aligned_to(workspace, 16) && (!rec || aligned_to(rec, 16)), AURPPO_EINVAL,
                   "aurppo_mlp_ppo_apply_f32: workspace / rec not 16-byte aligned");
    for (int k = 0; k < 12; ++k)
        AURPPO_REQUIRE(layout_h[k] >= 0 && layout_h[k] < n_params, AURPPO_ESHAPE, "aurppo_mlp_ppo_apply_f32: layout[%d]=%d", k,
                       layout_h[k]);
    const WsView wv = ws_view(workspace, n_params);
    double* stats = wv.stats;
    float* w1op = wv.w1op;
    int nb_upd = (n_params + kThreads * 4 - 1) / (kThreads * 4);
    if (nb_upd > 64) nb_upd = 64;
    const int nsb = next_idx ? stat_blocks_for(next_M) : 0;
    OperandCopies oc_apply = {layout_h[0], layout_h[6], D, w1op, layout_h[2], layout_h[8],
                              aurppo_knobs().k7_variant == 3 ? wv.wop3 : (unsigned short*)nullptr, {}};
    oc_apply.wide.wop = nullptr;
    hipLaunchKernelGGL(k_adam_chain, dim3(nb_upd + nsb), dim3(kThreads), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, n_params, sq_part, sq_part ? n_part : 0, (float)max_norm, lr_dev, step_dev, beta1, beta2, eps,
                       out_norm, (float)grad_scale, nb_upd, oc_apply,
                       reinterpret_cast<const float4*>(rec), rec_floats == 16 ? 4 : 1, next_idx, next_idx ? next_M : 0,
                       reinterpret_cast<double (*)[2]>(stats), (const double*)nullptr);
    AURPPO_LAUNCH_CHECK("k_adam_chain");
    return AURPPO_OK;
}

extern "C" int aurppo_mlp_ppo_apply_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const int* layout_h,
                                        int n_params, int D, double grad_scale, double max_norm, const float* lr_dev,
                                        const float* step_dev, double beta1, double beta2, double eps, float* out_norm,
                                        const float* rec, int rec_floats, const int32_t* next_idx, int next_M,
                                        void* workspace, void* stream) {
    return mlp_apply_impl(params, grads, exp_avg, exp_avg_sq, layout_h, n_params, D, grad_scale, nullptr, 0, max_norm, lr_dev, step_dev,
                          beta1, beta2, eps, out_norm, rec, rec_floats, next_idx, next_M, workspace, stream);
}

extern "C" int aurppo_mlp_ppo_apply_parts_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const int* layout_h,
                                              int n_params, int D, const double* sq_part, int n_part, double max_norm,
                                              const float* lr_dev, const float* step_dev, double beta1, double beta2, double eps,
                                              float* out_norm, const float* rec, int rec_floats, const int32_t* next_idx,
                                              int next_M, void* workspace, void* stream) {
    AURPPO_REQUIRE(sq_part && n_part > 0, AURPPO_EINVAL, "aurppo_mlp_ppo_apply_parts_f32: no partial sums");
    return mlp_apply_impl(params, grads, exp_avg, exp_avg_sq, layout_h, n_params, D, 1.0, sq_part, n_part, max_norm, lr_dev, step_dev,
                          beta1, beta2, eps, out_norm, rec, rec_floats, next_idx, next_M, workspace, stream);
}

namespace {
// rec64[b] = {rec4[b], actions[b][0 .. AW), 0 ...}: one 64-B line per sample for K7's record + action fetch
__global__ __launch_bounds__(256) void k_pack_rec64(const float4* __restrict__ rec4, const float* __restrict__ actions, int B,
                                                    int AW, float4* __restrict__ rec64) {
    const int i = blockIdx.x * 64 + (threadIdx.x >> 2), q = threadIdx.x & 3;   // 4 lanes per sample, one float4 each
    if (i >= B) return;
    float4 v;
    if (q == 0) {
        v = rec4[i];
    } else {
        const int k0 = 4 * (q - 1);
        const float* ar = actions + (size_t)i * AW;
        v.x = k0 + 0 < AW ? ar[k0 + 0] : 0.0f;
        v.y = k0 + 1 < AW ? ar[k0 + 1] : 0.0f;
        v.z = k0 + 2 < AW ? ar[k0 + 2] : 0.0f;
        v.w = k0 + 3 < AW ? ar[k0 + 3] : 0.0f;
    }
    rec64[(size_t)i * 4 + q] = v;
}
}  // namespace

extern "C" int aurppo_pack_records_f32(const float* rec4, const float* actions, int B, int action_floats, float* rec64,
                                       void* stream) {
    AURPPO_REQUIRE(rec4 && actions && rec64, AURPPO_EINVAL, "aurppo_pack_records_f32: null pointer");
    AURPPO_REQUIRE(B > 0 && action_floats >= 1 && action_floats <= 12, AURPPO_ESHAPE,
                   "aurppo_pack_records_f32: B=%d action_floats=%d (1..12)", B, action_floats);
    AURPPO_REQUIRE(aligned_to(rec4, 16) && aligned_to(rec64, 16), AURPPO_EINVAL, "aurppo_pack_records_f32: records not 16-byte aligned");
    hipLaunchKernelGGL(k_pack_rec64, dim3((B + 63) / 64), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(rec4),
                       actions, B, action_floats, reinterpret_cast<float4*>(rec64));
    AURPPO_LAUNCH_CHECK("k_pack_rec64");
    return AURPPO_OK;
}

extern "C" int aurppo_mlp_act_f32(const float* obs, const float* noise, int N, int D, int A, int continuous, int hidden,
                                  const float* params, const int* layout_h, int n_params, float* actions, float* logp,
                                  float* value, void* stream) {
    AURPPO_REQUIRE(obs && params && layout_h && value, AURPPO_EINVAL, "aurppo_mlp_act_f32: null pointer");
    AURPPO_REQUIRE(!noise || (actions && logp), AURPPO_EINVAL, "aurppo_mlp_act_f32: sampling needs actions and logp outputs");
    AURPPO_REQUIRE(hidden == H, AURPPO_ESHAPE, "aurppo_mlp_act_f32: hidden_dim=%d (only %d is built)", hidden, H);
    AURPPO_REQUIRE(D >= 1 && D <= H, AURPPO_ESHAPE, "aurppo_mlp_act_f32: state_dim=%d must be 1..%d", D, H);
    AURPPO_REQUIRE(A >= 1 && A <= AP && (continuous || A >= 2), AURPPO_ESHAPE, "aurppo_mlp_act_f32: action_dim=%d", A);
    AURPPO_REQUIRE(N > 0 && n_params > 0, AURPPO_ESHAPE, "aurppo_mlp_act_f32: N=%d n_params=%d", N, n_params);
    ActArgs a;
    a.obs = obs; a.noise = noise; a.params = params; a.actions = actions; a.logp = logp; a.value = value;
    a.N = N; a.D = D; a.A = A; a.continuous = continuous ? 1 : 0;
    for (int n = 0; n < 2; ++n) {
        a.L.w1[n] = layout_h[6 * n + 0]; a.L.b1[n] = layout_h[6 * n + 1]; a.L.w2[n] = layout_h[6 * n + 2];
        a.L.b2[n] = layout_h[6 * n + 3]; a.L.w3[n] = layout_h[6 * n + 4]; a.L.b3[n] = layout_h[6 * n + 5];
    }
    a.L.logstd = continuous ? layout_h[12] : 0;
    a.L.n_params = n_params;
    for (int k = 0; k < (continuous ? 13 : 12); ++k)
        AURPPO_REQUIRE(layout_h[k] >= 0 && layout_h[k] < n_params, AURPPO_ESHAPE, "aurppo_mlp_act_f32: layout[%d]=%d", k, layout_h[k]);
    static bool attr_set[kMaxDevices] = {false};
    const int dslot = aurppo_device_slot();
    if (!attr_set[dslot]) {
        AURPPO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_act),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)act_lds_bytes()));
        attr_set[dslot] = true;
    }
    hipLaunchKernelGGL(k_mlp_act, dim3((N + R - 1) / R), dim3(kThreads), act_lds_bytes(), (hipStream_t)stream, a);
    AURPPO_LAUNCH_CHECK("k_mlp_act");
    return AURPPO_OK;
}
