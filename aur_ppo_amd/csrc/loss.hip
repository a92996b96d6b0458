// K4 + K5: minibatch advantage normalisation and the clipped-surrogate PPO loss, forward AND
// backward in one pass  (src/ppo.py:225-264; src/robot_ppo.py:345-398).
//
// Three launches per minibatch, none of which touches the host:
//   k_adv_stats   : sum / sum-of-squares of the advantage slice (fp64 partials, fixed-order combine)
//   k_loss        : every workgroup re-derives mean/std from the partials, then one streaming pass
//                   reads the 7 per-sample inputs once, writes d loss/d newlogp and d loss/d newv,
//                   and leaves 6 fp64 partial sums per workgroup
//   k_loss_final  : one workgroup folds the partials into the 9 output scalars
// A launch boundary (~1.5 us) is cheaper on this part than an in-kernel agent-scope fence pair,
// which is why the combine steps are kernels and not "last block done" epilogues.
// Reductions are wave-shuffle -> LDS -> fixed-order, so results are run-to-run deterministic.
// HBM traffic: 28 B read + 8 B written per sample (+4 B for the statistics pass, L2-resident).
#include "ppo_math.h"

namespace {

constexpr int kMaxBlocks = 1024;
constexpr int kThreads = 256;
constexpr int kNW = kThreads / kWave;

struct LossWs {
    double stats[kMaxBlocks][2];
    double part[kMaxBlocks][6];
};

// STRIDE = 1: adv is a plain (M,) array; STRIDE = 4: adv points at field 1 of a packed (M,4) record
template <int STRIDE>
__global__ __launch_bounds__(kThreads) void k_adv_stats(const float* __restrict__ adv, int M,
                                                        double (*__restrict__ stats)[2]) {
    __shared__ double sc[2][kNW];
    double s = 0.0, q = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < M; i += gridDim.x * kThreads) {
        const double a = (double)adv[(size_t)i * STRIDE];
        s += a;
        q += a * a;
    }
    const double bs = block_sum<kNW>(s, sc[0]);
    const double bq = block_sum<kNW>(q, sc[1]);
    if (threadIdx.x == 0) {
        stats[blockIdx.x][0] = bs;
        stats[blockIdx.x][1] = bq;
    }
}

struct LossParams {
    PpoHyper h;
    int n_stat_blocks;
};

// PACKED: oldlogp points at a (M,4) record {old_logp, adv, ret, old_v}; adv/oldv/ret are ignored
template <bool PACKED>
__global__ __launch_bounds__(kThreads) void k_loss(const float* __restrict__ newlogp,
                                                   const float* __restrict__ oldlogp,
                                                   const float* __restrict__ adv, const float* __restrict__ newv,
                                                   const float* __restrict__ oldv, const float* __restrict__ ret,
                                                   const float* __restrict__ entropy, LossParams p,
                                                   float* __restrict__ g_newlogp, float* __restrict__ g_newv,
                                                   float* __restrict__ g_entropy, LossWs* __restrict__ ws) {
    __shared__ double sc[6][kNW];
    __shared__ float s_mean, s_std;
    // ---- minibatch statistics from the partials (same order in every workgroup)
    {
        double s = 0.0, q = 0.0;
        for (int b = threadIdx.x; b < p.n_stat_blocks; b += kThreads) {
            s += ws->stats[b][0];
            q += ws->stats[b][1];
        }
        const double ts = block_sum<kNW>(s, sc[0]);
        const double tq = block_sum<kNW>(q, sc[1]);
        if (threadIdx.x == 0) {
            const double m = ts / (double)p.h.M;
            double var = (tq - ts * m) / (double)(p.h.M - 1);  // M == 1 -> 0/0 = NaN, like torch.std
            if (var < 0.0) var = 0.0;
            s_mean = (float)m;
            s_std = (float)sqrt(var);
        }
        __syncthreads();
    }
    const float mean = s_mean;
    const float denom = s_std + 1e-8f;
    const float invM = 1.0f / (float)p.h.M;
    const float g_ent = -p.h.ent_coef * invM;
    double a_pg = 0.0, a_vl = 0.0, a_ent = 0.0, a_okl = 0.0, a_kl = 0.0, a_cf = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < p.h.M; i += gridDim.x * kThreads) {
        float ol_, a_raw, vo, R;
        if (PACKED) {
            const float4 r4 = reinterpret_cast<const float4*>(oldlogp)[i];
            ol_ = r4.x;
            a_raw = r4.y;
            R = r4.z;
            vo = r4.w;
        } else {
            ol_ = oldlogp[i];
            a_raw = adv[i];
            R = ret[i];
            vo = oldv[i];
        }
        const PpoSample t = ppo_sample(newlogp[i], ol_, a_raw, newv[i], vo, R, mean, denom, invM, p.h);
        a_okl += (double)t.okl;
        a_kl += (double)t.kl;
        a_cf += (double)t.cf;
        a_pg += (double)t.pg;
        a_vl += (double)t.vl;
        g_newlogp[i] = t.g_logp;
        g_newv[i] = t.g_v;
        a_ent += (double)entropy[i];
        g_entropy[i] = g_ent;
    }
    const double r0 = block_sum<kNW>(a_pg, sc[0]);
    const double r1 = block_sum<kNW>(a_vl, sc[1]);
    const double r2 = block_sum<kNW>(a_ent, sc[2]);
    const double r3 = block_sum<kNW>(a_okl, sc[3]);
    const double r4 = block_sum<kNW>(a_kl, sc[4]);
    const double r5 = block_sum<kNW>(a_cf, sc[5]);
    if (threadIdx.x == 0) {
        double* o = ws->part[blockIdx.x];
        o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3; o[4] = r4; o[5] = r5;
        if (blockIdx.x == 0) {
            // stash for k_loss_final (not aliased with any partial slot in use)
            ws->stats[kMaxBlocks - 1][0] = (double)mean;
            ws->stats[kMaxBlocks - 1][1] = (double)s_std;
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_loss_final(const LossWs* __restrict__ ws, int n_blocks, LossParams p,
                                                         float* __restrict__ out) {
    __shared__ double sc[6][kNW];
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < n_blocks; b += kThreads) {
#pragma unroll
        for (int k = 0; k < 6; ++k) a[k] += ws->part[b][k];
    }
    double r[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) r[k] = block_sum<kNW>(a[k], sc[k]);
    if (threadIdx.x == 0) {
        const double M = (double)p.h.M;
        const float pg = (float)(r[0] / M);
        const float vl = 0.5f * (float)(r[1] / M);
        const float ent = (float)(r[2] / M);
        out[AURPPO_S_PG] = pg;
        out[AURPPO_S_VL] = vl;
        out[AURPPO_S_ENT] = ent;
        out[AURPPO_S_OLD_KL] = (float)(r[3] / M);
        out[AURPPO_S_KL] = (float)(r[4] / M);
        out[AURPPO_S_CLIPFRAC] = (float)(r[5] / M);
        out[AURPPO_S_LOSS] = (pg - p.h.ent_coef * ent) + vl * p.h.vf_coef;
        out[AURPPO_S_ADV_MEAN] = (float)ws->stats[kMaxBlocks - 1][0];
        out[AURPPO_S_ADV_STD] = (float)ws->stats[kMaxBlocks - 1][1];
    }
}

}  // namespace

extern "C" size_t aurppo_loss_workspace_bytes(int M) {
    (void)M;
    return sizeof(LossWs);
}

static int loss_launch(bool packed, const float* newlogp, const float* oldlogp, const float* adv, const float* newv,
                       const float* oldv, const float* ret, const float* entropy, int M, double clip, double ent_coef,
                       double vf_coef, int norm_adv, int vloss_mode, float* out_scalars, float* g_newlogp,
                       float* g_newv, float* g_entropy, void* workspace, void* stream) {
    AURPPO_REQUIRE(newlogp && oldlogp && adv && newv && oldv && ret && entropy && out_scalars && g_newlogp && g_newv &&
                       g_entropy && workspace,
                   AURPPO_EINVAL, "aurppo_loss_fwd_bwd_f32: null pointer");
    AURPPO_REQUIRE(!packed || aligned_to(oldlogp, 16), AURPPO_EINVAL,
                   "aurppo_loss_fwd_bwd_packed_f32: rec not 16-byte aligned");
    AURPPO_REQUIRE(aligned_to(workspace, 16), AURPPO_EINVAL, "aurppo_loss_fwd_bwd_f32: workspace not 16-byte aligned");
    AURPPO_REQUIRE(vloss_mode >= 0 && vloss_mode <= 2, AURPPO_EINVAL, "aurppo_loss_fwd_bwd_f32: bad vloss_mode %d",
                   vloss_mode);
    AURPPO_REQUIRE(M > 0, AURPPO_ESHAPE, "aurppo_loss_fwd_bwd_f32: M=%d must be positive", M);
    LossWs* ws = reinterpret_cast<LossWs*>(workspace);
    hipStream_t s = (hipStream_t)stream;
    LossParams p;
    p.h = make_hyper(M, clip, ent_coef, vf_coef, norm_adv, vloss_mode);
    // 4 samples per lane per pass; the last stats slot is reserved for the mean/std stash
    int blocks = (M + kThreads * 4 - 1) / (kThreads * 4);
    if (blocks > kMaxBlocks - 1) blocks = kMaxBlocks - 1;
    p.n_stat_blocks = blocks;
    if (packed) {
        hipLaunchKernelGGL(k_adv_stats<4>, dim3(blocks), dim3(kThreads), 0, s, adv, M, ws->stats);
        AURPPO_LAUNCH_CHECK("k_adv_stats");
        hipLaunchKernelGGL(k_loss<true>, dim3(blocks), dim3(kThreads), 0, s, newlogp, oldlogp, adv, newv, oldv, ret,
                           entropy, p, g_newlogp, g_newv, g_entropy, ws);
    } else {
        hipLaunchKernelGGL(k_adv_stats<1>, dim3(blocks), dim3(kThreads), 0, s, adv, M, ws->stats);
        AURPPO_LAUNCH_CHECK("k_adv_stats");
        hipLaunchKernelGGL(k_loss<false>, dim3(blocks), dim3(kThreads), 0, s, newlogp, oldlogp, adv, newv, oldv, ret,
                           entropy, p, g_newlogp, g_newv, g_entropy, ws);
    }
    AURPPO_LAUNCH_CHECK("k_loss");
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(kThreads), 0, s, ws, blocks, p, out_scalars);
    AURPPO_LAUNCH_CHECK("k_loss_final");
    return AURPPO_OK;
}

extern "C" int aurppo_loss_fwd_bwd_f32(const float* newlogp, const float* oldlogp, const float* adv,
                                       const float* newv, const float* oldv, const float* ret,
                                       const float* entropy, int M, double clip, double ent_coef, double vf_coef,
                                       int norm_adv, int vloss_mode, float* out_scalars, float* g_newlogp,
                                       float* g_newv, float* g_entropy, void* workspace, void* stream) {
    return loss_launch(false, newlogp, oldlogp, adv, newv, oldv, ret, entropy, M, clip, ent_coef, vf_coef, norm_adv,
                       vloss_mode, out_scalars, g_newlogp, g_newv, g_entropy, workspace, stream);
}

extern "C" int aurppo_loss_fwd_bwd_packed_f32(const float* newlogp, const float* newv, const float* entropy,
                                              const float* rec, int M, double clip, double ent_coef, double vf_coef,
                                              int norm_adv, int vloss_mode, float* out_scalars, float* g_newlogp,
                                              float* g_newv, float* g_entropy, void* workspace, void* stream) {
    AURPPO_REQUIRE(rec, AURPPO_EINVAL, "aurppo_loss_fwd_bwd_packed_f32: null rec");
    return loss_launch(true, newlogp, rec, rec + 1, newv, rec + 3, rec + 2, entropy, M, clip, ent_coef, vf_coef,
                       norm_adv, vloss_mode, out_scalars, g_newlogp, g_newv, g_entropy, workspace, stream);
}
