// K6: global-norm gradient clip over one flat fp32 bucket -- nn.utils.clip_grad_norm_
// (src/ppo.py:268; src/robot_ppo.py:401 clips the actor's bucket only).
//
// The trainer keeps every parameter gradient as a view into ONE flat buffer (the same buffer the
// RCCL all-reduce uses), so the clip is two small launches instead of torch's per-tensor norm /
// stack / norm / per-tensor scale chain: fp64 partial sums of squares, then every workgroup
// re-derives the norm in fixed order and scales its slice.
#include "adam_math.h"

namespace {

constexpr int kMaxBlocks = 512;
constexpr int kThreads = 256;
constexpr int kNW = kThreads / kWave;

__global__ __launch_bounds__(kThreads) void k_sqnorm(const float* __restrict__ g, int64_t n,
                                                     double* __restrict__ part) {
    __shared__ double sc[kNW];
    double q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const double x = (double)g[i];
        q += x * x;
    }
    const double b = block_sum<kNW>(q, sc);
    if (threadIdx.x == 0) part[blockIdx.x] = b;
}

__global__ __launch_bounds__(kThreads) void k_clip_scale(float* __restrict__ g, int64_t n,
                                                         const double* __restrict__ part, int n_part, float max_norm,
                                                         float* __restrict__ out_norm) {
    __shared__ double sc[kNW];
    __shared__ float s_coef;
    double q = 0.0;
    for (int b = threadIdx.x; b < n_part; b += kThreads) q += part[b];
    const double t = block_sum<kNW>(q, sc);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(t);
        float coef = max_norm / (norm + 1e-6f);
        s_coef = coef < 1.0f ? coef : 1.0f;  // NaN norm -> NaN coef -> comparison false -> 1 (torch: clamp keeps NaN)
        if (coef != coef) s_coef = coef;
        if (blockIdx.x == 0) *out_norm = norm;
    }
    __syncthreads();
    const float coef = s_coef;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
        g[i] = g[i] * coef;
}

}  // namespace

extern "C" size_t aurppo_clip_workspace_bytes(int64_t n) {
    (void)n;
    return sizeof(double) * kMaxBlocks;
}

extern "C" int aurppo_grad_norm_clip_f32(float* flat_grads, int64_t n, double max_norm, float* out_norm,
                                         void* workspace, void* stream) {
    AURPPO_REQUIRE(flat_grads && out_norm && workspace, AURPPO_EINVAL, "aurppo_grad_norm_clip_f32: null pointer");
    AURPPO_REQUIRE(aligned_to(workspace, 8), AURPPO_EINVAL, "aurppo_grad_norm_clip_f32: workspace not 8-byte aligned");
    AURPPO_REQUIRE(n > 0, AURPPO_ESHAPE, "aurppo_grad_norm_clip_f32: n=%lld must be positive", (long long)n);
    int64_t want = (n + kThreads * 4 - 1) / (kThreads * 4);
    const int blocks = (int)(want > kMaxBlocks ? kMaxBlocks : want);
    double* part = reinterpret_cast<double*>(workspace);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sqnorm, dim3(blocks), dim3(kThreads), 0, s, flat_grads, n, part);
    AURPPO_LAUNCH_CHECK("k_sqnorm");
    hipLaunchKernelGGL(k_clip_scale, dim3(blocks), dim3(kThreads), 0, s, flat_grads, n, part, blocks,
                       (float)max_norm, out_norm);
    AURPPO_LAUNCH_CHECK("k_clip_scale");
    return AURPPO_OK;
}

// ---- K6b: global-norm clip + Adam in two launches over the flat bucket -------------------------------
// torch.optim.Adam(eps=1e-5) after nn.utils.clip_grad_norm_ (src/ppo.py:80,268-269).  torch's
// capturable foreach Adam costs ~35 tiny kernels per step on 13 small tensors (rocprof: ~160 us per
// minibatch); with parameters, gradients and both moment buffers flat it is one pass over 17 k floats.
// Formulas are torch's single-tensor Adam: m.lerp_(g, 1-b1); v = v*b2 + (1-b2)*g*g;
// p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps), scalars formed in fp64 then rounded.
namespace {

__global__ __launch_bounds__(kThreads) void k_sqnorm_step(const float* __restrict__ g, int64_t n,
                                                          double* __restrict__ part, float* __restrict__ step) {
    __shared__ double sc[kNW];
    double q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const double x = (double)g[i];
        q += x * x;
    }
    const double b = block_sum<kNW>(q, sc);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = b;
        if (blockIdx.x == 0) *step += 1.0f;   // the update kernel (next launch) reads the new step count
    }
}

__global__ __launch_bounds__(kThreads) void k_clip_adam(float* __restrict__ p, float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        int64_t clip_n, const double* __restrict__ part, int n_part,
                                                        float max_norm, const float* __restrict__ lr_dev,
                                                        const float* __restrict__ step, double beta1, double beta2,
                                                        double eps, float* __restrict__ out_norm) {
    __shared__ double sc[kNW];
    __shared__ float s_coef;
    const AdamScalars a = adam_scalars<kNW>(part, n_part, max_norm, lr_dev, step, beta1, beta2, eps, out_norm,
                                            blockIdx.x == 0, sc, &s_coef);
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
        (void)adam_update(p, g, m, v, i, i < clip_n, a);
}

}  // namespace

extern "C" int aurppo_clip_adam_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                    int64_t clip_n, double max_norm, const float* lr_dev, float* step_dev, double beta1,
                                    double beta2, double eps, float* out_norm, void* workspace, void* stream) {
    AURPPO_REQUIRE(params && grads && exp_avg && exp_avg_sq && lr_dev && step_dev && out_norm && workspace, AURPPO_EINVAL,
                   "aurppo_clip_adam_f32: null pointer");
    AURPPO_REQUIRE(aligned_to(workspace, 8), AURPPO_EINVAL, "aurppo_clip_adam_f32: workspace not 8-byte aligned");
    AURPPO_REQUIRE(n > 0 && clip_n >= 0 && clip_n <= n, AURPPO_ESHAPE, "aurppo_clip_adam_f32: n=%lld clip_n=%lld",
                   (long long)n, (long long)clip_n);
    int64_t want = ((clip_n ? clip_n : 1) + kThreads * 4 - 1) / (kThreads * 4);
    const int nb_norm = (int)(want > kMaxBlocks ? kMaxBlocks : want);
    want = (n + kThreads * 4 - 1) / (kThreads * 4);
    const int nb_upd = (int)(want > kMaxBlocks ? kMaxBlocks : want);
    double* part = reinterpret_cast<double*>(workspace);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sqnorm_step, dim3(nb_norm), dim3(kThreads), 0, s, grads, clip_n, part, step_dev);
    AURPPO_LAUNCH_CHECK("k_sqnorm_step");
    hipLaunchKernelGGL(k_clip_adam, dim3(nb_upd), dim3(kThreads), 0, s, params, grads, exp_avg, exp_avg_sq, n, clip_n,
                       part, nb_norm, (float)max_norm, lr_dev, step_dev, beta1, beta2, eps, out_norm);
    AURPPO_LAUNCH_CHECK("k_clip_adam");
    return AURPPO_OK;
}
