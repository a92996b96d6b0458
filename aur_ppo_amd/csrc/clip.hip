// K6: global-norm gradient clip over one flat fp32 bucket -- nn.utils.clip_grad_norm_
// (src/ppo.py:268; src/robot_ppo.py:401 clips the actor's bucket only).
//
// The trainer keeps every parameter gradient as a view into ONE flat buffer (the same buffer the
// RCCL all-reduce uses), so the clip is two small launches instead of torch's per-tensor norm /
// stack / norm / per-tensor scale chain: fp64 partial sums of squares, then every workgroup
// re-derives the norm in fixed order and scales its slice.
#include "common.h"

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
