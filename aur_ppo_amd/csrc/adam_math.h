// Shared pieces of the fused clip + Adam step (K6b): the clip coefficient from the partial sums of squares and
// torch's single-tensor Adam update for one element (src/ppo.py:80,268-269).  Used by clip.hip (flat bucket) and
// mlp.hip (the chained minibatch step).
#pragma once
#include "common.h"

struct AdamScalars {
    float coef;        // clip_grad_norm_ scale (<= 1, or NaN when the norm is NaN)
    float gscale;      // applied to the stored gradient first (1/world after a SUM all-reduce; 1 otherwise)
    float step_size;   // lr / (1 - beta1^t)
    float bc2_sqrt;    // sqrt(1 - beta2^t)
    float w1, b2, w2, eps;
};

// every thread of the block calls this; `sc` is shared scratch of blockDim.x / 64 doubles
template <int NW>
__device__ __forceinline__ AdamScalars adam_scalars(const double* __restrict__ part, int n_part, float max_norm,
                                                    const float* __restrict__ lr_dev, const float* __restrict__ step,
                                                    double beta1, double beta2, double eps, float* __restrict__ out_norm,
                                                    bool write_norm, double* sc, float* s_coef,
                                                    const double* __restrict__ bc = nullptr) {
    // the step count and the learning rate are fetched, and the bias corrections formed, while the partial sums are
    // still on their way: one memory round trip for the whole preamble instead of two
    const double tt = (double)*step;
    const double lr = (double)*lr_dev;
    double q = 0.0;
    for (int b = threadIdx.x; b < n_part; b += blockDim.x) q += part[b];
    // bc != nullptr: {1 - beta1^t, 1 - beta2^t} were formed by the launch that advanced t (k_mlp_reduce, one thread, beside its
    // slab loads): two double-precision pow() calls are the longest thing on this kernel's path otherwise
    const double bc1 = bc ? bc[0] : 1.0 - pow(beta1, tt);
    const double bc2 = bc ? bc[1] : 1.0 - pow(beta2, tt);
    const double t = block_sum<NW>(q, sc);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(t);
        float coef = max_norm / (norm + 1e-6f);
        *s_coef = coef < 1.0f ? coef : 1.0f;  // NaN norm -> NaN coef -> comparison false -> 1 (torch: clamp keeps NaN)
        if (coef != coef) *s_coef = coef;
        if (write_norm) *out_norm = norm;
    }
    __syncthreads();
    AdamScalars a;
    a.coef = *s_coef;
    a.gscale = 1.0f;
    a.step_size = (float)(lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.w1 = (float)(1.0 - beta1);
    a.b2 = (float)beta2;
    a.w2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    return a;
}

// adam_update (below) on operands the caller has already fetched (so that the loads overlap the clip preamble)
__device__ __forceinline__ float adam_update_pre(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, int64_t i, float p0, float g0, float m0, float v0,
                                                 const AdamScalars& a, bool store_g) {
    const float gi = (g0 * a.gscale) * a.coef;
    if (store_g) g[i] = gi;
    const float mi = m0 + a.w1 * (gi - m0);
    const float vi = v0 * a.b2 + (a.w2 * gi) * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / a.bc2_sqrt + a.eps;
    const float pn = p0 - a.step_size * (mi / denom);
    p[i] = pn;
    return pn;
}

// m.lerp_(g, 1-b1); v = v*b2 + (1-b2)*g*g; p -= step_size * m / (sqrt(v)/sqrt(1-b2^t) + eps); returns the new p
// store_g: leave the clipped gradient in g, as clip_grad_norm_ does (not when other workgroups of the same launch
// are still reading g to form the norm)
__device__ __forceinline__ float adam_update(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                             float* __restrict__ v, int64_t i, bool clip, const AdamScalars& a,
                                             bool store_g = true) {
    float gi = g[i] * a.gscale;
    if (clip) {
        gi = gi * a.coef;
        if (store_g) g[i] = gi;
    }
    const float mi = m[i] + a.w1 * (gi - m[i]);
    const float vi = v[i] * a.b2 + (a.w2 * gi) * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / a.bc2_sqrt + a.eps;
    const float pn = p[i] - a.step_size * (mi / denom);
    p[i] = pn;
    return pn;
}
