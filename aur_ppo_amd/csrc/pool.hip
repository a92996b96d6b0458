// K9: the element-wise tail of a convolution block of the robot policy's encoder, fused:
//     y = maxpool2x2( relu( x + bias[c] + scale[b] * plane[c,h,w] ) )        forward
//     dx, per-(b,c) bias-gradient sums                                         backward
// It stands in for `nn.ReLU` + `nn.MaxPool2d(2)` behind each `nn.Conv2d` of src/nets/base_cnns.py:28-45 (and the bias
// add of that convolution), and for the tile-the-gripper-state-into-a-plane + concat of
// src/models/robot_actor_critic.py:58-59,106-107 (the plane's convolution response enters as scale * plane).
//
// Why: rocprof of robot_ppo.update at BASELINE config 3's shape put 53 % of the GPU time outside the convolutions, in
// memory-bound passes over activations the size of the first block's output (8192 x 16 x 128 x 128 floats = 8.6 GB per
// minibatch and net): bias add, ReLU, max-pool forward with int64 indices, max-pool backward, ReLU backward, the bias
// gradient's reduction -- 11.5 passes over X per block.  Fused it is 1.25 X forward (read X, write X/4 and a byte mask
// per pooled element) and 1.3 X backward (read dY and the mask, write dX; the bias sums fall out of the same loads).
//
// torch semantics kept: max-pool takes the FIRST maximum of a window in row-major scan order (ties); ReLU passes no
// gradient at x <= 0; an odd trailing row / column is dropped by the pool (floor) and gets a zero gradient.
#include "common.h"

namespace {

constexpr int kPoolThreads = 256;

// TPP threads per (b, c) plane: a whole workgroup for the large planes of the first blocks, one wave per plane (four planes
// per workgroup) once a plane is a few hundred elements -- a 16x16 plane gave a 256-thread workgroup a quarter of a
// float4 per thread (2.2 TB/s against 5.5 on the large planes).  A thread walks pooled elements.
template <int TPP>
__global__ __launch_bounds__(kPoolThreads) void k_brp_fwd(const float* __restrict__ x, const float* __restrict__ bias,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ plane, float* __restrict__ y,
                                                          uint8_t* __restrict__ mask, int C, int H, int W, int n_planes) {
    const int bc = blockIdx.x * (kPoolThreads / TPP) + (int)threadIdx.x / TPP;
    if (bc >= n_planes) return;                      // wave-uniform for TPP = 64
    const int tpl = (int)threadIdx.x % TPP;
    const int c = bc % C, b = bc / C;
    const int Ho = H >> 1, Wo = W >> 1;
    const float bv = bias ? bias[c] : 0.0f;
    const float sv = scale ? scale[b] : 0.0f;
    const float* xp = x + (size_t)bc * H * W;
    const float* pp = plane ? plane + (size_t)c * H * W : nullptr;
    float* yp = y + (size_t)bc * Ho * Wo;
    uint8_t* mp = mask + (size_t)bc * Ho * Wo;
    const bool vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(xp) & 15) == 0) &&
                     (!pp || (reinterpret_cast<uintptr_t>(pp) & 15) == 0);
    if (vec) {
        // a thread takes two pooled elements: one float4 from each of the two input rows
        const int Wq = Wo >> 1;                      // pairs of pooled elements per pooled row
        for (int q = tpl; q < Ho * Wq; q += TPP) {
            const int ho = q / Wq, wq = q - ho * Wq;
            const size_t o0 = (size_t)(2 * ho) * W + 4 * wq;
            float4 r0 = *reinterpret_cast<const float4*>(xp + o0);
            float4 r1 = *reinterpret_cast<const float4*>(xp + o0 + W);
            if (pp) {
                const float4 p0 = *reinterpret_cast<const float4*>(pp + o0);
                const float4 p1 = *reinterpret_cast<const float4*>(pp + o0 + W);
                r0.x += sv * p0.x; r0.y += sv * p0.y; r0.z += sv * p0.z; r0.w += sv * p0.w;
                r1.x += sv * p1.x; r1.y += sv * p1.y; r1.z += sv * p1.z; r1.w += sv * p1.w;
            }
            const float v[2][4] = {{r0.x + bv, r0.y + bv, r1.x + bv, r1.y + bv}, {r0.z + bv, r0.w + bv, r1.z + bv, r1.w + bv}};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float m = v[e][0];
                int k = 0;
#pragma unroll
                for (int u = 1; u < 4; ++u)
                    if (v[e][u] > m) { m = v[e][u]; k = u; }      // strict: the first maximum wins, as torch's scan does
                const bool alive = m > 0.0f;
                yp[(size_t)ho * Wo + 2 * wq + e] = alive ? m : 0.0f;
                mp[(size_t)ho * Wo + 2 * wq + e] = alive ? (uint8_t)k : (uint8_t)4;
            }
        }
        return;
    }
    for (int q = tpl; q < Ho * Wo; q += TPP) {
        const int ho = q / Wo, wo = q - ho * Wo;
        float m = 0.0f;
        int k = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t o = (size_t)(2 * ho + (u >> 1)) * W + 2 * wo + (u & 1);
            const float v = xp[o] + (pp ? sv * pp[o] : 0.0f) + bv;
            if (u == 0 || v > m) { m = v; k = u; }
        }
        const bool alive = m > 0.0f;
        yp[q] = alive ? m : 0.0f;
        mp[q] = alive ? (uint8_t)k : (uint8_t)4;
    }
}

// TPP threads per (b, c) plane, as above; a thread walks INPUT elements (so every element of dx is written, the dropped
// odd row / column included) and the plane's threads leave its sum of live output gradients
template <int TPP>
__global__ __launch_bounds__(kPoolThreads) void k_brp_bwd(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                          float* __restrict__ dx, float* __restrict__ dbias_part,
                                                          int H, int W, int n_planes) {
    __shared__ double s_red[kPoolThreads / kWave];
    const int bc_raw = blockIdx.x * (kPoolThreads / TPP) + (int)threadIdx.x / TPP;
    const bool live = bc_raw < n_planes;             // wave-uniform for TPP = 64; a dead wave still joins block_sum's barrier
    const int bc = live ? bc_raw : 0;
    const int tpl = (int)threadIdx.x % TPP;
    const int Ho = H >> 1, Wo = W >> 1;
    const float* gp = dy + (size_t)bc * Ho * Wo;
    const uint8_t* mp = mask + (size_t)bc * Ho * Wo;
    float* dp = dx + (size_t)bc * H * W;
    const bool vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(dp) & 15) == 0);
    double part = 0.0;
    if (!live) {
    } else if (vec) {
        const int Wq = W >> 2;
        for (int q = tpl; q < H * Wq; q += TPP) {
            const int h = q / Wq, wq = q - h * Wq;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            const int ho = h >> 1;
            if (ho < Ho) {
                const int r = (h & 1) << 1;                           // window positions of this input row: r, r + 1
                const float g0 = gp[(size_t)ho * Wo + 2 * wq], g1 = gp[(size_t)ho * Wo + 2 * wq + 1];
                const int m0 = mp[(size_t)ho * Wo + 2 * wq], m1 = mp[(size_t)ho * Wo + 2 * wq + 1];
                o.x = m0 == r ? g0 : 0.0f;
                o.y = m0 == r + 1 ? g0 : 0.0f;
                o.z = m1 == r ? g1 : 0.0f;
                o.w = m1 == r + 1 ? g1 : 0.0f;
                if (!(h & 1)) part += (double)((m0 < 4 ? g0 : 0.0f) + (m1 < 4 ? g1 : 0.0f));   // each pooled element once
            }
            *reinterpret_cast<float4*>(dp + (size_t)h * W + 4 * wq) = o;
        }
    } else {
        for (int q = tpl; q < H * W; q += TPP) {
            const int h = q / W, w = q - h * W;
            const int ho = h >> 1, wo = w >> 1;
            float o = 0.0f;
            if (ho < Ho && wo < Wo) {
                const float g = gp[(size_t)ho * Wo + wo];
                const int m = mp[(size_t)ho * Wo + wo];
                o = m == (((h & 1) << 1) | (w & 1)) ? g : 0.0f;
                if (!(h & 1) && !(w & 1)) part += (double)(m < 4 ? g : 0.0f);
            }
            dp[q] = o;
        }
    }
    if (dbias_part) {
        if (TPP == kPoolThreads) {
            const double t = block_sum<kPoolThreads / kWave>(part, s_red);
            if (threadIdx.x == 0 && live) dbias_part[bc] = (float)t;
        } else {
            const double t = wave_sum(part);
            if (tpl == 0 && live) dbias_part[bc] = (float)t;
        }
    }
}

// out[k] = sum_b w[b] * x[b, k]   (the gradient of the state plane: robot_actor_critic.py:58-59's tiled input channel)
__global__ __launch_bounds__(256) void k_weighted_batch_sum(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ out, int B, long long K) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = 0;
    for (; b + 4 <= B; b += 4) {
        a0 += w[b] * x[(size_t)b * K + k];
        a1 += w[b + 1] * x[(size_t)(b + 1) * K + k];
        a2 += w[b + 2] * x[(size_t)(b + 2) * K + k];
        a3 += w[b + 3] * x[(size_t)(b + 3) * K + k];
    }
    for (; b < B; ++b) a0 += w[b] * x[(size_t)b * K + k];
    out[k] = (a0 + a1) + (a2 + a3);
}

}  // namespace

extern "C" int aurppo_bias_relu_pool2_fwd_f32(const float* x, const float* bias, const float* scale, const float* plane,
                                              float* y, uint8_t* mask, int B, int C, int H, int W, void* stream) {
    AURPPO_REQUIRE(x && y && mask, AURPPO_EINVAL, "aurppo_bias_relu_pool2_fwd_f32: null pointer");
    AURPPO_REQUIRE((scale == nullptr) == (plane == nullptr), AURPPO_EINVAL,
                   "aurppo_bias_relu_pool2_fwd_f32: scale and plane come together");
    AURPPO_REQUIRE(B > 0 && C > 0 && H >= 2 && W >= 2 && (long long)B * C < 2147483647LL, AURPPO_ESHAPE,
                   "aurppo_bias_relu_pool2_fwd_f32: B=%d C=%d H=%d W=%d", B, C, H, W);
    const int n_planes = B * C;
    if (H * W <= 1024)
        hipLaunchKernelGGL(k_brp_fwd<kWave>, dim3((n_planes + 3) / 4), dim3(kPoolThreads), 0, (hipStream_t)stream, x, bias, scale,
                           plane, y, mask, C, H, W, n_planes);
    else
        hipLaunchKernelGGL(k_brp_fwd<kPoolThreads>, dim3(n_planes), dim3(kPoolThreads), 0, (hipStream_t)stream, x, bias, scale,
                           plane, y, mask, C, H, W, n_planes);
    AURPPO_LAUNCH_CHECK("k_brp_fwd");
    return AURPPO_OK;
}

extern "C" int aurppo_bias_relu_pool2_bwd_f32(const float* dy, const uint8_t* mask, float* dx, float* dbias_part, int B,
                                              int C, int H, int W, void* stream) {
    AURPPO_REQUIRE(dy && mask && dx, AURPPO_EINVAL, "aurppo_bias_relu_pool2_bwd_f32: null pointer");
    AURPPO_REQUIRE(B > 0 && C > 0 && H >= 2 && W >= 2 && (long long)B * C < 2147483647LL, AURPPO_ESHAPE,
                   "aurppo_bias_relu_pool2_bwd_f32: B=%d C=%d H=%d W=%d", B, C, H, W);
    const int n_planes = B * C;
    if (H * W <= 1024)
        hipLaunchKernelGGL(k_brp_bwd<kWave>, dim3((n_planes + 3) / 4), dim3(kPoolThreads), 0, (hipStream_t)stream, dy, mask, dx,
                           dbias_part, H, W, n_planes);
    else
        hipLaunchKernelGGL(k_brp_bwd<kPoolThreads>, dim3(n_planes), dim3(kPoolThreads), 0, (hipStream_t)stream, dy, mask, dx,
                           dbias_part, H, W, n_planes);
    AURPPO_LAUNCH_CHECK("k_brp_bwd");
    return AURPPO_OK;
}

extern "C" int aurppo_weighted_batch_sum_f32(const float* x, const float* w, float* out, int B, int64_t K, void* stream) {
    AURPPO_REQUIRE(x && w && out, AURPPO_EINVAL, "aurppo_weighted_batch_sum_f32: null pointer");
    AURPPO_REQUIRE(B > 0 && K > 0, AURPPO_ESHAPE, "aurppo_weighted_batch_sum_f32: B=%d K=%lld", B, (long long)K);
    hipLaunchKernelGGL(k_weighted_batch_sum, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, out,
                       B, (long long)K);
    AURPPO_LAUNCH_CHECK("k_weighted_batch_sum");
    return AURPPO_OK;
}
