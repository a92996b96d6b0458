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
#include <type_traits>

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

// ---------------------------------------------------------------------------------------------------------------------
// K10: the FIRST block of the encoder in one kernel -- 3x3 convolution (padding 1) of the image channels and of the tiled
// gripper-state plane, bias, ReLU, 2x2 max-pool -- forward, and its backward straight to the weight / bias gradients.
// (src/nets/base_cnns.py:28-31 with the input of src/models/robot_actor_critic.py:58-59,106-107.)
//
// Why this block: its output before the pool is the largest tensor of the network (8192 x 16 x 128 x 128 floats = 8.6 GB
// per minibatch and net) and its convolution has 1-3 input channels -- MIOpen writes those 8.6 GB, K9 reads them back, and
// in backward K9 writes a gradient of that size which the weight-gradient kernel and the plane reduction read again.
// Here the forward reads the observation (64 KB per sample) and writes the pooled block (262 KB + 64 KB of masks); the
// backward reads the pooled gradient, the masks and the observation and leaves per-workgroup partial sums of dW / db --
// the full-resolution tensors never exist.  Arithmetic: 36 FMAs per pooled element and channel, on the VALU (a 1-3 channel
// convolution is no matrix product worth an MFMA tile); weights come through scalar loads (uniform per workgroup).
//
// Semantics are those of conv2d(cat[obs, state plane]) + bias -> ReLU -> MaxPool2d(2): first maximum in scan order, nothing
// through x <= 0.  The input needs no gradient (it is data), so the backward is complete with dW, db.
namespace {

constexpr int kFbThreads = 256;
constexpr int kFbCo = 16;      // output channels per workgroup (the plain encoder's first block has 16; wider ones loop groups)

// wave-wide sum with DPP moves (no LDS traffic): quads, row halves, rows, then the four row sums through readlane
__device__ __forceinline__ float wave_sum_dpp(float v) {
    auto mv = [](float x, auto ctrl) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true)); };
    v += mv(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    v += mv(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    v += mv(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    v += mv(v, std::integral_constant<int, 0x140>{});   // row_mirror: every lane now holds its row's sum
    const int iv = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48)));
}

// grid: (ceil(Ho*Wo / 256), Co / 16, B).  A thread owns one pooled element for 16 output channels.
template <int CI>
__global__ __launch_bounds__(kFbThreads) void k_first_block_fwd(const float* __restrict__ obs, const float* __restrict__ w,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ state, float* __restrict__ y,
                                                                uint8_t* __restrict__ mask, int Co, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int q = blockIdx.x * kFbThreads + threadIdx.x;
    const int cg = blockIdx.y, b = blockIdx.z;
    const bool live = q < Ho * Wo;
    const int ho = live ? q / Wo : 0, wo = live ? q - ho * Wo : 0;
    const float sv = state[b];
    // the 4x4 input patch of this pooled cell, per image channel (rows 2ho-1 .. 2ho+2, cols 2wo-1 .. 2wo+2), zero outside
    float p[CI][4][4];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
        const float* op = obs + ((size_t)b * CI + ci) * H * W;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = 2 * ho - 1 + r;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ww = 2 * wo - 1 + c;
                const bool in = live && h >= 0 && h < H && ww >= 0 && ww < W;
                p[ci][r][c] = in ? op[(size_t)(in ? h : 0) * W + (in ? ww : 0)] : 0.0f;
            }
        }
    }
    // border classes of the four conv positions (rows 2ho, 2ho+1; cols 2wo, 2wo+1) as 0/1 factors: the state plane's
    // response is the sum of the channel's state taps that fall inside the image = all - border row - border column + corner
    const float ft = ho == 0 ? 1.0f : 0.0f, fb = 2 * ho + 1 == H - 1 ? 1.0f : 0.0f;
    const float fl = wo == 0 ? 1.0f : 0.0f, fr = 2 * wo + 1 == W - 1 ? 1.0f : 0.0f;
    for (int cc = 0; cc < kFbCo; ++cc) {
        const int c = cg * kFbCo + cc;
        const float* wc = w + (size_t)c * (CI + 1) * 9;       // uniform address: scalar loads
        float acc[4];
        const float bv = bias ? bias[c] : 0.0f;
        const float* ts = wc + CI * 9;
        // state plane first, then the bias (the association of base_encoder.forward_split: (conv + state*plane) + bias)
        const float all = ((ts[0] + ts[1]) + (ts[2] + ts[3])) + ((ts[4] + ts[5]) + (ts[6] + ts[7])) + ts[8];
        const float r0 = ts[0] + ts[1] + ts[2], r2 = ts[6] + ts[7] + ts[8];
        const float c0 = ts[0] + ts[3] + ts[6], c2 = ts[2] + ts[5] + ts[8];
        const float pl[4] = {all - ft * r0 - fl * c0 + ft * fl * ts[0], all - ft * r0 - fr * c2 + ft * fr * ts[2],
                             all - fb * r2 - fl * c0 + fb * fl * ts[6], all - fb * r2 - fr * c2 + fb * fr * ts[8]};
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = 0.0f;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float wv = wc[ci * 9 + kh * 3 + kw];
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[u] = fmaf(p[ci][(u >> 1) + kh][(u & 1) + kw], wv, acc[u]);
                }
        float m = 0.0f;
        int k = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float v = (acc[u] + sv * pl[u]) + bv;
            if (u == 0 || v > m) { m = v; k = u; }
        }
        const bool alive = m > 0.0f;
        if (live) {
            const size_t o = (((size_t)b * Co + c) * Ho + ho) * Wo + wo;
            y[o] = alive ? m : 0.0f;
            mask[o] = alive ? (uint8_t)k : (uint8_t)4;
        }
    }
}

// grid: (1, Co / 16, B): one workgroup of 16 waves per sample and channel group, ONE WAVE PER OUTPUT CHANNEL; a wave walks all
// pooled elements of its channel and leaves dw_part[(b * Co/16 + cg)][channel][(CI+1)*9] and db_part[...][channel].
// (First version: 256 threads walking the 16 channels one after the other -- PMC showed 8.1 GB fetched against 3.2 GB
// algorithmic: the observation was re-read from beyond L2 once per channel.  Sixteen waves sweeping the same image together share
// it through the CU's L1, and a channel's sums are a wave reduction -- DPP, no LDS, no barrier.)
constexpr int kFbBwdThreads = kFbCo * kWave;

template <int CI>
__global__ __launch_bounds__(kFbBwdThreads) void k_first_block_bwd(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                                   const float* __restrict__ obs,
                                                                   const float* __restrict__ state,
                                                                   float* __restrict__ dw_part, float* __restrict__ db_part,
                                                                   int Co, int H, int W) {
    constexpr int NT = (CI + 1) * 9;
    const int Ho = H >> 1, Wo = W >> 1;
    const int cg = blockIdx.y, b = blockIdx.z;
    const float sv = state[b];
    const int lane = threadIdx.x & (kWave - 1);
    const int cc = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);      // this wave's channel within the group
    const int c = cg * kFbCo + cc;
    const float* gp = dy + ((size_t)b * Co + c) * Ho * Wo;
    const uint8_t* mp = mask + ((size_t)b * Co + c) * Ho * Wo;
    const float* ob = obs + (size_t)b * CI * H * W;
    // a[0 .. CI*9): image taps; then nine running sums of the output gradient for the state plane's taps (all / first row /
    // last row / first column / last column / four corners: a tap's gradient is all - excluded row - excluded column + corner)
    constexpr int NA = CI * 9 + 9;
    float a[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t) a[t] = 0.0f;
    const int dqh = kWave / Wo, dqw = kWave - dqh * Wo;             // a lane's position advances by 64 pooled elements
    int ho = lane / Wo, wo = lane - ho * Wo;
    // What paces this kernel is the vector-memory front end, not HBM (PMC: 3.1 GB fetched = algorithmic) and not arithmetic:
    // nine scattered dword gathers per position and channel.  A window row is three consecutive floats, so it is ONE 12-byte
    // load from a column base clamped into the image; the border cases pick their taps out of it with selects.
    typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
    for (int q = lane; q < Ho * Wo; q += kWave) {
        const int mk = mp[q];
        const float g = mk < 4 ? gp[q] : 0.0f;
        const int h = 2 * ho + ((mk & 3) >> 1), ww = 2 * wo + (mk & 1);      // where the maximum sat
        const bool t0 = h == 0, b0 = h == H - 1, l0 = ww == 0, r0 = ww == W - 1;
        const int rr[3] = {(t0 ? 0 : h - 1) * W, h * W, (b0 ? h : h + 1) * W};      // clamped rows (a clamped row's weight is 0)
        const int cb = l0 ? 0 : (r0 ? W - 3 : ww - 1);                              // first column of the 12-byte load (W >= 3)
        const float gr[3] = {t0 ? 0.0f : g, g, b0 ? 0.0f : g};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const float gk[3] = {l0 ? 0.0f : gr[kh], gr[kh], r0 ? 0.0f : gr[kh]};
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
                const f32x3u v = *reinterpret_cast<const f32x3u*>(ob + (size_t)ci * H * W + rr[kh] + cb);
                // columns ww-1, ww, ww+1: at the left border the load starts at column 0 (= ww), at the right one at W-3
                const float x0 = r0 ? v.y : v.x;                    // (left border: weight 0, any value)
                const float x1 = l0 ? v.x : (r0 ? v.z : v.y);
                const float x2 = l0 ? v.y : v.z;                    // (right border: weight 0)
                a[ci * 9 + kh * 3 + 0] = fmaf(gk[0], x0, a[ci * 9 + kh * 3 + 0]);
                a[ci * 9 + kh * 3 + 1] = fmaf(gk[1], x1, a[ci * 9 + kh * 3 + 1]);
                a[ci * 9 + kh * 3 + 2] = fmaf(gk[2], x2, a[ci * 9 + kh * 3 + 2]);
            }
        }
        float* sg = a + CI * 9;
        const float gt = t0 ? g : 0.0f, gb = b0 ? g : 0.0f;
        sg[0] += g; sg[1] += gt; sg[2] += gb; sg[3] += l0 ? g : 0.0f; sg[4] += r0 ? g : 0.0f;
        sg[5] += l0 ? gt : 0.0f; sg[6] += r0 ? gt : 0.0f; sg[7] += l0 ? gb : 0.0f; sg[8] += r0 ? gb : 0.0f;
        wo += dqw; ho += dqh;
        if (wo >= Wo) { wo -= Wo; ++ho; }
    }
    const size_t g_ = (size_t)b * gridDim.y + cg;
    float red[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t) red[t] = wave_sum_dpp(a[t]);          // fixed order: deterministic; every lane holds the sums
    if (lane == 0) {
        float* dst = dw_part + (g_ * kFbCo + cc) * NT;
#pragma unroll
        for (int t = 0; t < CI * 9; ++t) dst[t] = red[t];
        const float* G = red + CI * 9;         // all, top, bottom, left, right, tl, tr, bl, br
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                float v = G[0];
                if (kh == 0) v -= G[1];
                if (kh == 2) v -= G[2];
                if (kw == 0) v -= G[3];
                if (kw == 2) v -= G[4];
                if (kh == 0 && kw == 0) v += G[5];
                if (kh == 0 && kw == 2) v += G[6];
                if (kh == 2 && kw == 0) v += G[7];
                if (kh == 2 && kw == 2) v += G[8];
                dst[CI * 9 + kh * 3 + kw] = sv * v;
            }
        db_part[g_ * kFbCo + cc] = G[0];
    }
}

}  // namespace

extern "C" int aurppo_first_block_fwd_f32(const float* obs, const float* w, const float* bias, const float* state, float* y,
                                          uint8_t* mask, int B, int Ci, int Co, int H, int W, void* stream) {
    AURPPO_REQUIRE(obs && w && state && y && mask, AURPPO_EINVAL, "aurppo_first_block_fwd_f32: null pointer");
    AURPPO_REQUIRE(B > 0 && Ci >= 1 && Ci <= 3 && Co > 0 && Co % kFbCo == 0 && Co / kFbCo <= 65535 && H >= 2 && W >= 3,
                   AURPPO_ESHAPE, "aurppo_first_block_fwd_f32: B=%d Ci=%d Co=%d H=%d W=%d (Ci in 1..3, Co a multiple of 16, W >= 3)", B, Ci, Co, H, W);
    hipStream_t s = (hipStream_t)stream;
    const size_t in_s = (size_t)Ci * H * W, out_s = (size_t)Co * (H / 2) * (W / 2);
    for (int b0 = 0; b0 < B; b0 += 65535) {          // the sample index rides in gridDim.z (<= 65535): larger batches in slices
        const int nb = B - b0 < 65535 ? B - b0 : 65535;
        const dim3 grid(((H / 2) * (W / 2) + kFbThreads - 1) / kFbThreads, Co / kFbCo, nb);
        const float* o = obs + b0 * in_s;
        const float* st = state + b0;
        float* yy = y + b0 * out_s;
        uint8_t* mm = mask + b0 * out_s;
        if (Ci == 1) hipLaunchKernelGGL(k_first_block_fwd<1>, grid, dim3(kFbThreads), 0, s, o, w, bias, st, yy, mm, Co, H, W);
        else if (Ci == 2) hipLaunchKernelGGL(k_first_block_fwd<2>, grid, dim3(kFbThreads), 0, s, o, w, bias, st, yy, mm, Co, H, W);
        else hipLaunchKernelGGL(k_first_block_fwd<3>, grid, dim3(kFbThreads), 0, s, o, w, bias, st, yy, mm, Co, H, W);
        AURPPO_LAUNCH_CHECK("k_first_block_fwd");
    }
    return AURPPO_OK;
}

extern "C" int aurppo_first_block_bwd_f32(const float* dy, const uint8_t* mask, const float* obs, const float* state,
                                          float* dw_part, float* db_part, int B, int Ci, int Co, int H, int W, void* stream) {
    AURPPO_REQUIRE(dy && mask && obs && state && dw_part && db_part, AURPPO_EINVAL, "aurppo_first_block_bwd_f32: null pointer");
    AURPPO_REQUIRE(B > 0 && Ci >= 1 && Ci <= 3 && Co > 0 && Co % kFbCo == 0 && Co / kFbCo <= 65535 && H >= 2 && W >= 3,
                   AURPPO_ESHAPE, "aurppo_first_block_bwd_f32: B=%d Ci=%d Co=%d H=%d W=%d (W >= 3)", B, Ci, Co, H, W);
    hipStream_t s = (hipStream_t)stream;
    const size_t in_s = (size_t)Ci * H * W, out_s = (size_t)Co * (H / 2) * (W / 2);
    const size_t g_s = (size_t)(Co / kFbCo) * kFbCo;             // partial-sum rows per sample
    for (int b0 = 0; b0 < B; b0 += 65535) {
        const int nb = B - b0 < 65535 ? B - b0 : 65535;
        const dim3 grid(1, Co / kFbCo, nb);
        const float* g = dy + b0 * out_s;
        const uint8_t* mm = mask + b0 * out_s;
        const float* o = obs + b0 * in_s;
        const float* st = state + b0;
        float* dwp = dw_part + b0 * g_s * (size_t)((Ci + 1) * 9);
        float* dbp = db_part + b0 * g_s;
        if (Ci == 1) hipLaunchKernelGGL(k_first_block_bwd<1>, grid, dim3(kFbBwdThreads), 0, s, g, mm, o, st, dwp, dbp, Co, H, W);
        else if (Ci == 2) hipLaunchKernelGGL(k_first_block_bwd<2>, grid, dim3(kFbBwdThreads), 0, s, g, mm, o, st, dwp, dbp, Co, H, W);
        else hipLaunchKernelGGL(k_first_block_bwd<3>, grid, dim3(kFbBwdThreads), 0, s, g, mm, o, st, dwp, dbp, Co, H, W);
        AURPPO_LAUNCH_CHECK("k_first_block_bwd");
    }
    return AURPPO_OK;
}
