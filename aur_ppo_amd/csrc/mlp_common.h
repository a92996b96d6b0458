// Shared pieces of the fused MLP kernels (K7 k_mlp_step2 / k_mlp_step3, K8 k_mlp_act): tile constants, the
// argument block, the fp32 MFMA micro-kernels and the tanh used by every variant.
#pragma once
#include "ppo_math.h"

// Diagnostic build knob: s_setprio around the MFMA chunks of mma32 (0 = off).
#ifndef AURPPO_MMA_PRIO
#define AURPPO_MMA_PRIO 0
#endif

namespace aurppo_mlp {

constexpr int H = 64;        // hidden width
constexpr int R = 32;        // rows per tile
constexpr int LD = H + 1;    // LDS row stride of every 64-wide matrix (odd: conflict-free both ways)
constexpr int AP = 16;       // padded head width (action_dim <= 16)
constexpr int LDO = AP + 1;
constexpr int kThreads = 256;
constexpr int kMaxGrid = 256;
constexpr int kStatBlocks = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MlpLayout {  // float offsets into the flat parameter / gradient bucket
    int w1[2], b1[2], w2[2], b2[2], w3[2], b3[2];  // [0] actor, [1] critic
    int logstd;
    int n_params;
};

struct MlpArgs {
    const float* obs;      // (B, D) rollout observations (flattened buffer)
    const float* actions;  // (B, A)
    const float4* rec;     // (B, 4) {old_logp, adv, ret, old_v}; packed mode: (B, 16) with the action row in floats 4..15
    int rec_stride;        // float4s per record: 1, or 4 in packed mode (actions == nullptr)
    const int32_t* idx;    // (M,) minibatch permutation slice
    const float* params;   // flat bucket
    float* slabs;          // (grid, n_params) per-workgroup gradient slabs
    double* loss_part;     // (grid, 8)
    unsigned long long* stamps;  // diagnostic build: (grid, 16) cycle counters
    unsigned* tile_counter;  // next tile to hand out ([0]), zeroed by k_adv_stats_idx / k_mlp_reduce
    float* w1op;           // two-set kernel: W1 slices in MFMA B-operand order, [4 waves][32 k-steps][64 lanes]
    const void* wop3;      // k_mlp_step3: bf16 planes of the weights in operand order (bf16x3.h)
    const double* stats;   // (kStatBlocks, 2) advantage partial sums
    int n_stat_blocks;
    int D, A;
    int continuous;        // 1: Gaussian head (A action dims), 0: Categorical head (A logits, one action index)
    int static_tiles;      // diagnostic (AURPPO_STATIC_TILES): set s takes tiles s, s + S, s + 2S, ... -- a fixed summation order
    MlpLayout L;
    PpoHyper h;
};

// accumulator element e of a 32x32 MFMA block: (row, col) owned by this lane
__device__ __forceinline__ int acc_row(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }

// acc += A(32 x K) * B(K x 32); a_at(i,k) / b_at(k,j) fetch operand elements (LDS reads).
// K is a compile-time constant: the chain is fully unrolled in chunks of CH MFMAs whose 2*CH operand
// reads are issued one chunk ahead -- with one wave per SIMD nobody else hides the LDS latency (CH = 8);
// the two-set kernel runs two waves per SIMD and has half the registers, so it uses CH = 4.
template <int K, int CH = 8, bool FENCE = false, class FA, class FB>
__device__ __forceinline__ void mma32(f32x16& acc, FA a_at, FB b_at, int lane) {
    static_assert(K % (2 * CH) == 0, "K must be a multiple of the chunk depth");
    const int ij = lane & 31, kk = lane >> 5;
    float av[2][CH], bv[2][CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        av[0][u] = a_at(ij, 2 * u + kk);
        bv[0][u] = b_at(2 * u + kk, ij);
    }
#pragma unroll
    for (int c = 0; c < K / (2 * CH); ++c) {
        if (c + 1 < K / (2 * CH)) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                av[(c + 1) & 1][u] = a_at(ij, 2 * CH * (c + 1) + 2 * u + kk);
                bv[(c + 1) & 1][u] = b_at(2 * CH * (c + 1) + 2 * u + kk, ij);
            }
        }
#if AURPPO_MMA_PRIO
        __builtin_amdgcn_s_setprio(AURPPO_MMA_PRIO);
#endif
#pragma unroll
        for (int u = 0; u < CH; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][u], bv[c & 1][u], acc, 0, 0, 0);
#if AURPPO_MMA_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        // FENCE pins the pipeline depth to what is written here: without it the scheduler hoists every operand
        // read of the unrolled chain to the top, which costs ~2K registers the two-set kernel does not have.
        if (FENCE) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int K, int CH = 8, class FA, class FB>
__device__ __forceinline__ void mma32(f32x16& acc, FA a_at, FB b_at) {
    mma32<K, CH, false>(acc, a_at, b_at, (int)(threadIdx.x & 63));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16x16 output tile: acc += A(16 x K) * B(K x 16) on v_mfma_f32_16x16x4_f32 (lane l: A[l&15][l>>4],
// B[l>>4][l&15]; C: col = l&15, row = 4*(l>>4) + reg).  Two interleaved accumulators hide the 40-cycle
// dependent latency behind the 32-cycle issue interval; operands are read one 8-MFMA chunk ahead.
template <int K, bool FENCE = false, class FA, class FB>
__device__ __forceinline__ f32x4 mma16(FA a_at, FB b_at, int lane) {
    static_assert(K % 32 == 0, "K must be a multiple of 32");
    const int ij = lane & 15, kk = lane >> 4;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float av[2][8], bv[2][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        av[0][u] = a_at(ij, 4 * u + kk);
        bv[0][u] = b_at(4 * u + kk, ij);
    }
#pragma unroll
    for (int c = 0; c < K / 32; ++c) {
        if (c + 1 < K / 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[(c + 1) & 1][u] = a_at(ij, 32 * (c + 1) + 4 * u + kk);
                bv[(c + 1) & 1][u] = b_at(32 * (c + 1) + 4 * u + kk, ij);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c & 1][u], bv[c & 1][u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c & 1][u + 1], bv[c & 1][u + 1], acc1, 0, 0, 0);
        }
        if (FENCE) __builtin_amdgcn_sched_barrier(0);
    }
    return acc0 + acc1;
}

template <int K, class FA, class FB>
__device__ __forceinline__ f32x4 mma16(FA a_at, FB b_at) {
    return mma16<K, false>(a_at, b_at, (int)(threadIdx.x & 63));
}

// tanh(x) = 1 - 2 / (e^{2x} + 1): v_exp + v_rcp, absolute error ~1e-7 everywhere (saturates cleanly)
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// tanh(acc + bias) with the bias folded into the exponent's multiply: e^{2 (acc + bias)} = 2^{acc * c + bias * c}, c = 2 log2(e) -- one
// fused multiply-add in front of v_exp instead of an add and a multiply (bc = bias * kTanhC, formed once per block)
constexpr float kTanhC = 2.8853900817779268f;
__device__ __forceinline__ float tanh_fast_fma(float acc, float bc) {
    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(acc, kTanhC, bc));
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// All-reduce over aligned groups of 8 lanes with DPP moves (no LDS traffic): lane i <- i ^ 7 (row_half_mirror),
// then i ^ 1 and i ^ 2 (quad_perm) -- together every lane has combined all 8.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum8(float v) {
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    return v;
}
__device__ __forceinline__ float max8(float v) {
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    return v;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.0f;
    return z;
}

// mlp2.hip: the two-tile-set variant of K7 (8 waves per workgroup); same arguments, same slab / loss_part outputs.
size_t mlp_step2_lds_bytes();
int launch_mlp_step2(const MlpArgs& a, int grid, hipStream_t s);
// mlp3.hip: the same step on bf16 MFMAs over three-way bf16 splits (AURPPO_K7_VARIANT=3); its operand-order weight copies
size_t mlp_step3_lds_bytes();
size_t mlp_step3_wop_bytes();
int launch_mlp3_prep(const float* params, const MlpLayout& L, int D, void* wop3, hipStream_t s);
int launch_mlp_step3(const MlpArgs& a, int grid, hipStream_t s);
// mlp.hip: k_mlp_reduce alone (grads[p] = fixed-order sum over n_slabs slabs, loss scalars folded) -- mlp_wide.hip's tail.
// sq_part / step_dev != nullptr: also leave the clip's partial sums of squares (one per 64 parameters) and advance the
// Adam step count -- what launch_adam_tail (k_adam_chain without K7's extras: clip + Adam in one launch) then consumes.
int launch_mlp_reduce(const float* slabs, const double* loss_part, int n_slabs, int n_params, const PpoHyper& h, float* grads,
                      float* out_scalars, hipStream_t s, double* sq_part = nullptr, float* step_dev = nullptr,
                      unsigned* scratch_counter = nullptr,    // (with step_dev: a device word the kernel may clear)
                      double beta1 = 0.0, double beta2 = 0.0, double* bc_out = nullptr);   // bc_out: {1 - beta1^t, 1 - beta2^t} for launch_adam_tail
// One thread's share of a minibatch's advantage partial sums: elements first, first + step, ... in that order (the sums are the
// same bits as the plain loop's), four index loads and then four record loads in flight at a time -- the plain loop paid two
// dependent memory round trips per element.
__device__ __forceinline__ void adv_partial_sums(const float4* __restrict__ rec, int rec_stride, const int32_t* __restrict__ idx,
                                                 int M, int first, int step, double& s, double& q) {
    int i = first;
    for (; i + 3 * step < M; i += 4 * step) {
        const int j0 = idx[i], j1 = idx[i + step], j2 = idx[i + 2 * step], j3 = idx[i + 3 * step];
        const float x0 = rec[(size_t)j0 * rec_stride].y, x1 = rec[(size_t)j1 * rec_stride].y;
        const float x2 = rec[(size_t)j2 * rec_stride].y, x3 = rec[(size_t)j3 * rec_stride].y;
        s += (double)x0; q += (double)x0 * (double)x0;
        s += (double)x1; q += (double)x1 * (double)x1;
        s += (double)x2; q += (double)x2 * (double)x2;
        s += (double)x3; q += (double)x3 * (double)x3;
    }
    for (; i < M; i += step) {
        const double x = (double)rec[(size_t)idx[i] * rec_stride].y;
        s += x;
        q += x * x;
    }
}

// K7w's operand-order copies of the hidden layers (mlp_wide.hip: [net][layer][fwd | bwd][4 x 4 blocks][lane][16 k-steps]): where the
// optimizer launch drops an updated weight so that the next K7w launch needs no prepare pass.  wop == nullptr: nothing to refresh.
struct WideCopies {
    int w[2][3];       // float offsets of the hidden layers' weights in the bucket
    int NL, Hd, D;
    float* wop;
    unsigned short* wop3;   // != nullptr: k_mlpw3_step's bf16-plane copies are the ones to refresh (mlp_wide.hip: w3::wop_index)
};
// next_idx != nullptr: the launch also forms the next minibatch's advantage partial sums (stats: (kStatBlocks, 2) doubles).
int launch_adam_tail(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int n_params, const double* sq_part,
                     double max_norm, const float* lr_dev, const float* step_dev, double beta1, double beta2, double eps,
                     float* out_norm, hipStream_t s, const WideCopies* wide = nullptr, const float4* rec = nullptr, int rec_stride = 1,
                     const int32_t* next_idx = nullptr, int next_M = 0, double* stats = nullptr, const double* bc = nullptr);

}  // namespace aurppo_mlp
