// K7 on the bf16 matrix pipe (k_mlp_step3): the fused PPO minibatch step of mlp2.hip's k_mlp_step2 (src/ppo.py:219-267
// over src/models/actor_critic.py:8-51) with every fp32 matrix product formed from three-way bf16 splits of its operands
// (bf16x3.h: six v_mfma_f32_32x32x16_bf16 per K = 16, fp32 accumulate, dropped terms <= 2^-25 of a product) instead of
// v_mfma_f32_32x32x2_f32: 4.4 k matrix-pipe cycles per wave and tile instead of 11.5 k.
//
// Same arguments, same outputs (per-workgroup gradient slabs + loss partials), same structure as k_mlp_step2: 8 waves =
// two free-running TILE SETS of 4 waves (actor / critic x two 32-column halves), the eight phases of a 32-row tile
//     S (land tile)  F1  F2  F3 (head)  L (loss lanes)  B1  B2  B3
// separated by per-set / per-net software barriers.  What changes is where operands live:
//   * activations are LDS IMAGES of bf16 planes (bf16x3.h): X row-major as it arrives from HBM, H1 / H2 (later dZ1 / dZ2,
//     in place) feature-major, written straight from the accumulators with 8-byte stores; an operand fragment is one
//     ds_read_b128 along a row or two ds_read_b64_tr_b16 across rows -- 6 to 12 LDS instructions per six MFMAs where the
//     fp32 kernel issued 16 ds_read_b32 per eight;
//   * W1 and W2 never enter LDS: the preparation / optimizer kernels keep bf16-plane copies in MFMA B-operand order
//     (forward: W, backward: W^T slices; 144 KB, L2-resident) and a wave streams its 12 KB slice one phase ahead into 48
//     registers that W1 (B3 -> F1), W2 forward (F1 -> F2) and W2 backward (B1 -> B2) take turns in; only the head's W3 sits
//     in LDS (an image read both ways);
//   * the head runs on v_mfma_f32_16x16x32_bf16 (16 rows per wave), its backward on one K = 16 step.
#include <stdlib.h>

#pragma clang fp contract(fast)
#include "bf16x3.h"
#include "mlp_common.h"

using namespace aurppo_mlp;
using namespace bf3;

// Diagnostic build only (tools/mlp_stamps.py): wave 0 of each set accumulates, per phase, the cycles it spent working
// (slot k) and waiting at the phase's barrier (slot 8 + k) in LDS; dumped to the workspace at the end.
#ifdef AURPPO_MLP_STAMPS
#define STAMP3(k)                                                              \
    do {                                                                       \
        if (w == 0 && lane == 0) {                                             \
            const unsigned long long t__ = __builtin_readcyclecounter();       \
            s_stamp[set][k] += t__ - st_last;                                  \
            st_last = t__;                                                     \
        }                                                                      \
    } while (0)
#define XSTAMP(k)                                                              \
    do {                                                                       \
        if (w == 0 && lane == 0) {                                             \
            const unsigned long long t__ = __builtin_readcyclecounter();       \
            s_xstamp[set][k] += t__ - st_last;                                 \
        }                                                                      \
    } while (0)
#else
#define STAMP3(k) do { } while (0)
#define XSTAMP(k) do { } while (0)
#endif

namespace {

constexpr int kThreads3 = 512;
constexpr int kSetThreads = 256;
#ifndef AURPPO_BAR_SLEEP
#define AURPPO_BAR_SLEEP 1
#endif

// ---- dynamic LDS carve-up (bytes).  Shared by both sets:
constexpr int kW3Row = 128, kW3Plane = AP * kW3Row, kW3Net = 3 * kW3Plane;       // W3 image [a 16][i 64] per net
constexpr int oW3 = 0;                               // [2 nets][3 planes][16][128 B]
constexpr int oB1 = oW3 + 2 * kW3Net;                // float [2][H]
constexpr int oB2 = oB1 + 4 * 2 * H;                 // float [2][H]
constexpr int oB3 = oB2 + 4 * 2 * H;                 // float [2][AP]
constexpr int oLs = oB3 + 4 * 2 * AP;                // float [AP]
constexpr int oIvar = oLs + 4 * AP;                  // float [AP]
constexpr int kSharedBytes = oIvar + 4 * AP;
// per set:
constexpr int kDoRow = 64, kDoPlane = AP * kDoRow, kDoNet = 3 * kDoPlane;         // dOut image [a 16][s 32] per net
constexpr int pX = 0;                                // X image, 3 planes
constexpr int pH1 = pX + 3 * kXPlane;                // [2 nets] F image, 3 planes each; later dZ1
constexpr int pH2 = pH1 + 2 * 3 * kFPlane;           // [2 nets]; later dZ2
constexpr int pDo = pH2 + 2 * 3 * kFPlane;           // [2 nets] dOut image
constexpr int pOut = pDo + 2 * kDoNet;               // float [2][R][LDO] head outputs
constexpr int pRec = pOut + 4 * 2 * R * LDO;         // float4[R]
constexpr int pSrc = pRec + 16 * R;                  // int[R]
constexpr int pIdx = pSrc + 4 * R;                   // int[2][R]
constexpr int kSetBytes = pIdx + 4 * 2 * R;
static_assert(kSharedBytes % 16 == 0 && pH1 % 16 == 0 && pH2 % 16 == 0 && pDo % 16 == 0 && pOut % 16 == 0 && pRec % 16 == 0 &&
              kSetBytes % 16 == 0, "16-byte alignment of the images");
constexpr int kAccRegs = 72;                         // gW1 (32) + gW2 (32) + gW3 (2 x 4) per lane
constexpr int kDynBytes = kSharedBytes + 2 * kSetBytes;
// the hand-over at the end reuses the (dead) tile memory: parked accumulators, then the small column sums
constexpr int kParkBytes = 4 * 4 * kAccRegs * kWave;
constexpr int kSmallOff = kParkBytes;                // float [8 waves][8][5]
constexpr int kGbOff = kSmallOff + 4 * 8 * 8 * 5;    // float [4 roles][2][32]
static_assert(kGbOff + 4 * 4 * 2 * 32 <= kDynBytes, "hand-over scratch must fit the dead tiles");

// Epilogues, four values (one 8-byte store per plane) at a time so that nothing but the accumulator is live across them.
// tanh(acc + bias) of a 32x32 block into an F image:
// om[e] = 1 - tanh^2 of the same element, kept in registers for the backward pass (dz_from_regs): re-read from the image's planes it
// cost three unpacks and two adds per value to join and twelve LDS reads per block (the register file has had room for the 2 x 16
// values since the fragment addresses stopped being re-derived, bf16x3.h)
__device__ __forceinline__ void tanh_store(char* img, int f0, const f32x16& acc, float bias, int lane, float (&om)[16]) {
    const int f = f0 + (lane & 31), h = lane >> 5;
    const float bc = bias * kTanhC;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        unsigned a0, a1, a2, b0, b1, b2;
        const float t0 = tanh_fast_fma(acc[4 * gq + 0], bc), t1 = tanh_fast_fma(acc[4 * gq + 1], bc);
        const float t2 = tanh_fast_fma(acc[4 * gq + 2], bc), t3 = tanh_fast_fma(acc[4 * gq + 3], bc);
        om[4 * gq + 0] = 1.0f - t0 * t0; om[4 * gq + 1] = 1.0f - t1 * t1;
        om[4 * gq + 2] = 1.0f - t2 * t2; om[4 * gq + 3] = 1.0f - t3 * t3;
        split3(t0, t1, a0, a1, a2);
        split3(t2, t3, b0, b1, b2);
        const int o = foff(f, 4 * h) ^ (gq << 4);       // = foff(f, 8 gq + 4 h)
        *reinterpret_cast<u32x2*>(img + 0 * kFPlane + o) = u32x2{a0, b0};
        *reinterpret_cast<u32x2*>(img + 1 * kFPlane + o) = u32x2{a1, b1};
        *reinterpret_cast<u32x2*>(img + 2 * kFPlane + o) = u32x2{a2, b2};
    }
}
// dZ = dH * (1 - h^2), (1 - h^2) from the forward pass's registers, written over the block of h in the image; returns the lane's column sum
__device__ __forceinline__ float dz_from_regs(char* img, int f0, const f32x16& dh, const float (&om)[16], int lane) {
    const int f = f0 + (lane & 31), h = lane >> 5;
    float colsum = 0.0f;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const float d0 = dh[4 * gq + 0] * om[4 * gq + 0], d1 = dh[4 * gq + 1] * om[4 * gq + 1];
        const float d2 = dh[4 * gq + 2] * om[4 * gq + 2], d3 = dh[4 * gq + 3] * om[4 * gq + 3];
        colsum += (d0 + d1) + (d2 + d3);
        unsigned a0, a1, a2, b0, b1, b2;
        split3(d0, d1, a0, a1, a2);
        split3(d2, d3, b0, b1, b2);
        const int o = foff(f, 4 * h) ^ (gq << 4);
        *reinterpret_cast<u32x2*>(img + 0 * kFPlane + o) = u32x2{a0, b0};
        *reinterpret_cast<u32x2*>(img + 1 * kFPlane + o) = u32x2{a1, b1};
        *reinterpret_cast<u32x2*>(img + 2 * kFPlane + o) = u32x2{a2, b2};
    }
    return colsum;
}
__device__ __forceinline__ void lds_add(double* p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(kThreads3, 1) void k_mlp_step3(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ double s_red[2][kThreads3 / kWave];
    __shared__ double s_loss[2][6][R];          // per set, per quantity, per tile row: running sums
    __shared__ float s_mean, s_std;
    __shared__ int s_next[2][2];
    __shared__ int s_first[2];
    __shared__ int s_grab;
    __shared__ int s_bar[2];
    __shared__ int s_pbar[2][2];
#ifdef AURPPO_MLP_STAMPS
    __shared__ unsigned long long s_stamp[2][16];
    __shared__ unsigned long long s_xstamp[2][8];   // finer stamps inside F1 / B3 (diagnostic)
    unsigned long long st_last = 0;
    const unsigned long long rt_entry = wall_clock64();
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int set = wave >> 2, w = wave & 3, st = tid & (kSetThreads - 1);
    const int net = w >> 1, cb = w & 1;
    const int wi = net * 2 + cb;
    const int D = a.D, A = a.A;
    const int AW = a.continuous ? a.A : 1;
    const int out_dim[2] = {A, 1};

    char* const sW3 = lds + oW3 + net * kW3Net;                     // this wave's net
    float* const sB1 = reinterpret_cast<float*>(lds + oB1);
    float* const sB2 = reinterpret_cast<float*>(lds + oB2);
    float* const sB3 = reinterpret_cast<float*>(lds + oB3);
    float* const sLs = reinterpret_cast<float*>(lds + oLs);
    float* const sIvar = reinterpret_cast<float*>(lds + oIvar);
    char* const base = lds + kSharedBytes + set * kSetBytes;
    char* const sX = base + pX;
    char* const sH1 = base + pH1 + net * 3 * kFPlane;
    char* const sH2 = base + pH2 + net * 3 * kFPlane;
    char* const sDo = base + pDo + net * kDoNet;
    float* const sOut = reinterpret_cast<float*>(base + pOut);
    float4* const sRec = reinterpret_cast<float4*>(base + pRec);
    int* const sSrc = reinterpret_cast<int*>(base + pSrc);
    int* const sIdx = reinterpret_cast<int*>(base + pIdx);

    const int n_tiles = (a.h.M + R - 1) / R;
    unsigned* const tile_counter = a.tile_counter;
    int zero_off = 0;
    asm volatile("" : "+v"(zero_off));
    int n_idx = -1;
    bool n_ok = false;
    auto load_idx = [&](int tile, int st) -> int {
        const int m = tile * R + (st & (R - 1));
        const bool ok = tile < n_tiles && m < a.h.M;
        const int v = a.idx[ok ? m : 0];
        return ok ? v : -1;
    };
    int t1 = 0, t2 = 0, t3_raw = 0;
    const int n_sets = 2 * gridDim.x, my_set = 2 * blockIdx.x + set;
    const bool stat = a.static_tiles != 0;      // diagnostic: static stride, fixed summation order (see mlp2.hip)
    // A minibatch of fewer than four tiles per set (BASELINE config 4's shard: one): the first three tiles of a set are its own by
    // number, so every set starts without waiting for the counter -- the indices of all three, the first one's rows and the
    // statistics are requested before anything else in the launch (prologue 9.1 -> 4.4 us at M = 16 384, tools/mlp_stamps.py).
    // Larger ones: a workgroup draws its first eight tiles when it STARTS (beside the shuffle kernels some start late, and a
    // tile fixed to a late workgroup waits for it: 2.15 -> 2.24 ms per update at M = 131 072 with the fixed start), the rest one
    // by one from the counter.
    const bool fixed_start = !stat && n_tiles < 4 * n_sets;
    const bool early = fixed_start || stat;
    const int dyn_base = fixed_start ? 3 * n_sets : 0;

    const bool vec4 = (D % 4 == 0) && ((reinterpret_cast<size_t>(a.obs) & 15) == 0);
    const float* const act_base = a.actions ? a.actions : reinterpret_cast<const float*>(a.rec) + 4;
    const int act_stride = a.actions ? AW : 16;
    float xr[8];
    float ar[2], act_cur[2] = {0.0f, 0.0f};
    float4 p_rec = make_float4(0.f, 0.f, 0.f, 0.f);
    int p_src = -1;
    bool x_ok[2] = {false, false}, a_ok[2] = {false, false};
    // rows of a tile, given the sample indices of the rows this thread touches (s0 / s1: its two float4 rows, sl: its row of the
    // 8-lanes-per-row layout, sp: wave 0's record row)
    auto prefetch_src = [&](int s0, int s1, int sl_row, int sp, int st) {
        const int lj = st & 7;
        if (vec4) {
            const int c4 = (st & 15) * 4;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int src = p == 0 ? s0 : s1;
                x_ok[p] = src >= 0 && c4 < D;
                const float4 v = *reinterpret_cast<const float4*>(a.obs + (x_ok[p] ? (size_t)src * D + c4 : (size_t)0));
                xr[4 * p + 0] = v.x; xr[4 * p + 1] = v.y; xr[4 * p + 2] = v.z; xr[4 * p + 3] = v.w;
            }
        } else {
            const int src = sl_row;
            x_ok[0] = src >= 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) xr[u] = a.obs[(src >= 0 && lj + 8 * u < D) ? (size_t)src * D + lj + 8 * u : (size_t)0];
        }
        {
            const int src = sl_row;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                a_ok[u] = src >= 0 && lj + 8 * u < AW;
                ar[u] = act_base[a_ok[u] ? (size_t)src * act_stride + lj + 8 * u : (size_t)0];
            }
        }
        if (w == 0) {
            p_src = sp;
            p_rec = a.rec[(size_t)(p_src >= 0 ? p_src : 0) * a.rec_stride];
        }
    };
    auto prefetch = [&](const int* sidx, int st) {
        prefetch_src(sidx[st >> 4], sidx[(st >> 4) + 16], sidx[st >> 3], w == 0 ? sidx[st & (R - 1)] : -1, st);
    };
    // ---- first of all: the sample indices of this set's first tile (tile number = set number), straight from the index list;
    // its rows are requested as soon as they are here, beside the staging loads below.  (They used to wait for the tile grab,
    // the staging and two workgroup barriers: 9.1 us of prologue before a 12 us tile at M = 16 384, tools/mlp_stamps.py.)
    int e0 = -1, e1 = -1, el = -1, ep = -1, i1_early = -1;
    if (early) {
        auto gidx = [&](int row) -> int {
            const int m = my_set * R + row;
            const bool ok = my_set < n_tiles && m < a.h.M;
            const int v = a.idx[ok ? m : 0];
            return ok ? v : -1;
        };
        e0 = gidx(st >> 4);
        e1 = gidx((st >> 4) + 16);
        el = gidx(st >> 3);
        ep = gidx(st & (R - 1));
        if (w == 0) {
            i1_early = load_idx(my_set + n_sets, st);
            n_idx = load_idx(my_set + 2 * n_sets, st);
        }
    }
    // (the advantage partial sums too: nothing below depends on anything but the launch arguments)
    double st_s = 0.0, st_q = 0.0;
    for (int b = tid; b < a.n_stat_blocks; b += kThreads3) {
        st_s += a.stats[2 * b];
        st_q += a.stats[2 * b + 1];
    }
    // ---- this wave's weight slices, streamed in operand order: 12 fragments of 16 B per lane and matrix
    // (a wave-uniform base in scalar registers + the lane's 32-bit byte offset + an immediate per fragment: written as 36
    // 64-bit vector addresses they were formed once, hoisted out of the tile loop, spilled, and every load then waited behind
    // a scratch reload -- 3.5 k cycles per tile to ISSUE twelve loads; profiles/r03/k7_stamps_v3.txt)
    const char* const wrole0 = reinterpret_cast<const char*>(a.wop3) + (size_t)wi * kWopMats * kWopKs * 3 * 1024;
    const char* wrole = wrole0;
    int lane16 = lane * 16;
    bf16x8 wreg[12];
    auto load_w = [&](int mt) {
        const char* m = wrole + mt * (12 * 1024);
#pragma unroll
        for (int q = 0; q < 12; ++q) wreg[q] = *reinterpret_cast<const bf16x8*>(m + q * 1024 + lane16);
    };

    // ---- stage what stays in LDS for the whole launch: W3 images (both nets), biases, log-std
    {
        // every global load first, then the stores: the loads go out together
        float w3r[2][AP * H / kThreads3], b1r = 0.f, b2r = 0.f, b3r = 0.f, lsr = 0.f;
        static_assert(AP * H % kThreads3 == 0 && 2 * H <= kThreads3, "staging slots");
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int q = 0; q < AP * H / kThreads3; ++q) {
                const int e = tid + q * kThreads3, o = e / H;
                w3r[n][q] = a.params[o < out_dim[n] ? a.L.w3[n] + e : a.L.w3[n]];
            }
        if (tid < 2 * H) {
            const int e = tid % H;
            b1r = a.params[(tid < H ? a.L.b1[0] : a.L.b1[1]) + e];
            b2r = a.params[(tid < H ? a.L.b2[0] : a.L.b2[1]) + e];
        }
        if (tid < 2 * AP) {
            const int e = tid % AP, od = tid < AP ? A : 1, b3 = tid < AP ? a.L.b3[0] : a.L.b3[1];
            b3r = a.params[e < od ? b3 + e : b3];
        }
        if (tid < AP) lsr = a.params[(a.continuous && tid < A) ? a.L.logstd + tid : a.L.w2[0]];
        int grab_raw = 0;
        if (!early && tid == 0) grab_raw = (int)atomicAdd(tile_counter + zero_off, 8u);
        load_w(0);                                   // W1 slice of the first tile
        if (early) prefetch_src(e0, e1, el, ep, st); // (the index loads are the oldest in flight: only they are waited for here)
        if (tid == 0) s_grab = grab_raw;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int q = 0; q < AP * H / kThreads3; ++q) {
                const int e = tid + q * kThreads3, o = e / H, i = e % H;
                store_plain1(lds + oW3 + n * kW3Net, kW3Row, kW3Plane, o, i, o < out_dim[n] ? w3r[n][q] : 0.0f);
            }
        if (tid < 2 * H) {
            sB1[tid] = b1r;
            sB2[tid] = b2r;
        }
        if (tid < 2 * AP) sB3[tid] = (tid % AP) < (tid < AP ? A : 1) ? b3r : 0.0f;
        if (tid < AP) {
            const float ls = (a.continuous && tid < A) ? lsr : 0.0f;
            const float sd = expf(ls);
            sLs[tid] = ls;
            sIvar[tid] = 1.0f / (sd * sd);
        }
    }
    // X image (columns >= D stay zero for the whole launch) and dOut images (the critic's rows 1.. stay zero) of this set
    {
        u32x4* z = reinterpret_cast<u32x4*>(base + pX);
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (int e = st; e < 3 * kXPlane / 16; e += kSetThreads) z[e] = zero;
        u32x4* zd = reinterpret_cast<u32x4*>(base + pDo);
        for (int e = st; e < 2 * kDoNet / 16; e += kSetThreads) zd[e] = zero;
    }
    for (int e = tid; e < 2 * 6 * R; e += kThreads3) (&s_loss[0][0][0])[e] = 0.0;
    __syncthreads();
    if (w == 0) {
        const int base_t = early ? my_set : s_grab + 4 * set;
        t1 = early ? my_set + n_sets : base_t + 1;
        t2 = early ? my_set + 2 * n_sets : base_t + 2;
        if (stat) {
            t3_raw = my_set + 3 * n_sets;
        } else if (!early) {
            t3_raw = base_t + 3;
        } else if (lane == 0) {
            t3_raw = (int)atomicAdd(tile_counter + zero_off, 1u);      // first used in the first tile's loss phase
        }
        if (lane == 0) {
            s_first[set] = base_t < n_tiles ? 1 : 0;
            s_bar[set] = 0;
            s_pbar[set][0] = 0;
            s_pbar[set][1] = 0;
        }
        const int i0 = early ? ep : load_idx(base_t, st), i1 = early ? i1_early : load_idx(t1, st);
        if (!early) n_idx = load_idx(t2, st);
        n_ok = n_idx >= 0;
        if (st < R) {
            sIdx[st] = i0;
            sIdx[R + st] = i1;
        }
    }
    // ---- minibatch advantage statistics from the partials (same order in every workgroup)
    {
        const double ts = block_sum<kThreads3 / kWave>(st_s, s_red[0]);
        const double tq = block_sum<kThreads3 / kWave>(st_q, s_red[1]);
        if (tid == 0) {
            const double m = ts / (double)a.h.M;
            double var = (tq - ts * m) / (double)(a.h.M - 1);
            if (var < 0.0) var = 0.0;
            s_mean = (float)m;
            s_std = (float)sqrt(var);
        }
    }
    __syncthreads();
    auto uniform = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    const float mean = uniform(s_mean), denom = uniform(s_std + 1e-8f);
    const float invM = uniform(1.0f / (float)a.h.M);
    const float g_ent = uniform(-a.h.ent_coef * invM);
    if (!early) prefetch(sIdx, st);
    float ent_sum = 0.0f;
    if (a.continuous)
        for (int k = 0; k < A; ++k) ent_sum += (0.5f + 0.9189385332046727f) + sLs[k];
    const float ent_gauss = uniform(ent_sum);

    // ---- persistent accumulators (registers); layouts as in k_mlp_step2
    f32x16 gW1[2] = {zero16(), zero16()};
    f32x16 gW2[2] = {zero16(), zero16()};
    f32x4 gW3[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float gb1 = 0.0f, gb2 = 0.0f;
    float g_b3a[2] = {0.0f, 0.0f}, g_ls[2] = {0.0f, 0.0f}, g_b3c = 0.0f;

    __syncthreads();

    int bar_gen = 0;
    auto set_bar = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) (void)__hip_atomic_fetch_add(&s_bar[set], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        bar_gen += 4;
        while (__hip_atomic_load(&s_bar[set], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - bar_gen < 0)
            __builtin_amdgcn_s_sleep(AURPPO_BAR_SLEEP);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    int pbar_gen = 0;
    auto pair_bar = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) (void)__hip_atomic_fetch_add(&s_pbar[set][net], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pbar_gen += 2;
        while (__hip_atomic_load(&s_pbar[set][net], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - pbar_gen < 0)
            __builtin_amdgcn_s_sleep(AURPPO_BAR_SLEEP);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto wfrag = [&](int ks) {
        Frag3 f;
        f.p[0] = wreg[3 * ks + 0];
        f.p[1] = wreg[3 * ks + 1];
        f.p[2] = wreg[3 * ks + 2];
        return f;
    };
    const int nks1 = (D + 15) >> 4;              // k-steps of layer 1 that hold anything (X and W1 are zero beyond D)
#ifdef AURPPO_MLP_STAMPS
    if (tid < 32) (&s_stamp[0][0])[tid] = 0ull;
    if (tid < 16) (&s_xstamp[0][0])[tid] = 0ull;
    __syncthreads();
    st_last = __builtin_readcyclecounter();
    const unsigned long long clk0 = st_last, rt0 = wall_clock64();
#endif

    float om1[16], om2[16];                     // 1 - H1^2, 1 - H2^2 of this wave's blocks, from the forward to the backward phases
    for (int it = 0; s_first[set] != 0; ++it) {
        int ln = lane, sl = st;
        asm volatile("" : "+v"(ln), "+v"(sl));    // opaque per-tile copies: LDS addresses are re-derived inside the phases
        {
            int wz = 0;                            // ... and the weight base (an opaque zero added to it keeps it a global pointer)
            asm volatile("" : "+s"(wz));
            wrole = wrole0 + wz;
            lane16 = ln * 16;
        }
        const int lr = sl >> 3, lj = sl & 7;
        {   // ---- S: land the prefetched tile as bf16 planes
            if (vec4) {
                const int c4 = (sl & 15) * 4;
                if (c4 < D) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const bool ok = x_ok[p];
                        store_x4(sX, (sl >> 4) + 16 * p, c4, ok ? xr[4 * p + 0] : 0.0f, ok ? xr[4 * p + 1] : 0.0f,
                                 ok ? xr[4 * p + 2] : 0.0f, ok ? xr[4 * p + 3] : 0.0f);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (lj + 8 * u < D) store_x1(sX, lr, lj + 8 * u, x_ok[0] ? xr[u] : 0.0f);
            }
            act_cur[0] = a_ok[0] ? ar[0] : 0.0f;
            act_cur[1] = a_ok[1] ? ar[1] : 0.0f;
            if (sl < R) {
                sSrc[sl] = p_src;
                sRec[sl] = p_rec;
                sIdx[(it & 1) * R + sl] = n_ok ? n_idx : -1;
            }
            if (w == 0 && sl == 0) s_next[set][it & 1] = t1 < n_tiles ? 1 : 0;
        }
        STAMP3(0);
        set_bar();
        STAMP3(8);
        {   // ---- F1: H1 = tanh(X W1^T + b1): A = rows of the X image, B = the streamed W1 slice
            f32x16 acc = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                if (ks < nks1) acc = mma32x3(x_rows(sX, ks, ln), wfrag(ks), acc);
            __builtin_amdgcn_sched_barrier(0);
            XSTAMP(0);
            load_w(1);                                   // W2 (forward) arrives behind the epilogue and the barrier
            XSTAMP(1);
            tanh_store(sH1, cb * 32, acc, sB1[net * H + cb * 32 + (ln & 31)], ln, om1);
        }
        STAMP3(1);
        pair_bar();
        STAMP3(9);
        {   // ---- F2: H2 = tanh(H1 W2^T + b2): A = the H1 image read across its rows
            f32x16 acc = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = mma32x3(f_cols(sH1, ks, ln), wfrag(ks), acc);
            tanh_store(sH2, cb * 32, acc, sB2[net * H + cb * 32 + (ln & 31)], ln, om2);
        }
        STAMP3(2);
        pair_bar();
        STAMP3(10);
        {   // ---- F3: head, 16 rows per wave on 16x16x32: out[s][a] = H2[s][:] . W3[a][:] + b3[a]
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                acc = mma16x3(f_cols16(sH2, 16 * cb, ks, ln), plain_rows(sW3, kW3Row, kW3Plane, ln & 15, 32 * ks + 8 * (ln >> 4)), acc);
            const int col = ln & 15;
            const float bias = sB3[net * AP + col];
#pragma unroll
            for (int e = 0; e < 4; ++e) sOut[(net * R + cb * 16 + 4 * (ln >> 4) + e) * LDO + col] = acc[e] + bias;
        }
        STAMP3(3);
        set_bar();
        STAMP3(11);
        {   // ---- L: distribution + PPO terms, 8 lanes per row; the head gradients go out as bf16 planes ([a][s] image)
            if (w == 0) {
                const int t3 = dyn_base + __builtin_amdgcn_readfirstlane(t3_raw);
                if (stat) t3_raw = t3 + n_sets;
                else if (ln == 0) t3_raw = (int)atomicAdd(tile_counter + zero_off, 1u);
                const int m = t3 * R + (sl & (R - 1));
                n_ok = t3 < n_tiles && m < a.h.M;
                n_idx = a.idx[n_ok ? m : 0];
                t1 = t2;
                t2 = t3;
            }
            const float* mu = sOut + (0 * R + lr) * LDO;
            const float* vv = sOut + (1 * R + lr) * LDO;
            const bool real = sSrc[lr] >= 0;
            const float4 rc = sRec[lr];
            const float v_new = vv[0];
            const int k0 = lj, k1 = lj + 8;
            const float m0 = mu[k0], m1 = mu[k1];
            float logp = 0.0f, ent = 0.0f, d0, d1;
            PpoSample t;
            if (a.continuous) {
                const float iv0 = k0 < A ? sIvar[k0] : 0.0f, iv1 = k1 < A ? sIvar[k1] : 0.0f;
                const float z0 = act_cur[0] - m0, z1 = act_cur[1] - m1;
                if (k0 < A) logp += (-(z0 * z0) * (0.5f * iv0) - sLs[k0]) - 0.9189385332046727f;
                if (k1 < A) logp += (-(z1 * z1) * (0.5f * iv1) - sLs[k1]) - 0.9189385332046727f;
                logp = sum8(logp);
                ent = ent_gauss;
                t = ppo_sample(logp, rc.x, rc.y, v_new, rc.w, rc.z, mean, denom, invM, a.h);
                d0 = (real && k0 < A) ? t.g_logp * (z0 * iv0) : 0.0f;
                d1 = (real && k1 < A) ? t.g_logp * (z1 * iv1) : 0.0f;
                if (real && k0 < A) g_ls[0] += t.g_logp * (z0 * z0 * iv0 - 1.0f) + g_ent;
                if (real && k1 < A) g_ls[1] += t.g_logp * (z1 * z1 * iv1 - 1.0f) + g_ent;
            } else {
                const int ai = (int)sum8(act_cur[0]);
                const float z0 = k0 < A ? m0 : -INFINITY, z1 = k1 < A ? m1 : -INFINITY;
                const float mx = max8(fmaxf(z0, z1));
                const float se = sum8((k0 < A ? expf(z0 - mx) : 0.0f) + (k1 < A ? expf(z1 - mx) : 0.0f));
                const float lse = mx + logf(se);
                const float lp0 = k0 < A ? z0 - lse : 0.0f, lp1 = k1 < A ? z1 - lse : 0.0f;
                const float p0 = k0 < A ? expf(lp0) : 0.0f, p1 = k1 < A ? expf(lp1) : 0.0f;
                ent = sum8(-(p0 * lp0) - p1 * lp1);
                logp = sum8((k0 == ai ? lp0 : 0.0f) + (k1 == ai ? lp1 : 0.0f));
                t = ppo_sample(logp, rc.x, rc.y, v_new, rc.w, rc.z, mean, denom, invM, a.h);
                d0 = (real && k0 < A) ? t.g_logp * ((k0 == ai ? 1.0f : 0.0f) - p0) + g_ent * (-p0 * (lp0 + ent)) : 0.0f;
                d1 = (real && k1 < A) ? t.g_logp * ((k1 == ai ? 1.0f : 0.0f) - p1) + g_ent * (-p1 * (lp1 + ent)) : 0.0f;
            }
            // rows of tile it+1 (its indices landed in LDS two tiles ago): issued here, landed at the next S -- five phases
            // cover the HBM latency, and the 16 registers they arrive in are not live through the forward phases
            prefetch(sIdx + ((it + 1) & 1) * R, sl);
            char* const doA = base + pDo;                 // actor's dOut image
            store_plain1(doA, kDoRow, kDoPlane, k0, lr, d0);
            store_plain1(doA, kDoRow, kDoPlane, k1, lr, d1);
            g_b3a[0] += d0;
            g_b3a[1] += d1;
            if (lj == 0) {
                const float gv = real ? t.g_v : 0.0f;
                store_plain1(doA + kDoNet, kDoRow, kDoPlane, 0, lr, gv);      // critic: row 0 of its image
                g_b3c += gv;
                if (real) {
                    double* L = &s_loss[set][0][lr];
                    lds_add(L + 0 * R, (double)t.pg); lds_add(L + 1 * R, (double)t.vl); lds_add(L + 2 * R, (double)ent);
                    lds_add(L + 3 * R, (double)t.okl); lds_add(L + 4 * R, (double)t.kl); lds_add(L + 5 * R, (double)t.cf);
                }
            }
        }
        STAMP3(4);
        set_bar();
        STAMP3(12);
        {   // ---- B1: dH2 = dOut W3 (one K = 16 step), dW3 += dOut^T H2, dZ2 in place over this wave's half of H2
            f32x16 acc = zero16();
            acc = mma32x3(plain_cols(sDo, kDoRow, kDoPlane, 0, 0, ln), plain_cols(sW3, kW3Row, kW3Plane, 0, cb * 32, ln), acc);
            {
                const Frag3 da = plain_rows(sDo, kDoRow, kDoPlane, ln & 15, 8 * (ln >> 4));
#pragma unroll
                for (int q = 0; q < 2; ++q) gW3[q] = mma16x3(da, f_rows16(sH2, cb * 32 + 16 * q, ln), gW3[q]);
            }
            float colsum = dz_from_regs(sH2, cb * 32, acc, om2, ln);
            colsum += __shfl_xor(colsum, 32, kWave);
            gb2 += colsum;
        }
        STAMP3(5);
        pair_bar();
        STAMP3(13);
        {   // ---- B2: dW2 += dZ2^T H1 (this wave's in-block), dH1 = dZ2 W2 -> dZ1 in place over this wave's half of H1
            load_w(2);                                   // W2 (backward) for dH1, behind the dW2 chains
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const Frag3 hb = f_rows(sH1, cb * 32, ks, ln);
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) gW2[ob] = mma32x3(f_rows(sH2, ob * 32, ks, ln), hb, gW2[ob]);
            }
            f32x16 acc = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = mma32x3(f_cols(sH2, ks, ln), wfrag(ks), acc);
            float colsum = dz_from_regs(sH1, cb * 32, acc, om1, ln);
            colsum += __shfl_xor(colsum, 32, kWave);
            gb1 += colsum;
        }
        STAMP3(6);
        pair_bar();
        STAMP3(14);
        {   // ---- B3: dW1 += dZ1^T X (this wave's 32 state columns), B = the X image read across its rows
            if (cb * 32 < D) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const Frag3 xb = x_cols(sX, ks, cb * 32, ln);
#pragma unroll
                    for (int ob = 0; ob < 2; ++ob) gW1[ob] = mma32x3(f_rows(sH1, ob * 32, ks, ln), xb, gW1[ob]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            XSTAMP(2);
            load_w(0);                                   // W1 slice for the next tile's F1
            XSTAMP(3);
        }
        STAMP3(7);
        set_bar();
        STAMP3(15);
        if (!s_next[set][it & 1]) break;
    }
#ifdef AURPPO_MLP_STAMPS
    const unsigned long long rt_loop_end = wall_clock64();
#endif
    __syncthreads();   // the hand-over below reuses the tile memory: both sets must have left the loop
#ifdef AURPPO_MLP_STAMPS
    const unsigned long long rt_both_done = wall_clock64();
    unsigned long long my_stamp = 0;
    if (tid < 32) my_stamp = (&s_stamp[0][0])[tid];
#endif

    int le = lane, se = st;
    asm volatile("" : "+v"(le), "+v"(se));
    // ---- hand-over: set 1 parks its accumulators in the (dead) tile memory, set 0 adds them and writes the slab
    float* const fl = reinterpret_cast<float*>(lds);
    float* park = fl + (size_t)wi * kAccRegs * kWave + le;
    float (*s_small)[8][5] = reinterpret_cast<float (*)[8][5]>(lds + kSmallOff);
    float (*s_gb)[2][32] = reinterpret_cast<float (*)[2][32]>(lds + kGbOff);
    float hs[5] = {g_b3a[0], g_b3a[1], g_ls[0], g_ls[1], (se & 7) == 0 ? g_b3c : 0.0f};
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        hs[q] += __shfl_xor(hs[q], 8, kWave);
        hs[q] += __shfl_xor(hs[q], 16, kWave);
        hs[q] += __shfl_xor(hs[q], 32, kWave);
    }
    if (le < 8) {
#pragma unroll
        for (int q = 0; q < 5; ++q) s_small[set * 4 + w][le][q] = hs[q];
    }
    if (set == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            park[(0 + e) * kWave] = gW1[0][e];
            park[(16 + e) * kWave] = gW1[1][e];
            park[(32 + e) * kWave] = gW2[0][e];
            park[(48 + e) * kWave] = gW2[1][e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            park[(64 + e) * kWave] = gW3[0][e];
            park[(68 + e) * kWave] = gW3[1][e];
        }
        if (le < 32) {
            s_gb[wi][0][le] = gb1;
            s_gb[wi][1][le] = gb2;
        }
    }
    __syncthreads();
    if (set == 0) {
        float* slab = a.slabs + (size_t)blockIdx.x * a.L.n_params;
        const int col = cb * 32 + (le & 31);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int o = ob * 32 + acc_row(e, le);
                if (col < D) slab[a.L.w1[net] + o * D + col] = gW1[ob][e] + park[(ob * 16 + e) * kWave];
                slab[a.L.w2[net] + o * H + col] = gW2[ob][e] + park[(32 + ob * 16 + e) * kWave];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = 4 * (le >> 4) + e, c = cb * 32 + (le & 15);   // 16x16 accumulator layout
            if (o < out_dim[net]) {
                slab[a.L.w3[net] + o * H + c] = gW3[0][e] + park[(64 + e) * kWave];
                slab[a.L.w3[net] + o * H + c + 16] = gW3[1][e] + park[(68 + e) * kWave];
            }
        }
        if (le < 32) {
            slab[a.L.b1[net] + col] = gb1 + s_gb[wi][0][le];
            slab[a.L.b2[net] + col] = gb2 + s_gb[wi][1][le];
        }
        if (w == 0) {
            if (le < AP) {
                const int j = le & 7, u = le >> 3;
                float b3 = 0.0f, dl = 0.0f;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) {
                    b3 += s_small[ww][j][u];
                    dl += s_small[ww][j][2 + u];
                }
                if (le < A) slab[a.L.b3[0] + le] = b3;
                if (a.continuous && le < A) slab[a.L.logstd + le] = dl;
            }
            if (le == 0) {
                float c = 0.0f;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) c += s_small[ww][0][4];
                slab[a.L.b3[1]] = c;
            }
            double* lp = a.loss_part + (size_t)blockIdx.x * 8;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const double x = wave_sum(s_loss[le >> 5][q][le & 31]);
                if (le == 0) lp[q] = x;
            }
            if (le == 0) {
                lp[6] = (double)mean;
                lp[7] = (double)s_std;
            }
        }
    }
#ifdef AURPPO_MLP_STAMPS
    if (tid < 32) a.stamps[(size_t)blockIdx.x * 40 + tid] = my_stamp;
    if (tid < 1) {
        a.stamps[(size_t)blockIdx.x * 40 + 39] = s_xstamp[0][0];
    }
    if (tid >= 64 && tid < 68) a.stamps[(size_t)256 * 40 + (size_t)blockIdx.x * 4 + (tid - 64)] = s_xstamp[0][tid - 64];
    if (tid == 0) {
        a.stamps[(size_t)blockIdx.x * 40 + 32] = __builtin_readcyclecounter() - clk0;
        a.stamps[(size_t)blockIdx.x * 40 + 33] = wall_clock64() - rt0;
        a.stamps[(size_t)blockIdx.x * 40 + 34] = rt_entry;
        a.stamps[(size_t)blockIdx.x * 40 + 35] = rt0 - rt_entry;
        a.stamps[(size_t)blockIdx.x * 40 + 36] = rt_loop_end - rt0;
        a.stamps[(size_t)blockIdx.x * 40 + 37] = rt_both_done - rt_loop_end;
        a.stamps[(size_t)blockIdx.x * 40 + 38] = wall_clock64() - rt_both_done;
    }
#endif
}

// wop3[...] = the bf16 planes of W1 / W2 of both nets in operand order (bf16x3.h), written destination-first so that the
// padding (state columns >= D) is zero without a clearing pass
__global__ __launch_bounds__(256) void k_mlp3_prep(const float* __restrict__ params, MlpLayout L, int D, unsigned short* __restrict__ wop3) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < 4 * kWopMats * kWopKs * 64 * 8; e += gridDim.x * 256) {
        const int j = e & 7, lane = (e >> 3) & 63, ks = (e >> 9) & 3, mt = (e >> 11) % kWopMats, wi = (e >> 11) / kWopMats;
        const int net = wi >> 1, cb = wi & 1;
        const int k = 16 * ks + 8 * (lane >> 5) + j, c = cb * 32 + (lane & 31);
        float v;
        if (mt == 0) v = k < D ? params[L.w1[net] + c * D + k] : 0.0f;          // W1[o = c][d = k]
        else if (mt == 1) v = params[L.w2[net] + c * H + k];                      // W2[o = c][i = k]
        else v = params[L.w2[net] + k * H + c];                                   // W2[o2 = k][i = c]
        unsigned p0, p1, p2;
        split3(v, 0.0f, p0, p1, p2);
        const int at = wop3_index(wi, mt, ks, 0, lane, j);
        wop3[at] = (unsigned short)p0;
        wop3[at + kWopBlock] = (unsigned short)p1;
        wop3[at + 2 * kWopBlock] = (unsigned short)p2;
    }
}

}  // namespace

namespace aurppo_mlp {

size_t mlp_step3_lds_bytes() { return (size_t)kDynBytes; }
size_t mlp_step3_wop_bytes() { return sizeof(unsigned short) * (size_t)kWopElems; }

int launch_mlp3_prep(const float* params, const MlpLayout& L, int D, void* wop3, hipStream_t s) {
    hipLaunchKernelGGL(k_mlp3_prep, dim3(24), dim3(256), 0, s, params, L, D, reinterpret_cast<unsigned short*>(wop3));
    AURPPO_LAUNCH_CHECK("k_mlp3_prep");
    return AURPPO_OK;
}

int launch_mlp_step3(const MlpArgs& a, int grid, hipStream_t s) {
    static bool attr_set[kMaxDevices] = {false};
    const int dslot = aurppo_device_slot();
    if (!attr_set[dslot]) {
        AURPPO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_step3), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)mlp_step3_lds_bytes()));
        attr_set[dslot] = true;
    }
    hipLaunchKernelGGL(k_mlp_step3, dim3(grid), dim3(kThreads3), mlp_step3_lds_bytes(), s, a);
    AURPPO_LAUNCH_CHECK("k_mlp_step3");
    return AURPPO_OK;
}

}  // namespace aurppo_mlp
