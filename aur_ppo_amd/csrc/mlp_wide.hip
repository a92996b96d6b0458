// K7w / K8w: the fused PPO minibatch step (as K7, mlp.hip / mlp2.hip) and the rollout step (as K8) for every MLP
// actor-critic the reference's CLI can ask for that is NOT the default 64-64 one: hidden_dim <= 128, num_layers 1..3,
// state_dim <= 128 (src/nets/nets.py:19-53 builds `num_layers` Tanh layers of `hidden_dim`; flags src/run_ppo.py:33,37).
// Same inputs, same outputs (per-workgroup gradient slabs + loss partials, folded by k_mlp_reduce; the optimizer
// step is K6b's), same fp32 MFMA arithmetic -- what differs is where things live, because at 128 x 3 neither the
// weights (2 x 192 KB) nor the weight-gradient accumulators (2 x 200 KB) of BOTH nets fit one CU:
//   * a workgroup (4 waves, one per SIMD) works on ONE net: actor and critic share nothing but the input rows and the
//     advantage statistics (the loss is a sum of an actor-only and a critic-only term), so even and odd workgroups
//     take the two nets of the same row tiles and write disjoint halves of the same slab;
//   * wave w owns the w-th 32-column block of every layer: its weight-gradient blocks (out-block 0..3, in-block w) of
//     every layer stay in registers over the whole launch (13 accumulators of 16 registers at 128 x 3: the 512-register
//     budget of one wave per SIMD is what makes that possible);
//   * the weights are not staged in LDS: a preparation kernel lays every hidden layer out once per launch in MFMA
//     B-operand order (forward: W^T blocks, backward: W blocks; zero-padded to 32-multiples), and a wave streams each
//     32x32 block with four 16-byte loads per lane from L2, one block ahead of the MFMA chain that consumes it;
//   * activations live in LDS ([32 rows][129]); dZ_l overwrites H_l in place (a wave only ever reads its own column
//     block of H_l once dZ_l is being formed), so a layer costs one 16.5 KB buffer.
// When nothing is wider than 64 (two column blocks) a workgroup carries BOTH nets instead -- waves 0,1 the actor's
// column blocks, waves 2,3 the critic's -- in a geometry (row stride 65, 256 registers) that fits a CU twice, so two
// independent workgroups overlap each other's epilogues and barriers the way k_mlp_step2's two tile sets do.
// Tiles are handed out by a counter.  Measured against the per-op path it replaces in DESIGN section 4.3c.
#include <stdlib.h>

#pragma clang fp contract(fast)
#include "bf16x3.h"
#include "mlp_common.h"

using namespace aurppo_mlp;

namespace {

constexpr int HPW = 128;            // widest (padded) layer
constexpr int LDW = HPW + 1;        // LDS row stride of an activation matrix (odd: conflict-free row- and column-wise)
constexpr int MAXL = 3;             // hidden layers
constexpr int kOpBlk = 64 * 16;     // floats of one 32x32 block in operand order: [lane][16 k-steps]
constexpr int kOpLayer = 16 * kOpBlk;                 // 4 x 4 blocks
constexpr int kOpFloats = 2 * MAXL * 2 * kOpLayer;    // [net][layer][fwd | bwd]
constexpr int kMaxSlabs = 2 * kMaxGrid;               // k_mlp_reduce folds up to 512

struct WideLayout {   // float offsets into the flat parameter / gradient bucket; layer NL is the head
    int w[2][MAXL + 1], b[2][MAXL + 1];
    int logstd, n_params;
};

struct WideArgs {
    const float* obs;      // (B, D)
    const float* actions;  // (B, AW) or nullptr (packed records)
    const float4* rec;     // (B, 4) | (B, 16) packed
    int rec_stride;
    const int32_t* idx;    // (M,)
    const float* params;
    const float* wop;      // operand-order copies (k_mlpw_prep)
    const unsigned short* wop3;   // k_mlpw3_step: bf16 planes of the hidden layers in operand order (k_mlpw3_prep)
    float* slabs;          // (pairs, n_params)
    double* loss_part;     // (pairs, 8)
    const double* stats;   // (n_stat_blocks, 2)
    unsigned* tile_counter;   // [2]: next tile to hand out, per net (one-net workgroups) or [0] alone (both-net workgroups)
    int n_stat_blocks;
    int D, A, Hd, continuous;
    WideLayout L;
    PpoHyper h;
    // rollout step
    const float* noise;
    float* out_actions;
    float* out_logp;
    float* out_value;
    int N, net_base, net_count;
    int static_tiles;      // diagnostic (AURPPO_STATIC_TILES): workgroup p of P takes tiles p, p + P, ... -- a fixed summation order
};

// stats[b] = partial (sum, sum of squares) of the minibatch's advantages; wop = every hidden layer of both nets in
// operand order.  Forward copy, block (ob, kb): lane l, step m holds W[ob*32 + (l & 31)][kb*32 + 2m + (l >> 5)]
// (B[k][j] = W[j][k]); backward copy, block (jb, kb): W[kb*32 + 2m + (l >> 5)][jb*32 + (l & 31)] (B[k][j] = W[k][j]).
__global__ __launch_bounds__(256) void k_mlpw_prep(const float* __restrict__ params, WideLayout L, int NL, int D, int Hd,
                                                   float* __restrict__ wop, const float4* __restrict__ rec, int rec_stride,
                                                   const int32_t* __restrict__ idx, int M, double (*__restrict__ stats)[2],
                                                   int n_stat_blocks, unsigned* __restrict__ tile_counter) {
    __shared__ double sc[2][4];
    if (tile_counter && blockIdx.x == 0 && threadIdx.x < 2) tile_counter[threadIdx.x] = 0u;
    if ((int)blockIdx.x < n_stat_blocks) {
        double s = 0.0, q = 0.0;
        adv_partial_sums(rec, rec_stride, idx, M, blockIdx.x * 256 + threadIdx.x, n_stat_blocks * 256, s, q);
        const double bs = block_sum<4>(s, sc[0]);
        const double bq = block_sum<4>(q, sc[1]);
        if (threadIdx.x == 0) {
            stats[blockIdx.x][0] = bs;
            stats[blockIdx.x][1] = bq;
        }
        return;
    }
    const int b = blockIdx.x - n_stat_blocks, nb = gridDim.x - n_stat_blocks;
    for (int e = b * 256 + threadIdx.x; e < kOpFloats; e += nb * 256) {
        const int m = e & 15, lane = (e >> 4) & 63, blk = (e >> 10) & 15, dir = (e >> 14) & 1, nl = e >> 15;
        const int n = nl / MAXL, l = nl - n * MAXL;
        if (l >= NL || (dir == 1 && l == 0)) continue;
        const int in_dim = l == 0 ? D : Hd;
        const float* W = params + L.w[n][l];
        const int hi = blk >> 2, kb = blk & 3;
        int row, col;
        if (dir == 0) {
            row = hi * 32 + (lane & 31);
            col = kb * 32 + 2 * m + (lane >> 5);
        } else {
            row = kb * 32 + 2 * m + (lane >> 5);
            col = hi * 32 + (lane & 31);
        }
        wop[e] = (row < Hd && col < in_dim) ? W[row * in_dim + col] : 0.0f;
    }
}

__device__ __forceinline__ const float* op_block(const float* wop, int net, int l, int dir, int hi, int kb, int lane) {
    return wop + (size_t)((net * MAXL + l) * 2 + dir) * kOpLayer + (hi * 4 + kb) * kOpBlk + lane * 16;
}

__device__ __forceinline__ void load_b(float (&b)[16], const float* p) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float4 v = q[u];
        b[4 * u + 0] = v.x; b[4 * u + 1] = v.y; b[4 * u + 2] = v.z; b[4 * u + 3] = v.w;
    }
}

// acc += A(32 x 32) * B(32 x 32): A from LDS through a_at(i, k), B already in registers (operand order)
template <class FA>
__device__ __forceinline__ void mma_breg(f32x16& acc, FA a_at, const float (&b)[16], int lane) {
    const int ij = lane & 31, kk = lane >> 5;
    float av[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) av[m] = a_at(ij, 2 * m + kk);
#pragma unroll
    for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], b[m], acc, 0, 0, 0);
}

// One output block of a layer: acc = sum over nkb in-blocks of In[:, kb] * Wop[block (hi, kb)], the weight blocks
// streamed one ahead.  In: [R][LDW] in LDS.
// LOAD_FIRST = false: `early` already holds (or is loading) the layer's first block -- the caller issued that load a phase
// early, behind work that does not need it, so its L2 latency is not on the chain (EARLY below; not with three 128-wide
// layers, whose accumulators leave no registers to hold a block across a phase).
template <bool LOAD_FIRST, int LD_>
__device__ __forceinline__ f32x16 stream_layer(const float* In, const float* wop, int net, int l, int dir, int hi, int nkb,
                                               int lane, float (&early)[16]) {
    f32x16 acc = zero16();
    float bq[2][16];
    if (LOAD_FIRST) {
        load_b(bq[0], op_block(wop, net, l, dir, hi, 0, lane));
    } else {
#pragma unroll
        for (int m = 0; m < 16; ++m) bq[0][m] = early[m];
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        if (kb < nkb) {
            if (kb + 1 < nkb) load_b(bq[(kb + 1) & 1], op_block(wop, net, l, dir, hi, kb + 1, lane));
            mma_breg(acc, [&](int i, int k) { return In[i * LD_ + kb * 32 + k]; }, bq[kb & 1], lane);
        }
    }
    return acc;
}

struct WideLds {
    float* sX;            // [R][LDW]
    float* sH[MAXL];      // [R][LDW] each; dZ_l overwrites H_l in the backward pass
    float* sW3;           // [AP][LDW] head weights of this net, rows >= out_dim zero
    float* sOut;          // [R][LDO] head outputs, then their gradients
    float* sAct;          // [R][LDO]
    float* sDls;          // [R][LDO] per-sample d logstd terms (actor)
    float* sB;            // [MAXL][HPW] hidden biases, zero beyond Hd
    float* sB3;           // [AP]
    float* sLs;           // [AP]
    float* sIvar;         // [AP]  1 / sigma^2 (step) | sigma (act)
    float4* sRec;         // [R]
    int* sSrc;            // [R]
    int* sIdx;            // [2][R]
};

// [sX][per-net block x nslots][sAct sDls sLs sIvar sRec sSrc sIdx]: one net per workgroup, or both (nslots = 2) when the
// layers are at most 64 wide -- see k_mlpw_step.  NARROW (that case): row stride 65 and 64 bias slots per layer, so that two
// such workgroups fit one CU's 160 KB.
template <bool NARROW> struct Geo {
    static constexpr int LD = NARROW ? 65 : LDW;      // row stride of an activation matrix (odd either way)
    static constexpr int HP = NARROW ? 64 : HPW;      // bias slots per layer
};
template <bool NARROW, int NL>
constexpr int net_floats() { return NL * R * Geo<NARROW>::LD + AP * Geo<NARROW>::LD + R * LDO + MAXL * Geo<NARROW>::HP + AP; }
constexpr int kTailFloats = 2 * R * LDO + 2 * AP;
template <bool NARROW, int NL>
__device__ __forceinline__ WideLds carve(float* lds, int slot, int nslots) {
    constexpr int LD_ = Geo<NARROW>::LD;
    WideLds s;
    s.sX = lds;
    float* nb = lds + R * LD_ + slot * net_floats<NARROW, NL>();
    for (int l = 0; l < MAXL; ++l) s.sH[l] = nb + (l < NL ? l : 0) * R * LD_;
    s.sW3 = nb + NL * R * LD_;
    s.sOut = s.sW3 + AP * LD_;
    s.sB = s.sOut + R * LDO;
    s.sB3 = s.sB + MAXL * Geo<NARROW>::HP;
    float* tail = lds + R * LD_ + nslots * net_floats<NARROW, NL>();
    s.sAct = tail;
    s.sDls = s.sAct + R * LDO;
    s.sLs = s.sDls + R * LDO;
    s.sIvar = s.sLs + AP;
    s.sRec = reinterpret_cast<float4*>(s.sIvar + AP);   // every term above is a multiple of 4 floats
    s.sSrc = reinterpret_cast<int*>(s.sRec + R);
    s.sIdx = s.sSrc + R;
    return s;
}
static_assert((R * LDW) % 4 == 0 && (R * 65) % 4 == 0 && net_floats<false, 1>() % 4 == 0 && net_floats<false, 2>() % 4 == 0 &&
              net_floats<false, 3>() % 4 == 0 && net_floats<true, 1>() % 4 == 0 && net_floats<true, 2>() % 4 == 0 &&
              net_floats<true, 3>() % 4 == 0 && kTailFloats % 4 == 0, "sRec must be 16-B aligned");
template <bool NARROW, int NL>
constexpr size_t wide_lds_bytes(int nslots) {
    return sizeof(float) * (size_t)(R * Geo<NARROW>::LD + nslots * net_floats<NARROW, NL>() + kTailFloats + 4 * R + R + 2 * R);
}
static_assert(2 * (wide_lds_bytes<true, 3>(2) + 1024) <= 160 * 1024, "two both-net workgroups of a <= 64-wide policy must fit one CU's LDS");

// weights that stay in LDS: the head, every bias, log-std
template <int NL, bool NARROW>
__device__ __forceinline__ void stage_small(const WideArgs& a, const WideLds& s, int net, bool act_mode) {
    constexpr int LD_ = Geo<NARROW>::LD, HP_ = Geo<NARROW>::HP;
    const int tid = threadIdx.x, Hd = a.Hd, out_dim = net == 0 ? a.A : 1;
    for (int e = tid; e < R * LD_; e += kThreads) s.sX[e] = 0.0f;
    for (int e = tid; e < AP * LD_; e += kThreads) {
        const int o = e / LD_, i = e - o * LD_;
        s.sW3[e] = (o < out_dim && i < Hd) ? a.params[a.L.w[net][NL] + o * Hd + i] : 0.0f;
    }
    for (int e = tid; e < NL * HP_; e += kThreads) {
        const int l = e / HP_, c = e - l * HP_;
        s.sB[e] = c < Hd ? a.params[a.L.b[net][l] + c] : 0.0f;
    }
    if (tid < AP) {
        s.sB3[tid] = tid < out_dim ? a.params[a.L.b[net][NL] + tid] : 0.0f;
        const float ls = (a.continuous && tid < a.A) ? a.params[a.L.logstd + tid] : 0.0f;
        const float sd = expf(ls);
        s.sLs[tid] = ls;
        s.sIvar[tid] = act_mode ? sd : 1.0f / (sd * sd);
    }
    for (int e = tid; e < R * LDO; e += kThreads) s.sDls[e] = 0.0f;
}

// forward pass of one net over the tile in sX: H_1 .. H_NL, then the head into sOut (+ bias)
template <int NL, bool EARLY, bool NARROW>
__device__ __forceinline__ void forward_tile(const WideArgs& a, const WideLds& s, int net, int cb, int HB, int DB,
                                             float (&early)[16]) {
    constexpr int LD_ = Geo<NARROW>::LD, HP_ = Geo<NARROW>::HP;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        if (cb < HB) {
            const float* In = l == 0 ? s.sX : s.sH[l - 1];
            const f32x16 acc = stream_layer<!EARLY, LD_>(In, a.wop, net, l, 0, cb, l == 0 ? DB : HB, lane, early);
            if (EARLY && l + 1 < NL) load_b(early, op_block(a.wop, net, l + 1, 0, cb, 0, lane));   // behind the epilogue and the barrier
            const int col = cb * 32 + (lane & 31);
            const float bias = s.sB[l * HP_ + col];
            float* Hl = s.sH[l];
#pragma unroll
            for (int e = 0; e < 16; ++e) Hl[acc_row(e, lane) * LD_ + col] = tanh_fast(acc[e] + bias);
        }
        __syncthreads();
    }
    if (cb < 2) {   // head: 16 rows per wave as one 16x16 tile, K = HB blocks of 32
        const float* Hin = s.sH[NL - 1] + cb * 16 * LD_;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
            if (kb < HB)
                acc += mma16<32>([&](int i, int k) { return Hin[i * LD_ + kb * 32 + k]; },
                                 [&](int k, int j) { return s.sW3[j * LD_ + kb * 32 + k]; });
        const int col = lane & 15;
        const float bias = s.sB3[col];
#pragma unroll
        for (int e = 0; e < 4; ++e) s.sOut[(cb * 16 + 4 * (lane >> 4) + e) * LDO + col] = acc[e] + bias;
    }
    __syncthreads();
}

// DUAL = false: a workgroup is one net (blockIdx & 1), wave w owns column block w (layers up to 128 wide).
// DUAL = true (layers and state at most 64 wide, i.e. two column blocks): a workgroup is BOTH nets of its tiles -- waves
// 0,1 the actor's two column blocks, waves 2,3 the critic's -- so no wave idles and a tile's rows are fetched once.
template <int NL, bool DUAL>
__global__ __launch_bounds__(256, DUAL ? 2 : 1) void k_mlpw_step(const WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double s_red[2][kThreads / kWave];
    __shared__ float s_mean, s_std;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int net = DUAL ? (w >> 1) : (int)(blockIdx.x & 1), cb = DUAL ? (w & 1) : w;
    const int pair = DUAL ? (int)blockIdx.x : (int)(blockIdx.x >> 1);
    constexpr int LD_ = Geo<DUAL>::LD;
    constexpr int XS = DUAL ? 8 : 16;    // staging slots per thread: columns (tid & 7) + 8u cover 64 or 128 state floats
    const WideLds s = carve<DUAL, NL>(lds, DUAL ? net : 0, DUAL ? 2 : 1);
    const int hw = DUAL ? 1 : 3;          // the actor wave that also forms the head-side column sums
    const int lrow = DUAL ? (tid & 127) : tid;   // loss lanes: the first 32 lanes of each net's first wave
    const int D = a.D, A = a.A, Hd = a.Hd;
    const int HB = (Hd + 31) >> 5, DB = (D + 31) >> 5;
    const int AW = a.continuous ? A : 1;
    const int out_dim = net == 0 ? A : 1;
    const bool own_out = NL < 3 && DB < HB;   // (not with three layers: no registers for a second code path)
    constexpr bool EARLY = DUAL ? NL == 1 : NL < 3;   // holding a weight block across a phase costs 16 registers
    constexpr int CHW = (NL == 3 || DUAL) ? 4 : 8;   // operand read-ahead of the LDS-fed chains: three layers of accumulators leave fewer registers

    if (DUAL) {
        stage_small<NL, DUAL>(a, carve<DUAL, NL>(lds, 0, 2), 0, false);
        stage_small<NL, DUAL>(a, carve<DUAL, NL>(lds, 1, 2), 1, false);
    } else {
        stage_small<NL, DUAL>(a, s, net, false);
    }
    {
        double sm = 0.0, q = 0.0;
        for (int b = tid; b < a.n_stat_blocks; b += kThreads) {
            sm += a.stats[2 * b];
            q += a.stats[2 * b + 1];
        }
        const double ts = block_sum<kThreads / kWave>(sm, s_red[0]);
        const double tq = block_sum<kThreads / kWave>(q, s_red[1]);
        if (tid == 0) {
            const double m = ts / (double)a.h.M;
            double var = (tq - ts * m) / (double)(a.h.M - 1);
            if (var < 0.0) var = 0.0;
            s_mean = (float)m;
            s_std = (float)sqrt(var);
        }
    }
    __syncthreads();
    const float mean = s_mean, denom = s_std + 1e-8f;
    const float invM = 1.0f / (float)a.h.M;
    const float g_ent = -a.h.ent_coef * invM;

    // persistent accumulators: dW_l blocks (out-block ob, in-block w), head block (rows < AP, in-block w), bias columns
    constexpr int OBN = DUAL ? 2 : 4;   // out-blocks a layer can have
    f32x16 gW[NL][OBN];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int ob = 0; ob < OBN; ++ob) gW[l][ob] = zero16();
    f32x16 gW3 = zero16();
    float gb[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) gb[l] = 0.0f;
    double l_a = 0, l_b = 0, l_c = 0, l_d = 0, l_e = 0;   // actor: pg, ent, okl, kl, cf; critic: vl in l_a
    float g_b3c = 0.0f, g_head = 0.0f;

    const int n_tiles = (a.h.M + R - 1) / R;
    // staging slots: 8 threads per row, columns (tid & 7) + 8u
    const int x_r = tid >> 3, x_c0 = tid & 7;
    float xr[XS], ar[2];
    unsigned xok = 0u;               // bit u: xr[u] is real; bit 16 + u: ar[u] (the zeroing waits until the tile lands)
    float4 p_rec = make_float4(0.f, 0.f, 0.f, 0.f);
    int p_src = -1, n_raw = 0;
    bool n_ok = false;
    auto load_idx = [&](int tile) -> int {
        const int m = tile * R + tid;
        return (tile < n_tiles && m < a.h.M) ? a.idx[m] : -1;
    };
    const float* const act_base = a.actions ? a.actions : reinterpret_cast<const float*>(a.rec) + 4;
    const int act_stride = a.actions ? AW : 16;
    // Every load below is unconditional (a piece that is not real reads element 0) and nothing is computed from its result here:
    // a conditional load is a write to its destination on the other path, and the wait the compiler puts in front of that write --
    // or in front of an `ok ? v : 0` select -- is a wait for every load above it (k_mlpw3_step's lesson).
    auto prefetch = [&](const int* sidx) {
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int src = sidx[tq >> 3];
        xok = 0u;
#pragma unroll
        for (int u = 0; u < XS; ++u) {
            const int c = (tq & 7) + 8 * u;
            const bool ok = src >= 0 && c < D;
            const unsigned row = ok ? (unsigned)src : 0u, col = ok ? (unsigned)c : 0u;
            xr[u] = a.obs[(size_t)row * (unsigned)D + col];
            xok |= ok ? (1u << u) : 0u;
        }
        if (DUAL || net == 0) {
            int sa[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) sa[u] = sidx[(tq + u * kThreads) >> 4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = tq & 15;
                const bool ok = sa[u] >= 0 && c < AW;
                const unsigned row = ok ? (unsigned)sa[u] : 0u, col = ok ? (unsigned)c : 0u;
                ar[u] = act_base[(size_t)row * (unsigned)act_stride + col];
                xok |= ok ? (1u << (16 + u)) : 0u;
            }
        }
        p_src = sidx[tq & (R - 1)];
        p_rec = a.rec[(size_t)(p_src >= 0 ? p_src : 0) * a.rec_stride];
    };
    auto prefetch_idx = [&](int tile) {
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int m = tile * R + (tq & (R - 1));
        n_ok = tile < n_tiles && m < a.h.M;
        n_raw = a.idx[n_ok ? m : 0];
    };
    // Tiles are handed out by a counter, not strided statically: a workgroup that starts late (its CU held by one of the
    // side stream's shuffle kernels) then simply takes fewer -- with static striding every launch that overlapped them ran
    // at the pace of its latest workgroup (in-situ 247 us min, 443 max).  s_tile[(it + k) & 3], k = 0..3: the tiles this
    // workgroup works on next; rows are fetched one tile ahead, indices two to three.
    __shared__ int s_tile[4];
    unsigned* const ctr = a.tile_counter + (DUAL ? 0 : net);
    const bool stat = a.static_tiles != 0;
    const int n_wg = DUAL ? (int)gridDim.x : (int)(gridDim.x >> 1);   // workgroups that share this net's tiles
    if (tid == 0) {       // four consecutive tiles to start with; later ones one at a time (ids only ever grow)
        if (stat) {
            s_tile[0] = pair; s_tile[1] = pair + n_wg; s_tile[2] = pair + 2 * n_wg; s_tile[3] = pair + 3 * n_wg;
        } else {
            const int t0 = (int)atomicAdd(ctr, 4u);
            s_tile[0] = t0; s_tile[1] = t0 + 1; s_tile[2] = t0 + 2; s_tile[3] = t0 + 3;
        }
    }
    __syncthreads();
    if (tid < R) {
        s.sIdx[tid] = load_idx(s_tile[0]);
        s.sIdx[R + tid] = load_idx(s_tile[1]);
    }
    __syncthreads();
    prefetch(s.sIdx);
    prefetch_idx(s_tile[2]);
    for (int it = 0;; ++it) {
        const int tile = s_tile[it & 3];
        if (tile >= n_tiles) break;
        const int tile3 = s_tile[(it + 3) & 3];
        // ---- land the prefetched tile, start fetching the next one
        float early[16];
        if (EARLY && cb < HB) load_b(early, op_block(a.wop, net, 0, 0, cb, 0, lane));   // layer 1's first weight block, behind the landing
#pragma unroll
        for (int u = 0; u < XS; ++u) {
            const int c = x_c0 + 8 * u;
            if (c < D) s.sX[x_r * LD_ + c] = ((xok >> u) & 1u) ? xr[u] : 0.0f;
        }
        if (DUAL || net == 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = tid + u * kThreads;
                s.sAct[(e >> 4) * LDO + (e & 15)] = ((xok >> (16 + u)) & 1u) ? ar[u] : 0.0f;
            }
        }
        if (tid < R) {
            s.sSrc[tid] = p_src;
            s.sRec[tid] = p_rec;
            s.sIdx[(it & 1) * R + tid] = n_ok ? n_raw : -1;
        }
        __syncthreads();
        // the next tile's rows are fetched behind this tile's math -- except with three layers, whose accumulators leave
        // no registers to hold them that long: there they are fetched at the end of the tile (latency exposed, ~5 %)
        if (NL < 3) prefetch(s.sIdx + ((it + 1) & 1) * R);
        prefetch_idx(tile3);

        forward_tile<NL, EARLY, DUAL>(a, s, net, cb, HB, DB, early);
        // the tile after the three already known: asked for here, where this wave has no loads queued behind the atomic
        // (returns are in order), stored in s_tile at the end of the tile
        int tile4 = 0;
        if (tid == 0) tile4 = stat ? tile + 4 * n_wg : (int)atomicAdd(ctr, 1u);

        // ---- loss lanes (one per row): this net's half of the PPO terms; head outputs become their gradients
        if (lrow < R) {
            float* out = s.sOut + lrow * LDO;
            if (s.sSrc[lrow] >= 0) {
                const float4 rc = s.sRec[lrow];
                if (net == 1) {
                    const PpoSample t = ppo_sample(rc.x, rc.x, rc.y, out[0], rc.w, rc.z, mean, denom, invM, a.h);
                    l_a += t.vl;
                    out[0] = t.g_v;
                    g_b3c += t.g_v;
                } else if (a.continuous) {
                    const float* act = s.sAct + lrow * LDO;
                    float logp = 0.0f, ent = 0.0f;
                    for (int k = 0; k < A; ++k) {
                        const float ls = s.sLs[k];
                        const float zk = act[k] - out[k];
                        logp += (-(zk * zk) * (0.5f * s.sIvar[k]) - ls) - 0.9189385332046727f;
                        ent += (0.5f + 0.9189385332046727f) + ls;
                    }
                    const PpoSample t = ppo_sample(logp, rc.x, rc.y, rc.w, rc.w, rc.z, mean, denom, invM, a.h);
                    l_a += t.pg; l_b += ent; l_c += t.okl; l_d += t.kl; l_e += t.cf;
                    for (int k = 0; k < A; ++k) {
                        const float zk = act[k] - out[k];
                        out[k] = t.g_logp * (zk * s.sIvar[k]);
                        s.sDls[lrow * LDO + k] = t.g_logp * (zk * zk * s.sIvar[k] - 1.0f) + g_ent;
                    }
                } else {
                    const float* act = s.sAct + lrow * LDO;
                    float mx = out[0];
                    for (int k = 1; k < A; ++k) mx = fmaxf(mx, out[k]);
                    float se = 0.0f;
                    for (int k = 0; k < A; ++k) se += expf(out[k] - mx);
                    const float lse = mx + logf(se);
                    const int ai = (int)act[0];
                    float logp = 0.0f, ent = 0.0f;
                    for (int k = 0; k < A; ++k) {
                        const float lpk = out[k] - lse;
                        ent -= expf(lpk) * lpk;
                        if (k == ai) logp = lpk;
                    }
                    const PpoSample t = ppo_sample(logp, rc.x, rc.y, rc.w, rc.w, rc.z, mean, denom, invM, a.h);
                    l_a += t.pg; l_b += ent; l_c += t.okl; l_d += t.kl; l_e += t.cf;
                    for (int k = 0; k < A; ++k) {
                        const float lpk = out[k] - lse;
                        const float pk = expf(lpk);
                        out[k] = t.g_logp * ((k == ai ? 1.0f : 0.0f) - pk) + g_ent * (-pk * (lpk + ent));
                    }
                }
            } else {
                for (int k = 0; k < AP; ++k) out[k] = s.sDls[lrow * LDO + k] = 0.0f;
            }
        }
        __syncthreads();

        // ---- head backward: column sums (d b3, d logstd), dH_NL -> dZ_NL (in place), dW3
        if (net == 0 && w == hw && lane < 2 * AP) {
            const float* src = lane < AP ? s.sOut + lane : s.sDls + (lane - AP);
            float cs = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) cs += src[r * LDO];
            g_head += cs;
        }
        if (cb < HB) {
            float* HL = s.sH[NL - 1];
            f32x16 acc = zero16();
            mma32<AP>(acc, [&](int i, int k) { return s.sOut[i * LDO + k]; },
                      [&](int k, int j) { return s.sW3[k * LD_ + cb * 32 + j]; });
            mma32<R, CHW>(gW3, [&](int i, int k) { return i < AP ? s.sOut[k * LDO + i] : 0.0f; },
                          [&](int k, int j) { return HL[k * LD_ + cb * 32 + j]; }, lane);
            const int col = cb * 32 + (lane & 31);
            float colsum = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float* hp = HL + acc_row(e, lane) * LD_ + col;
                const float h = *hp;
                const float dz = acc[e] * (1.0f - h * h);
                colsum += dz;
                *hp = dz;
            }
            colsum += __shfl_xor(colsum, 32, kWave);
            gb[NL - 1] += colsum;
        }
        __syncthreads();
        // ---- hidden layers, top down: dW_l, dH_{l-1} -> dZ_{l-1} (in place)
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            if (cb < HB) {
                const float* dZ = s.sH[l];
                float* Hp = s.sH[l - 1];
                if (EARLY) load_b(early, op_block(a.wop, net, l, 1, cb, 0, lane));   // dH's first weight block, behind the dW chains
#pragma unroll
                for (int ob = 0; ob < OBN; ++ob)
                    if (ob < HB)
                        mma32<R, CHW>(gW[l][ob], [&](int i, int k) { return dZ[k * LD_ + ob * 32 + i]; },
                                      [&](int k, int j) { return Hp[k * LD_ + cb * 32 + j]; }, lane);
                const f32x16 acc = stream_layer<!EARLY, LD_>(dZ, a.wop, net, l, 1, cb, HB, lane, early);
                const int col = cb * 32 + (lane & 31);
                float colsum = 0.0f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float* hp = Hp + acc_row(e, lane) * LD_ + col;
                    const float h = *hp;
                    const float dz = acc[e] * (1.0f - h * h);
                    colsum += dz;
                    *hp = dz;
                }
                colsum += __shfl_xor(colsum, 32, kWave);
                gb[l - 1] += colsum;
            }
            __syncthreads();
        }
        // ---- dW_1: a wave owns the state's in-block cb and walks the out-blocks, as for the other layers -- unless the
        // state is narrower than the layer (DB < HB), where owning OUT-block cb and walking the in-blocks keeps every
        // wave busy (gW[0][k] is then block (cb, k) instead of (k, cb))
        if (own_out) {
            if (cb < HB) {
                const float* dZ = s.sH[0];
#pragma unroll
                for (int ib = 0; ib < OBN; ++ib)
                    if (ib < DB)
                        mma32<R, CHW>(gW[0][ib], [&](int i, int k) { return dZ[k * LD_ + cb * 32 + i]; },
                                      [&](int k, int j) { return s.sX[k * LD_ + ib * 32 + j]; }, lane);
            }
        } else if (cb < DB) {
            const float* dZ = s.sH[0];
#pragma unroll
            for (int ob = 0; ob < OBN; ++ob)
                if (ob < HB)
                    mma32<R, CHW>(gW[0][ob], [&](int i, int k) { return dZ[k * LD_ + ob * 32 + i]; },
                                  [&](int k, int j) { return s.sX[k * LD_ + cb * 32 + j]; }, lane);
        }
        if (NL >= 3) prefetch(s.sIdx + ((it + 1) & 1) * R);
        if (tid == 0) s_tile[it & 3] = tile4;               // everyone read this slot at the top of the tile
        __syncthreads();
    }

    // ---- this workgroup's half of the pair's slab
    float* slab = a.slabs + (size_t)pair * a.L.n_params;
    {
        const int col = cb * 32 + (lane & 31);
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int in_dim = l == 0 ? D : Hd;
#pragma unroll
            for (int ob = 0; ob < OBN; ++ob) {
                if (l == 0 && own_out) {       // gW[0][ob] = block (out cb, in ob)
                    const int c0 = ob * 32 + (lane & 31);
                    if (ob < DB && cb < HB && c0 < D) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int o = cb * 32 + acc_row(e, lane);
                            if (o < Hd) slab[a.L.w[net][0] + o * D + c0] = gW[0][ob][e];
                        }
                    }
                } else if (ob < HB && col < in_dim) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int o = ob * 32 + acc_row(e, lane);
                        if (o < Hd) slab[a.L.w[net][l] + o * in_dim + col] = gW[l][ob][e];
                    }
                }
            }
            if (lane < 32 && col < Hd) slab[a.L.b[net][l] + col] = gb[l];
        }
        if (col < Hd) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int o = acc_row(e, lane);
                if (o < out_dim) slab[a.L.w[net][NL] + o * Hd + col] = gW3[e];
            }
        }
    }
    if (net == 0 && w == hw) {
        if (lane < A) slab[a.L.b[0][NL] + lane] = g_head;
        if (a.continuous && lane >= AP && lane - AP < A) slab[a.L.logstd + lane - AP] = g_head;
    }
    if (cb == 0) {
        float c = lane < R ? g_b3c : 0.0f;
        double v5[5] = {l_a, l_b, l_c, l_d, l_e};
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) c += __shfl_down(c, off, kWave);
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            double x = lane < R ? v5[q] : 0.0;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) x += __shfl_down(x, off, kWave);
            v5[q] = x;
        }
        if (lane == 0) {
            double* lp = a.loss_part + (size_t)pair * 8;   // {pg, vl, ent, okl, kl, cf, mean, std}
            if (net == 0) {
                lp[0] = v5[0]; lp[2] = v5[1]; lp[3] = v5[2]; lp[4] = v5[3]; lp[5] = v5[4];
            } else {
                slab[a.L.b[1][NL]] = c;
                lp[1] = v5[0];
                lp[6] = (double)mean;
                lp[7] = (double)s_std;
            }
        }
    }
}

// =====================================================================================================================
// k_mlpw3_step -- K7w on the bf16 matrix pipe (round 4): k_mlpw_step<NL, false>'s geometry (one net per workgroup, four
// waves, wave w = the w-th 32-column block of every layer, weight gradients persistent in registers) with k_mlp_step3's
// arithmetic and operand handling (bf16x3.h: every fp32 operand as three bf16 planes, six v_mfma_f32_32x32x16_bf16 per
// K = 16, fp32 accumulate -- fp32-equivalent, tools/bf16x3_check.hip):
//   * activations are LDS images of bf16 planes -- X as two 64-column X images side by side, H_l (later dZ_l, in place) as
//     128-feature F images -- written straight from the accumulators by the tanh / dZ epilogues; a fragment is one
//     ds_read_b128 or two ds_read_b64_tr_b16 where the fp32 kernel issued a ds_read_b32 per matrix instruction (rocprofv3:
//     k_mlpw_step<3,false> kept the matrix pipe busy 48 % of its 862 us -- on fp32 instructions that cost 2.7x the cycles);
//   * the weights stream from L2 as bf16 planes in B-operand order (k_mlpw3_prep; refreshed in place by the optimizer launch
//     of a chained minibatch), a layer's whole slice (<= 8 k-steps x 3 planes = 96 registers) one phase ahead of its use --
//     one wave per SIMD has the 512-register budget for it next to the 12 x 16 accumulator registers;
//   * head, loss lanes, column sums and the tile queue are k_mlpw_step's; the head gradients also go out as a bf16-plane
//     image ([a 16][s 32]) for the two products that consume them.
// Same arguments, slabs and loss partials as k_mlpw_step: k_mlp_reduce and the optimizer launch do not know the difference.
namespace w3 {
using namespace bf3;
constexpr int kFPlaneW = HPW * kFRow;            // F image of 128 features: bytes per plane
constexpr int kXHalf = 3 * kXPlane;              // one 64-column X image (three planes)
constexpr int kW3RowW = 2 * HPW, kW3PlaneW = AP * kW3RowW;       // W3 image [a 16][i 128]
constexpr int kDoRowW = 64, kDoPlaneW = AP * kDoRowW;            // dOut image [a 16][s 32]
constexpr int kWopKs = 8;                        // k-steps of a weight slice (128 / 16)
constexpr int kWopSlice = kWopKs * 3 * 512;      // bf16 elements of one (slot, column block) slice
constexpr int kWopSlot = 4 * kWopSlice;          // one (net, layer, direction)
constexpr int kWopElems = 2 * MAXL * 2 * kWopSlot;
__device__ __host__ __forceinline__ int wop_slot(int net, int l, int dir) { return (net * MAXL + l) * 2 + dir; }
// element (slot, column block cb, k-step ks, plane p, lane, j)
__device__ __host__ __forceinline__ int wop_index(int slot, int cb, int ks, int p, int lane, int j) {
    return slot * kWopSlot + cb * kWopSlice + ((ks * 3 + p) * 64 + lane) * 8 + j;
}

// byte offsets of the workgroup's LDS
constexpr int oX = 0;                                  // [2 halves][3 planes][4 KB]
constexpr int oH = oX + 2 * kXHalf;                    // [MAXL][3 planes][8 KB]
constexpr int oW3 = oH + MAXL * 3 * kFPlaneW;          // [3 planes][4 KB]
constexpr int oDo = oW3 + 3 * kW3PlaneW;               // [3 planes][1 KB]
constexpr int oOut = oDo + 3 * kDoPlaneW;              // float [R][LDO]
constexpr int oAct = oOut + 4 * R * LDO;               // float [R][LDO]
constexpr int oDls = oAct + 4 * R * LDO;               // float [R][LDO]
constexpr int oB = oDls + 4 * R * LDO;                 // float [MAXL][HPW]
constexpr int oB3 = oB + 4 * MAXL * HPW;               // float [AP]
constexpr int oLs = oB3 + 4 * AP;                      // float [AP]
constexpr int oIvar = oLs + 4 * AP;                    // float [AP]
constexpr int oRec = oIvar + 4 * AP;                   // float4 [R]
constexpr int oSrc = oRec + 16 * R;                    // int [R]
constexpr int oIdx = oSrc + 4 * R;                     // int [2][R]
constexpr int oStage = (oIdx + 4 * 2 * R + 15) / 16 * 16;      // float [R][HPW]: the next tile's rows as they arrive (LDS-DMA)
constexpr int oStageA = oStage + 4 * R * HPW;          // float [R][16]: its action rows
constexpr int kBytes = oStageA + 4 * R * 16;
static_assert(oH % 16 == 0 && oW3 % 16 == 0 && oDo % 16 == 0 && oOut % 16 == 0 && oRec % 16 == 0, "16-byte alignment of the images");
static_assert(kBytes <= 160 * 1024, "one workgroup per CU");

template <int PL>
__device__ __forceinline__ Frag3 f_rows_p(const char* img, int f0, int ks, int lane) {      // (addresses as in bf16x3.h: lane part + k-step constant)
    const int o = (foff(lane & 31, 8 * (lane >> 5)) ^ (ks << 5)) + f0 * kFRow;
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = lds_b128(img + p * PL + o);
    return f;
}
template <int PL>
__device__ __forceinline__ Frag3 f_cols_p(const char* img, int ks, int lane) {
    const TrLane t = tr_lane32(lane);
    const int o0 = foff(t.kq, t.m0) + 16 * ks * kFRow, o1 = foff(t.kq + 4, t.m0) + 16 * ks * kFRow;
    Frag3 r;
#pragma unroll
    for (int p = 0; p < 3; ++p) r.p[p] = join_tr(lds_tr(img + p * PL + o0), lds_tr(img + p * PL + o1));
    return r;
}
template <int PL>
__device__ __forceinline__ Frag3 f_cols16_p(const char* img, int s0, int ks, int lane) {
    const TrLane t = tr_lane16(lane);
    const int o0 = foff(t.kq, s0 + t.m0) + 32 * ks * kFRow, o1 = foff(t.kq + 4, s0 + t.m0) + 32 * ks * kFRow;
    Frag3 r;
#pragma unroll
    for (int p = 0; p < 3; ++p) r.p[p] = join_tr(lds_tr(img + p * PL + o0), lds_tr(img + p * PL + o1));
    return r;
}
template <int PL>
__device__ __forceinline__ Frag3 f_rows16_p(const char* img, int f0, int lane) {
    const int o = foff(f0 + (lane & 15), 8 * (lane >> 4));
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = lds_b128(img + p * PL + o);
    return f;
}
// LDS-DMA: 16 / 4 bytes per lane from global memory straight into LDS at dst + lane * 16 / 4 (wave-uniform dst), no registers
__device__ __forceinline__ void dma16(const void* src, void* dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_uniform, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const void* src, void* dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_uniform, 4, 0, 0);
}
// sum / maximum over the 16 lanes of a DPP row (every lane gets the result)
__device__ __forceinline__ float row16_sum(float v) {
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) v += __shfl_xor(v, m, 16);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 16));
    return v;
}
// tanh(acc + bias) of a 32x32 block into an F image, four values (one 8-byte store per plane) at a time
// (om[e] = 1 - tanh^2 of the same element stays in registers for the backward pass, as in k_mlp_step3: joined again from the image's
// planes it cost three unpacks and two adds per value and twelve LDS reads per block)
template <int PL>
__device__ __forceinline__ void tanh_store_p(char* img, int f0, const f32x16& acc, float bias, int lane, float (&om)[16]) {
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
        *reinterpret_cast<u32x2*>(img + 0 * PL + o) = u32x2{a0, b0};
        *reinterpret_cast<u32x2*>(img + 1 * PL + o) = u32x2{a1, b1};
        *reinterpret_cast<u32x2*>(img + 2 * PL + o) = u32x2{a2, b2};
    }
}
// dZ = dH * (1 - h^2), (1 - h^2) from the forward pass's registers, written over the block of h in the image; returns the lane's column sum
template <int PL>
__device__ __forceinline__ float dz_from_regs_p(char* img, int f0, const f32x16& dh, const float (&om)[16], int lane) {
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
        const int o = foff(f, 4 * h) ^ (gq << 4);       // = foff(f, 8 gq + 4 h)
        *reinterpret_cast<u32x2*>(img + 0 * PL + o) = u32x2{a0, b0};
        *reinterpret_cast<u32x2*>(img + 1 * PL + o) = u32x2{a1, b1};
        *reinterpret_cast<u32x2*>(img + 2 * PL + o) = u32x2{a2, b2};
    }
    return colsum;
}
}  // namespace w3

// wop3 = the bf16 planes of every hidden layer of both nets in operand order (w3::wop_index), written destination-first so
// that the padding (rows >= Hd, columns >= the layer's input width) is zero without a clearing pass.  The first n_stat_blocks
// workgroups form the advantage partial sums instead (as k_mlpw_prep).
__global__ __launch_bounds__(256) void k_mlpw3_prep(const float* __restrict__ params, WideLayout L, int NL, int D, int Hd,
                                                    unsigned short* __restrict__ wop3, const float4* __restrict__ rec, int rec_stride,
                                                    const int32_t* __restrict__ idx, int M, double (*__restrict__ stats)[2],
                                                    int n_stat_blocks, unsigned* __restrict__ tile_counter) {
    __shared__ double sc[2][4];
    if (tile_counter && blockIdx.x == 0 && threadIdx.x < 2) tile_counter[threadIdx.x] = 0u;
    if ((int)blockIdx.x < n_stat_blocks) {
        double s = 0.0, q = 0.0;
        adv_partial_sums(rec, rec_stride, idx, M, blockIdx.x * 256 + threadIdx.x, n_stat_blocks * 256, s, q);
        const double bs = block_sum<4>(s, sc[0]);
        const double bq = block_sum<4>(q, sc[1]);
        if (threadIdx.x == 0) {
            stats[blockIdx.x][0] = bs;
            stats[blockIdx.x][1] = bq;
        }
        return;
    }
    const int b = blockIdx.x - n_stat_blocks, nb = gridDim.x - n_stat_blocks;
    constexpr int per_slot = 4 * w3::kWopKs * 512;     // plane-0 elements of one slot
    for (int e = b * 256 + threadIdx.x; e < 2 * MAXL * 2 * per_slot; e += nb * 256) {
        const int j = e & 7, lane = (e >> 3) & 63, ks = (e >> 9) & 7, cb = (e >> 12) & 3, slot = e >> 14;
        const int dir = slot & 1, nl = slot >> 1, n = nl / MAXL, l = nl - n * MAXL;
        if (l >= NL || (dir == 1 && l == 0)) continue;
        const int in_dim = l == 0 ? D : Hd;
        const float* W = params + L.w[n][l];
        const int k = 16 * ks + 8 * (lane >> 5) + j, c = cb * 32 + (lane & 31);
        // forward: B[k][n = out c] = W[c][k]; backward: B[k = out][n = in c] = W[k][c]
        const int row = dir == 0 ? c : k, col = dir == 0 ? k : c;
        const float v = (row < Hd && col < in_dim) ? W[row * in_dim + col] : 0.0f;
        unsigned p0, p1, p2;
        bf3::split3(v, 0.0f, p0, p1, p2);
        const int at = w3::wop_index(slot, cb, ks, 0, lane, j);
        wop3[at] = (unsigned short)p0;
        wop3[at + 512] = (unsigned short)p1;
        wop3[at + 1024] = (unsigned short)p2;
    }
}

// Diagnostic build only (-DK7W_STAMPS, tools/k7w_stamps.py): thread 0 of every workgroup accumulates, per segment of the tile loop,
// the cycles up to its closing barrier (work) and inside it (wait)
#ifdef K7W_STAMPS
__device__ unsigned long long g_k7w_stamps[kMaxSlabs][32];
#define WBAR(k)                                                        \
    do {                                                               \
        if (tid == 0) {                                                \
            const unsigned long long t_ = __builtin_readcyclecounter(); \
            wst[(k)] += t_ - wt0;                                      \
            wt0 = t_;                                                  \
        }                                                              \
        __syncthreads();                                               \
        if (tid == 0) {                                                \
            const unsigned long long t_ = __builtin_readcyclecounter(); \
            wst[10 + (k)] += t_ - wt0;                                 \
            wt0 = t_;                                                  \
        }                                                              \
    } while (0)
// (inside a segment: cycles since the last stamp go to slot k, and stay part of the segment's work)
#define WSUB(k)                                                        \
    do {                                                               \
        if (tid == 0) {                                                \
            const unsigned long long t_ = __builtin_readcyclecounter(); \
            wst[(k)] += t_ - wsub0;                                    \
            wsub0 = t_;                                                \
        }                                                              \
    } while (0)
#define WSUB0() do { if (tid == 0) wsub0 = __builtin_readcyclecounter(); } while (0)
#else
#define WBAR(k) __syncthreads()
#define WSUB(k) do { } while (0)
#define WSUB0() do { } while (0)
#endif

// HK > 0: the hidden layers run HK k-steps of 16 (8: 113..128 units, 6: 81..96; zero-padded) and DK > 0: layer 1 runs DK k-steps
// (state width <= 16 DK, zero-padded) -- every trip count of the matrix chains is then a compile-time constant (the phase-level `cb < HB` stays a run-time
// test even so: as a constant it let the compiler's code motion loose across the phases and cost the three-layer build 51 spilled
// registers, among them freshly loaded weight slices).  With run-time counts each k-step sat in
// its own branch, and the wait the compiler places at such a join is lgkmcnt(0): the fragments requested for the NEXT k-step were
// waited for before the current one's products were issued (75-87 cycles per matrix instruction, tools/k7w_stamps.py).
template <int NL, int HK, int DK>
__global__ __launch_bounds__(256, 1) void k_mlpw3_step(const WideArgs a) {
    using namespace w3;
#ifdef K7W_STAMPS
    unsigned long long wst[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) wst[k] = 0;
    unsigned long long wt0 = 0, wsub0 = 0;
    const unsigned long long wt_entry = __builtin_readcyclecounter();
#endif
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    __shared__ double s_red[2][kThreads / kWave];
    __shared__ float s_mean, s_std;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int net = (int)(blockIdx.x & 1), cb = w, pair = (int)(blockIdx.x >> 1);
    const int D = a.D, A = a.A, Hd = a.Hd;
    const int HB = (Hd + 31) >> 5, DB = (D + 31) >> 5;
    constexpr bool FULL = HK > 0;
    constexpr int HBK = (HK + 1) / 2;         // column blocks of 32 (compile-time builds)
    const int nksD = DK > 0 ? DK : (D + 15) >> 4, nksH = FULL ? HK : (Hd + 15) >> 4, nks2H = FULL ? HBK : (Hd + 31) >> 5;
    const int AW = a.continuous ? A : 1;
    const int out_dim = net == 0 ? A : 1;

    char* const sX = ldsb + oX;
    char* const sW3 = ldsb + oW3;
    char* const sDo = ldsb + oDo;
    float* const sOut = reinterpret_cast<float*>(ldsb + oOut);
    float* const sAct = reinterpret_cast<float*>(ldsb + oAct);
    float* const sDls = reinterpret_cast<float*>(ldsb + oDls);
    float* const sB = reinterpret_cast<float*>(ldsb + oB);
    float* const sB3 = reinterpret_cast<float*>(ldsb + oB3);
    float* const sLs = reinterpret_cast<float*>(ldsb + oLs);
    float* const sIvar = reinterpret_cast<float*>(ldsb + oIvar);
    float4* const sRec = reinterpret_cast<float4*>(ldsb + oRec);
    int* const sSrc = reinterpret_cast<int*>(ldsb + oSrc);
    int* const sIdx = reinterpret_cast<int*>(ldsb + oIdx);
    auto sH = [&](int l) -> char* { return ldsb + oH + l * 3 * kFPlaneW; };

    // ---- what stays in LDS for the whole launch: zeroed images, W3 image, biases, log-std
    {
        u32x4* z = reinterpret_cast<u32x4*>(ldsb);
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (int e = tid; e < oOut / 16; e += kThreads) z[e] = zero;          // X, H, W3 and dOut images
        for (int e = tid; e < R * LDO; e += kThreads) sDls[e] = 0.0f;
    }
    __syncthreads();
    for (int e = tid; e < AP * HPW; e += kThreads) {
        const int o = e / HPW, i = e - o * HPW;
        if (o < out_dim && i < Hd) store_plain1(sW3, kW3RowW, kW3PlaneW, o, i, a.params[a.L.w[net][NL] + o * Hd + i]);
    }
    for (int e = tid; e < NL * HPW; e += kThreads) {
        const int l = e / HPW, c = e - l * HPW;
        sB[e] = c < Hd ? a.params[a.L.b[net][l] + c] : 0.0f;
    }
    if (tid < AP) {
        sB3[tid] = tid < out_dim ? a.params[a.L.b[net][NL] + tid] : 0.0f;
        const float ls = (a.continuous && tid < A) ? a.params[a.L.logstd + tid] : 0.0f;
        const float sd = expf(ls);
        sLs[tid] = ls;
        sIvar[tid] = 1.0f / (sd * sd);
    }
    {
        double sm = 0.0, q = 0.0;
        for (int b = tid; b < a.n_stat_blocks; b += kThreads) {
            sm += a.stats[2 * b];
            q += a.stats[2 * b + 1];
        }
        const double ts = block_sum<kThreads / kWave>(sm, s_red[0]);
        const double tq = block_sum<kThreads / kWave>(q, s_red[1]);
        if (tid == 0) {
            const double m = ts / (double)a.h.M;
            double var = (tq - ts * m) / (double)(a.h.M - 1);
            if (var < 0.0) var = 0.0;
            s_mean = (float)m;
            s_std = (float)sqrt(var);
        }
    }
    __syncthreads();
    const float mean = s_mean, denom = s_std + 1e-8f;
    const float invM = 1.0f / (float)a.h.M;
    const float g_ent = -a.h.ent_coef * invM;
    // loss lanes: lane k16 of a DPP row works on head output k16 of rows lrow and lrow + 16
    const int k16 = tid & 15, lrow = tid >> 4;
    const float my_ls = sLs[k16], my_ivar = sIvar[k16];
    float ent_c = 0.0f;                                     // the Gaussian's entropy: the same for every row
    for (int k = 0; k < A; ++k) ent_c += (0.5f + 0.9189385332046727f) + sLs[k];

    // persistent accumulators: dW_l blocks (out-block ob, in-block cb), the head's two 16x16 blocks, bias columns
    f32x16 gW[NL][4];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) gW[l][ob] = zero16();
    f32x4 gW3[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float gb[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) gb[l] = 0.0f;
    double l_a = 0, l_b = 0, l_c = 0, l_d = 0, l_e = 0;
    float g_b3c = 0.0f, g_head = 0.0f;

    // ---- this wave's weight slices: a layer's slice (<= 8 k-steps x 3 planes of 16 B per lane) one phase ahead of its use
    // (a wave-uniform base in scalar registers + the lane's 32-bit byte offset + an immediate per fragment; written as 64-bit
    // vector addresses they are formed once, hoisted out of the tile loop and spilled -- every load then waits behind a scratch
    // reload: k_mlp_step3's lesson)
    const char* const wbase0 = reinterpret_cast<const char*>(a.wop3) + 2 * (size_t)(cb * kWopSlice);
    const char* wbase = wbase0;
    int lane16 = lane * 16;
    // (in two halves: k-steps 0..3 are requested one phase ahead and stay live across the epilogue and the barrier in between;
    // k-steps 4..7 are requested at the top of the phase that uses them, behind the first half's 24 matrix instructions -- held
    // whole, the 96 registers of a slice pushed the kernel's vector registers into scratch)
    bf16x8 wreg[3 * kWopKs];
#ifdef K7W_EXP_NO_WLOAD
    int n_loaded = 0, n_loaded_hi = 0;
#endif
    auto load_w = [&](int l, int dir, int nks) {          // first half
#ifdef K7W_EXP_NO_WLOAD       // timing experiment (WRONG results): the weight slices are fetched once per launch
        if (wbase != wbase0 + 0 || n_loaded++) return;
#endif
        int wz = 0;
        asm volatile("" : "+s"(wz));          // (an opaque zero per request: the loads stay where they are asked for)
        const char* m = wbase + wz + 2 * (size_t)(wop_slot(net, l, dir) * kWopSlot);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            if (ks < nks) {
#pragma unroll
                for (int p = 0; p < 3; ++p) wreg[3 * ks + p] = *reinterpret_cast<const bf16x8*>(m + (ks * 3 + p) * 1024 + lane16);
            }
    };
    auto load_w_hi = [&](int l, int dir, int nks) {       // second half
#ifdef K7W_EXP_NO_WLOAD
        if (n_loaded_hi++) return;
#endif
        int wz = 0;
        asm volatile("" : "+s"(wz));
        const char* m = wbase + wz + 2 * (size_t)(wop_slot(net, l, dir) * kWopSlot);
#pragma unroll
        for (int ks = 4; ks < kWopKs; ++ks)
            if (ks < nks) {
#pragma unroll
                for (int p = 0; p < 3; ++p) wreg[3 * ks + p] = *reinterpret_cast<const bf16x8*>(m + (ks * 3 + p) * 1024 + lane16);
            }
    };
    auto wfrag = [&](int ks) {
        Frag3 f;
        f.p[0] = wreg[3 * ks + 0];
        f.p[1] = wreg[3 * ks + 1];
        f.p[2] = wreg[3 * ks + 2];
        return f;
    };

    const int n_tiles = (a.h.M + R - 1) / R;
    // staging: 32 chunk slots of 4 columns per row (128 state floats), four slots per thread
    const bool vec4 = (D % 4 == 0) && ((reinterpret_cast<size_t>(a.obs) & 15) == 0);
    unsigned xok = 0u;                // which staged pieces are real (bit u: state piece u of this thread; bit 16 + u: action piece u)
    float4 p_rec = make_float4(0.f, 0.f, 0.f, 0.f);
    int p_src = -1;
    auto load_idx = [&](int tile) -> int {
        const int m = tile * R + tid;
        return (tile < n_tiles && m < a.h.M) ? a.idx[m] : -1;
    };
    const float* const act_base = a.actions ? a.actions : reinterpret_cast<const float*>(a.rec) + 4;
    const int act_stride = a.actions ? AW : 16;
    char* const sStage = ldsb + oStage;
    char* const sStageA = ldsb + oStageA;
    // The next tile's rows go from HBM straight into an LDS staging area (LDS-DMA: lane l of a wave writes 16 / 4 bytes at the
    // wave's base + l x 16 / 4, which IS the [row][column] order of the pieces a wave asks for), unconditionally (a piece that is
    // not real reads element 0; `xok` remembers which are).  As loads into registers -- sixteen values per thread held for a whole
    // tile in a kernel that has no register to spare -- their destinations doubled as temporaries, and the waits the compiler put
    // in front of those reuses stood the wave through the gathers' HBM latency in the middle of layer 1 (tools/k7w_stamps.py: F1 took
    // 7.5-8.1 k cycles for its 24 matrix instructions).
    auto prefetch = [&](const int* sidx) {
        int tq = tid;
        asm volatile("" : "+v"(tq));       // (an opaque copy: what is derived from it is formed here each time, not once and spilled)
        // (the rows' indices are read from LDS first, all of them, and a piece that is not real is pointed at element 0 by 32-bit
        // selects: written as `ok ? row * D + c : 0` the 64-bit product sat in a branch per piece, each behind its own LDS wait --
        // 1.8 k cycles to issue eight requests, tools/k7w_stamps.py)
        xok = 0u;
        if (vec4) {
            int src[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) src[u] = sidx[(tq + u * kThreads) >> 5];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c4 = (tq & 31) * 4;
                const bool ok = src[u] >= 0 && c4 < D;
                const unsigned row = ok ? (unsigned)src[u] : 0u, col = ok ? (unsigned)c4 : 0u;
                dma16(a.obs + ((size_t)row * (unsigned)D + col), sStage + (w * 64 + u * kThreads) * 16);
                xok |= ok ? (1u << u) : 0u;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = tq + u * kThreads, r = e >> 7, c = e & 127;
                const int s1 = sidx[r];
                const bool ok = s1 >= 0 && c < D;
                const unsigned row = ok ? (unsigned)s1 : 0u, col = ok ? (unsigned)c : 0u;
                dma4(a.obs + ((size_t)row * (unsigned)D + col), sStage + (w * 64 + u * kThreads) * 4);
                xok |= ok ? (1u << u) : 0u;
            }
        }
        if (net == 0) {
            int sa[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) sa[u] = sidx[(tq + u * kThreads) >> 4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = tq & 15;
                const bool ok = sa[u] >= 0 && c < AW;
                const unsigned row = ok ? (unsigned)sa[u] : 0u, col = ok ? (unsigned)c : 0u;
                dma4(act_base + ((size_t)row * (unsigned)act_stride + col), sStageA + (w * 64 + u * kThreads) * 4);
                xok |= ok ? (1u << (16 + u)) : 0u;
            }
        }
        // (every thread, no branch: a conditional load is a write to its destination on the other path, and the wait the compiler
        // puts in front of that write is a wait for every load above it)
        p_src = sidx[tq & (R - 1)];
        p_rec = a.rec[(size_t)(p_src >= 0 ? p_src : 0) * a.rec_stride];
    };
    // the index row of the tile three ahead, the same way: raw value + whether it is real
    int n_raw = 0;
    bool n_ok = false;
    auto prefetch_idx = [&](int tile) {
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int m = tile * R + (tq & (R - 1));
        n_ok = tile < n_tiles && m < a.h.M;
        n_raw = a.idx[n_ok ? m : 0];
    };
    __shared__ int s_tile[4];
    unsigned* const ctr = a.tile_counter + net;
    const bool stat = a.static_tiles != 0;
    const int n_wg = (int)(gridDim.x >> 1);
    if (tid == 0) {
        if (stat) {
            s_tile[0] = pair; s_tile[1] = pair + n_wg; s_tile[2] = pair + 2 * n_wg; s_tile[3] = pair + 3 * n_wg;
        } else {
            const int t0 = (int)atomicAdd(ctr, 4u);
            s_tile[0] = t0; s_tile[1] = t0 + 1; s_tile[2] = t0 + 2; s_tile[3] = t0 + 3;
        }
    }
    if (cb < HB) load_w(0, 0, nksD);                        // layer 1's slice for the first tile
    __syncthreads();
    if (tid < R) {
        sIdx[tid] = load_idx(s_tile[0]);
        sIdx[R + tid] = load_idx(s_tile[1]);
    }
    __syncthreads();
    prefetch(sIdx);
    prefetch_idx(s_tile[2]);

#ifdef K7W_STAMPS
    wt0 = __builtin_readcyclecounter();
    wst[20] = wt0 - wt_entry;            // prologue
#endif
    float om[NL][16];                 // 1 - H_l^2 of this wave's blocks, from the forward to the backward phases
    for (int it = 0;; ++it) {
        const int tile = s_tile[it & 3];
        if (tile >= n_tiles) break;
#ifdef K7W_STAMPS
        wst[21] += 1;                    // tiles
#endif
        const int tile3 = s_tile[(it + 3) & 3];
        int ln = lane;
        asm volatile("" : "+v"(ln));          // opaque per-tile copy: LDS addresses are re-derived inside the phases, not hoisted and spilled
        {
            int wz = 0;                        // ... and the weight base (an opaque zero keeps it a scalar pointer per tile)
            asm volatile("" : "+s"(wz));
            wbase = wbase0 + wz;
            lane16 = ln * 16;
        }
        // ---- land the prefetched tile as bf16 planes
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's own pieces have landed in the staging area
        int tl = tid;
        asm volatile("" : "+v"(tl));
        if (vec4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = tl + u * kThreads, r = e >> 5, c4 = (e & 31) * 4;
                const bool ok = (xok >> u) & 1u;
                const float4 v = *reinterpret_cast<const float4*>(sStage + e * 16);
                if (c4 < D) store_x4(sX + (c4 >> 6) * kXHalf, r, c4 & 63, ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = tl + u * kThreads, r = e >> 7, c = e & 127;
                const float v = *reinterpret_cast<const float*>(sStage + e * 4);
                if (c < D) store_x1(sX + (c >> 6) * kXHalf, r, c & 63, ((xok >> u) & 1u) ? v : 0.0f);
            }
        }
        if (net == 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = tl + u * kThreads;
                const float v = *reinterpret_cast<const float*>(sStageA + e * 4);
                sAct[(e >> 4) * LDO + (e & 15)] = ((xok >> (16 + u)) & 1u) ? v : 0.0f;
            }
        }
        if (tid < R) {
            sSrc[tid] = p_src;
            sRec[tid] = p_rec;
            sIdx[(it & 1) * R + tid] = n_ok ? n_raw : -1;
        }
        WBAR(0);

        // ---- forward: H_1 .. H_NL
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            f32x16 acc = zero16();
            WSUB0();
            if (cb < HB) {
                // (every chain below asks for the fragments of k-step ks + 1 BEFORE it issues the six products of k-step ks: with the
                // reads and the products of a pair of k-steps in one scheduling region the compiler put the reads first and the wave
                // waited out their latency in front of every pair -- 75-87 cycles per matrix instruction instead of 32,
                // tools/k7w_stamps.py)
                int lc = ln;
                asm volatile("" : "+v"(lc));       // (an opaque copy per chain: its fragment addresses are formed here, not at the top of the tile)
                if (l == 0) {
                    load_w_hi(0, 0, nksD);
                    Frag3 f0 = x_rows(sX, 0, lc), f1 = f0;          // two fragment sets take turns (no copies between them)
#pragma unroll
                    for (int ks = 0; ks < kWopKs; ks += 2) {
                        if (DK > 0 ? ks + 1 < DK : ks + 1 < nksD) f1 = x_rows(sX + ((ks + 1) >> 2) * kXHalf, (ks + 1) & 3, lc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (DK > 0 ? ks < DK : ks < nksD) acc = mma32x3(f0, wfrag(ks), acc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (DK > 0 ? ks + 2 < DK : ks + 2 < nksD) f0 = x_rows(sX + ((ks + 2) >> 2) * kXHalf, (ks + 2) & 3, lc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (DK > 0 ? ks + 1 < DK : ks + 1 < nksD) acc = mma32x3(f1, wfrag(ks + 1), acc);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    load_w_hi(l, 0, nksH);
                    Frag3 f0 = f_cols_p<kFPlaneW>(sH(l - 1), 0, lc), f1 = f0;
#pragma unroll
                    for (int ks = 0; ks < kWopKs; ks += 2) {
                        if (FULL ? ks + 1 < HK : ks + 1 < nksH) f1 = f_cols_p<kFPlaneW>(sH(l - 1), ks + 1, lc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (FULL ? ks < HK : ks < nksH) acc = mma32x3(f0, wfrag(ks), acc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (FULL ? ks + 2 < HK : ks + 2 < nksH) f0 = f_cols_p<kFPlaneW>(sH(l - 1), ks + 2, lc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (FULL ? ks + 1 < HK : ks + 1 < nksH) acc = mma32x3(f1, wfrag(ks + 1), acc);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef K7W_STAMPS
            { float sink = acc[0] + acc[15]; asm volatile("" :: "v"(sink)); }      // the chain has drained
#endif
            WSUB(l == 0 ? 24 : 28);
            if (l == 0) {
                // the next tile's rows, behind this tile's math.  Requested HERE, behind layer 1's products: the wait for their weight
                // slice (requested in the previous trip of the loop) is a vmcnt(0), and ahead of it these gathers' whole HBM latency
                // sat in every tile (tools/k7w_stamps.py: F1 took 8.1 k cycles for its 24 matrix instructions)
                prefetch(sIdx + ((it + 1) & 1) * R);
                prefetch_idx(tile3);
            }
            WSUB(25);
            if (cb < HB) {
                // the slice used next: the following layer's forward copy, or (behind the last layer) the top layer's backward copy
                if (l + 1 < NL) load_w(l + 1, 0, nksH);
                else if (NL > 1) load_w(NL - 1, 1, nksH);
                tanh_store_p<kFPlaneW>(sH(l), cb * 32, acc, sB[l * HPW + cb * 32 + (ln & 31)], ln, om[l]);
            }
            WSUB(l == 0 ? 26 : 29);
            WBAR(1 + l);
        }
        if (cb < 2) {   // head: 16 rows per wave on 16x16x32
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                if (ks < nks2H)
                    acc = mma16x3(f_cols16_p<kFPlaneW>(sH(NL - 1), 16 * cb, ks, ln),
                                  plain_rows(sW3, kW3RowW, kW3PlaneW, ln & 15, 32 * ks + 8 * (ln >> 4)), acc);
            const int col = ln & 15;
            const float bias = sB3[col];
#pragma unroll
            for (int e = 0; e < 4; ++e) sOut[(cb * 16 + 4 * (ln >> 4) + e) * LDO + col] = acc[e] + bias;
        }
        WBAR(4);
        int tile4 = 0;
        if (tid == 0) tile4 = stat ? tile + 4 * n_wg : (int)atomicAdd(ctr, 1u);

        // ---- loss lanes: this net's half of the PPO terms; head outputs become their gradients, fp32 (column sums) and as a
        // bf16-plane image (the two products below).  SIXTEEN lanes per row (lane k = head output k; rows tid >> 4 and + 16), the
        // sums over the outputs by butterfly inside the 16 lanes: as one lane per row walking its outputs, this phase was 6.2 k
        // cycles of a 51.6 k-cycle tile with 224 of the workgroup's threads waiting (tools/k7w_stamps.py)
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {      // (not unrolled: two passes' temporaries at once cost the 3-layer build 130 spilled registers)
            const int row = lrow + 16 * pass;
            float* const out = sOut + row * LDO;
            const bool live = sSrc[row] >= 0;
            const float4 rc = sRec[row];
            const float o_k = out[k16];
            float g_out = 0.0f;                       // d loss / d head output k of this row
            if (net == 1) {
                if (k16 == 0 && live) {
                    const PpoSample t = ppo_sample(rc.x, rc.x, rc.y, o_k, rc.w, rc.z, mean, denom, invM, a.h);
                    l_a += t.vl;
                    g_out = t.g_v;
                    g_b3c += t.g_v;
                }
            } else if (a.continuous) {
                const float zk = sAct[row * LDO + k16] - o_k;
                const float logp = row16_sum(k16 < A ? (-(zk * zk) * (0.5f * my_ivar) - my_ls) - 0.9189385332046727f : 0.0f);
                const PpoSample t = ppo_sample(logp, rc.x, rc.y, rc.w, rc.w, rc.z, mean, denom, invM, a.h);
                if (live) {
                    if (k16 == 0) { l_a += t.pg; l_b += ent_c; l_c += t.okl; l_d += t.kl; l_e += t.cf; }
                    if (k16 < A) {
                        g_out = t.g_logp * (zk * my_ivar);
                        sDls[row * LDO + k16] = t.g_logp * (zk * zk * my_ivar - 1.0f) + g_ent;
                    }
                } else {
                    sDls[row * LDO + k16] = 0.0f;
                }
            } else {
                const bool in = k16 < A;
                const float mx = row16_max(in ? o_k : -3.0e38f);
                const float se = row16_sum(in ? expf(o_k - mx) : 0.0f);
                const float lse = mx + logf(se);
                const int ai = (int)sAct[row * LDO];
                const float lpk = o_k - lse, pk = in ? expf(lpk) : 0.0f;
                const float ent = -row16_sum(in ? pk * lpk : 0.0f);
                const float logp = row16_sum(in && k16 == ai ? lpk : 0.0f);
                const PpoSample t = ppo_sample(logp, rc.x, rc.y, rc.w, rc.w, rc.z, mean, denom, invM, a.h);
                if (live) {
                    if (k16 == 0) { l_a += t.pg; l_b += ent; l_c += t.okl; l_d += t.kl; l_e += t.cf; }
                    if (in) g_out = t.g_logp * ((k16 == ai ? 1.0f : 0.0f) - pk) + g_ent * (-pk * (lpk + ent));
                }
                sDls[row * LDO + k16] = 0.0f;
            }
            out[k16] = g_out;
            store_plain1(sDo, kDoRowW, kDoPlaneW, k16, row, g_out);
        }
        WBAR(5);

        // ---- head backward: column sums (d b3, d logstd), dH_NL -> dZ_NL (in place), dW3
        if (net == 0 && w == 3 && lane < 2 * AP) {
            const float* src = lane < AP ? sOut + lane : sDls + (lane - AP);
            float cs = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) cs += src[r * LDO];
            g_head += cs;
        }
        if (cb < HB) {
            char* const HL = sH(NL - 1);
            f32x16 acc = zero16();
            acc = mma32x3(plain_cols(sDo, kDoRowW, kDoPlaneW, 0, 0, ln), plain_cols(sW3, kW3RowW, kW3PlaneW, 0, cb * 32, ln), acc);
            {
                const Frag3 da = plain_rows(sDo, kDoRowW, kDoPlaneW, ln & 15, 8 * (ln >> 4));
#pragma unroll
                for (int q = 0; q < 2; ++q) gW3[q] = mma16x3(da, f_rows16_p<kFPlaneW>(HL, cb * 32 + 16 * q, ln), gW3[q]);
            }
            float colsum = dz_from_regs_p<kFPlaneW>(HL, cb * 32, acc, om[NL - 1], ln);
            colsum += __shfl_xor(colsum, 32, kWave);
            gb[NL - 1] += colsum;
        }
        WBAR(6);
        // ---- hidden layers, top down: dW_l, dH_{l-1} -> dZ_{l-1} (in place)
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            if (cb < HB) {
                const char* const dZ = sH(l);
                char* const Hp = sH(l - 1);
                load_w_hi(l, 1, nksH);                      // (behind the dW chains below)
                int lc = ln;
                asm volatile("" : "+v"(lc));
                {   // dW_l: eight (k-step, out-block) products, the next one's dZ fragment (and the next k-step's H fragment) asked for first
                    Frag3 hb0 = f_rows_p<kFPlaneW>(Hp, cb * 32, 0, lc), hb1 = hb0;
                    Frag3 d0 = f_rows_p<kFPlaneW>(dZ, 0, 0, lc), d1 = d0;
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {            // q = 4 ks + ob
                        if (FULL ? ((q + 1) & 3) < HBK : ((q + 1) & 3) < HB) d1 = f_rows_p<kFPlaneW>(dZ, ((q + 1) & 3) * 32, (q + 1) >> 2, lc);
                        if (q == 2) hb1 = f_rows_p<kFPlaneW>(Hp, cb * 32, 1, lc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (FULL ? (q & 3) < HBK : (q & 3) < HB) gW[l][q & 3] = mma32x3(d0, q < 4 ? hb0 : hb1, gW[l][q & 3]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (q + 2 < 8 && (FULL ? ((q + 2) & 3) < HBK : ((q + 2) & 3) < HB)) d0 = f_rows_p<kFPlaneW>(dZ, ((q + 2) & 3) * 32, (q + 2) >> 2, lc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (FULL ? ((q + 1) & 3) < HBK : ((q + 1) & 3) < HB) gW[l][(q + 1) & 3] = mma32x3(d1, q < 4 ? hb0 : hb1, gW[l][(q + 1) & 3]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                f32x16 acc = zero16();
                {
                    Frag3 f0 = f_cols_p<kFPlaneW>(dZ, 0, lc), f1 = f0;
#pragma unroll
                    for (int ks = 0; ks < kWopKs; ks += 2) {
                        if (FULL ? ks + 1 < HK : ks + 1 < nksH) f1 = f_cols_p<kFPlaneW>(dZ, ks + 1, lc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (FULL ? ks < HK : ks < nksH) acc = mma32x3(f0, wfrag(ks), acc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (FULL ? ks + 2 < HK : ks + 2 < nksH) f0 = f_cols_p<kFPlaneW>(dZ, ks + 2, lc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (FULL ? ks + 1 < HK : ks + 1 < nksH) acc = mma32x3(f1, wfrag(ks + 1), acc);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (l - 1 >= 1) load_w(l - 1, 1, nksH);
                else load_w(0, 0, nksD);                    // layer 1's forward slice for the next tile
                float colsum = dz_from_regs_p<kFPlaneW>(Hp, cb * 32, acc, om[l - 1], ln);
                colsum += __shfl_xor(colsum, 32, kWave);
                gb[l - 1] += colsum;
            }
            WBAR(6 + l);
        }
        // ---- dW_1: this wave's 32 state columns against every out-block
        if (cb < DB) {
            const char* const dZ = sH(0);
            int lc = ln;
            asm volatile("" : "+v"(lc));
            Frag3 xb0 = x_cols(sX + (cb >> 1) * kXHalf, 0, (cb & 1) * 32, lc), xb1 = xb0;
            Frag3 d0 = f_rows_p<kFPlaneW>(dZ, 0, 0, lc), d1 = d0;
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                if (FULL ? ((q + 1) & 3) < HBK : ((q + 1) & 3) < HB) d1 = f_rows_p<kFPlaneW>(dZ, ((q + 1) & 3) * 32, (q + 1) >> 2, lc);
                if (q == 2) xb1 = x_cols(sX + (cb >> 1) * kXHalf, 1, (cb & 1) * 32, lc);
                __builtin_amdgcn_sched_barrier(0);
                if (FULL ? (q & 3) < HBK : (q & 3) < HB) gW[0][q & 3] = mma32x3(d0, q < 4 ? xb0 : xb1, gW[0][q & 3]);
                __builtin_amdgcn_sched_barrier(0);
                if (q + 2 < 8 && (FULL ? ((q + 2) & 3) < HBK : ((q + 2) & 3) < HB)) d0 = f_rows_p<kFPlaneW>(dZ, ((q + 2) & 3) * 32, (q + 2) >> 2, lc);
                __builtin_amdgcn_sched_barrier(0);
                if (FULL ? ((q + 1) & 3) < HBK : ((q + 1) & 3) < HB) gW[0][(q + 1) & 3] = mma32x3(d1, q < 4 ? xb0 : xb1, gW[0][(q + 1) & 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (tid == 0) s_tile[it & 3] = tile4;
        WBAR(9);
    }

#ifdef K7W_STAMPS
    if (tid == 0) {
        wst[22] = __builtin_readcyclecounter() - wt_entry;     // entry to the end of the tile loop
#pragma unroll
        for (int k = 0; k < 32; ++k) g_k7w_stamps[blockIdx.x][k] = wst[k];
    }
#endif
    // ---- this workgroup's half of the pair's slab
    float* slab = a.slabs + (size_t)pair * a.L.n_params;
    {
        const int col = cb * 32 + (lane & 31);
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int in_dim = l == 0 ? D : Hd;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                if (ob < HB && col < in_dim) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int o = ob * 32 + acc_row(e, lane);
                        if (o < Hd) slab[a.L.w[net][l] + o * in_dim + col] = gW[l][ob][e];
                    }
                }
            }
            if (lane < 32 && col < Hd) slab[a.L.b[net][l] + col] = gb[l];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {       // 16x16 accumulator layout: row = head output, col = hidden unit
            const int o = 4 * (lane >> 4) + e, c = cb * 32 + (lane & 15);
            if (o < out_dim && cb < HB) {
                if (c < Hd) slab[a.L.w[net][NL] + o * Hd + c] = gW3[0][e];
                if (c + 16 < Hd) slab[a.L.w[net][NL] + o * Hd + c + 16] = gW3[1][e];
            }
        }
    }
    if (net == 0 && w == 3) {
        if (lane < A) slab[a.L.b[0][NL] + lane] = g_head;
        if (a.continuous && lane >= AP && lane - AP < A) slab[a.L.logstd + lane - AP] = g_head;
    }
    {   // the loss sums and the critic's bias gradient sit on lane 0 of every DPP row of all four waves
        __shared__ double s_fin[kThreads / kWave][6];
        double v6[6] = {l_a, l_b, l_c, l_d, l_e, (double)g_b3c};
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            double x = k16 == 0 ? v6[q] : 0.0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, kWave);
            if (lane == 0) s_fin[w][q] = x;
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) v6[q] = ((s_fin[0][q] + s_fin[1][q]) + s_fin[2][q]) + s_fin[3][q];
            double* lp = a.loss_part + (size_t)pair * 8;
            if (net == 0) {
                lp[0] = v6[0]; lp[2] = v6[1]; lp[3] = v6[2]; lp[4] = v6[3]; lp[5] = v6[4];
            } else {
                slab[a.L.b[1][NL]] = (float)v6[5];
                lp[1] = v6[0];
                lp[6] = (double)mean;
                lp[7] = (double)s_std;
            }
        }
    }
}

// K8w: policy.evaluate(next_obs) under no_grad + the three buffer row stores (src/ppo.py:103-108), or value only
// (noise == nullptr, src/ppo.py:161).  Workgroup = (row tile, net).
template <int NL>
__global__ __launch_bounds__(256) void k_mlpw_act(const WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const WideLds s = carve<false, NL>(lds, 0, 1);
    const int tid = threadIdx.x;
    const int net = a.net_base + (int)(blockIdx.x % a.net_count), row0 = (int)(blockIdx.x / a.net_count) * R;
    const int D = a.D, A = a.A, HB = (a.Hd + 31) >> 5, DB = (D + 31) >> 5;
    stage_small<NL, false>(a, s, net, true);
    __syncthreads();
    {
        const int r = tid >> 3, n = row0 + r;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = (tid & 7) + 8 * u;
            if (c < D) s.sX[r * LDW + c] = n < a.N ? a.obs[(size_t)n * D + c] : 0.0f;
        }
    }
    __syncthreads();
    float early[16];
    if ((tid >> 6) < HB) load_b(early, op_block(a.wop, net, 0, 0, tid >> 6, 0, tid & 63));
    forward_tile<NL, true, false>(a, s, net, tid >> 6, HB, DB, early);
    if (tid >= R || row0 + tid >= a.N) return;
    const int n = row0 + tid;
    const float* mu = s.sOut + tid * LDO;
    if (net == 1) {
        a.out_value[n] = mu[0];
        return;
    }
    if (!a.noise) return;
    if (a.continuous) {
        float lp = 0.0f;
        for (int k = 0; k < A; ++k) {
            const float ls = s.sLs[k], sd = s.sIvar[k];
            const float act = mu[k] + sd * a.noise[(size_t)n * A + k];
            a.out_actions[(size_t)n * A + k] = act;
            const float z = act - mu[k];                               // as evaluate() forms it: (a - mu)
            lp += (-(z * z) / (2.0f * (sd * sd)) - ls) - 0.9189385332046727f;
        }
        a.out_logp[n] = lp;
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
        a.out_actions[n] = (float)pick;
        a.out_logp[n] = mu[pick] - lse;
    }
}

struct WideWs {
    double* stats;       // (kStatBlocks, 2)
    double* loss_part;   // (kMaxSlabs, 8)
    float* wop;          // kOpFloats
    unsigned* tile_counter;   // [2] (+ padding to 64 B)
    float* slabs;        // (kMaxSlabs, n_params)
    double* sq_part;     // (ceil(n_params / 64)) clip partial sums left by k_mlp_reduce
    unsigned short* wop3;   // k_mlpw3_step's bf16-plane operand copies (w3::kWopElems), behind everything else
};
// slabs a launch can write: two both-net workgroups per CU for the narrow shapes (small n_params), one pair per two CUs otherwise
int wide_slab_cap(int hidden, int D) { return (hidden <= 64 && D <= 64) ? kMaxSlabs : kMaxGrid / 2; }
WideWs wide_ws(void* workspace, int n_params, int slab_cap) {
    WideWs v;
    char* p = reinterpret_cast<char*>(workspace);
    v.stats = reinterpret_cast<double*>(p);
    v.loss_part = v.stats + 2 * kStatBlocks;
    v.wop = reinterpret_cast<float*>(v.loss_part + 8 * kMaxSlabs);
    v.tile_counter = reinterpret_cast<unsigned*>(v.wop + kOpFloats);
    v.slabs = reinterpret_cast<float*>(v.tile_counter + 16);
    v.sq_part = reinterpret_cast<double*>(v.slabs + (((size_t)slab_cap * (size_t)n_params + 15) / 16) * 16);
    v.wop3 = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(v.sq_part) +
                                               ((sizeof(double) * (size_t)((n_params + 63) / 64) + 63) / 64) * 64);
    return v;
}

int fill_layout(WideLayout& L, const int* layout_h, int NL, int continuous, int n_params, const char* who) {
    // layout_h: for net in (actor, critic): w_0, b_0, ..., w_NL, b_NL (layer NL = head); then logstd
    const int per = 2 * (NL + 1);
    for (int n = 0; n < 2; ++n)
        for (int l = 0; l <= NL; ++l) {
            L.w[n][l] = layout_h[n * per + 2 * l];
            L.b[n][l] = layout_h[n * per + 2 * l + 1];
        }
    L.logstd = continuous ? layout_h[2 * per] : 0;
    L.n_params = n_params;
    for (int k = 0; k < 2 * per + (continuous ? 1 : 0); ++k)
        AURPPO_REQUIRE(layout_h[k] >= 0 && layout_h[k] < n_params, AURPPO_ESHAPE, "%s: layout[%d]=%d", who, k, layout_h[k]);
    return AURPPO_OK;
}

int check_shape(int D, int A, int continuous, int hidden, int num_layers, const char* who) {
    AURPPO_REQUIRE(hidden >= 1 && hidden <= HPW, AURPPO_ESHAPE, "%s: hidden_dim=%d must be 1..%d", who, hidden, HPW);
    AURPPO_REQUIRE(num_layers >= 1 && num_layers <= MAXL, AURPPO_ESHAPE, "%s: num_layers=%d must be 1..%d", who, num_layers, MAXL);
    AURPPO_REQUIRE(D >= 1 && D <= HPW, AURPPO_ESHAPE, "%s: state_dim=%d must be 1..%d", who, D, HPW);
    AURPPO_REQUIRE(A >= 1 && A <= AP && (continuous || A >= 2), AURPPO_ESHAPE,
                   "%s: action_dim=%d must be 1..%d (>= 2 logits for a Categorical head)", who, A, AP);
    return AURPPO_OK;
}

template <class K>
int launch_wide(K kernel, bool* attr_done, int grid, size_t lds_bytes, hipStream_t s, const WideArgs& a) {
    if (!*attr_done) {
        AURPPO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes));
        *attr_done = true;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), lds_bytes, s, a);
    return AURPPO_OK;
}

}  // namespace

extern "C" size_t aurppo_mlp_wide_workspace_bytes(int n_params, int hidden, int state_dim) {
    // the slab region is sized from the shape (512 slabs of a narrow policy's few parameters, 128 of a wide one's many: 52 MB
    // instead of 207 MB at 3 x 128 over 128 state floats); n_params = 0: the operand copies alone, all K8w needs
    const size_t slabs = (size_t)wide_slab_cap(hidden, state_dim) * (size_t)(n_params > 0 ? n_params : 0);
    return sizeof(double) * (2 * kStatBlocks + 8 * kMaxSlabs) + sizeof(float) * (size_t)kOpFloats + 64 +
           sizeof(float) * slabs + 64 + sizeof(double) * (size_t)((n_params + 63) / 64) + 128 +
           sizeof(unsigned short) * (size_t)w3::kWopElems + 64;
}

// Which kernel aurppo_mlp_wide_ppo_*_f32 launches for a net shape: 1 = k_mlpw_step<., true> (both nets per workgroup, fp32 MFMA:
// nothing wider than 64), 2 = k_mlpw_step<., false> (one net per workgroup, fp32 MFMA), 3 = k_mlpw3_step (one net per workgroup,
// bf16 MFMA over three-way splits -- the default for the wider shapes; AURPPO_K7W_VARIANT=2 selects 2).
extern "C" int aurppo_k7w_kernel(int hidden, int state_dim) {
    if (hidden <= 64 && state_dim <= 64) return 1;
    return aurppo_knobs().k7w_variant == 3 ? 3 : 2;
}

namespace {
struct WideTail {   // the optimizer half of aurppo_mlp_wide_ppo_minibatch_f32
    float* params_rw;
    float* exp_avg;
    float* exp_avg_sq;
    double max_norm;
    const float* lr_dev;
    float* step_dev;
    double beta1, beta2, eps;
    float* out_norm;
    const int32_t* next_idx;   // the slice stepped next: its statistics (and the refreshed operand copies) are left by this call
    int next_M;
    int chained;               // the previous call named this idx as its next_idx: no prepare pass
};
}  // namespace

static int wide_step_impl(const float* obs, const float* actions, const float* rec, const int32_t* idx, int M, int D, int A,
                          int continuous, int hidden, int num_layers, const float* params, const int* layout_h, int n_params,
                          float* grads, double clip, double ent_coef, double vf_coef, int norm_adv, int vloss_mode,
                          float* out_scalars, void* workspace, void* stream, void* ev_begin, void* ev_end, const WideTail* tail,
                          const char* who) {
    AURPPO_REQUIRE(obs && rec && idx && params && layout_h && grads && out_scalars && workspace, AURPPO_EINVAL, "%s: null pointer", who);
    AURPPO_REQUIRE(actions || (continuous ? A : 1) <= 12, AURPPO_ESHAPE,
                   "%s: packed records hold at most 12 action floats (action_dim=%d)", who, A);
    int rc = check_shape(D, A, continuous, hidden, num_layers, who);
    if (rc != AURPPO_OK) return rc;
    AURPPO_REQUIRE(M > 0 && n_params > 0, AURPPO_ESHAPE, "%s: M=%d n_params=%d", who, M, n_params);
    AURPPO_REQUIRE(vloss_mode >= 0 && vloss_mode <= 2, AURPPO_EINVAL, "%s: bad vloss_mode %d", who, vloss_mode);
    AURPPO_REQUIRE(aligned_to(workspace, 64) && aligned_to(rec, 16), AURPPO_EINVAL, "%s: workspace not 64-byte / rec not 16-byte aligned", who);
    WideArgs a = {};
    a.obs = obs; a.actions = actions; a.rec = reinterpret_cast<const float4*>(rec); a.rec_stride = actions ? 1 : 4;
    a.idx = idx; a.params = params;
    a.D = D; a.A = A; a.Hd = hidden; a.continuous = continuous ? 1 : 0;
    rc = fill_layout(a.L, layout_h, num_layers, continuous, n_params, who);
    if (rc != AURPPO_OK) return rc;
    a.h = make_hyper(M, clip, ent_coef, vf_coef, norm_adv, vloss_mode);
    const WideWs wv = wide_ws(workspace, n_params, wide_slab_cap(hidden, D));
    a.stats = wv.stats; a.loss_part = wv.loss_part; a.wop = wv.wop; a.slabs = wv.slabs; a.tile_counter = wv.tile_counter;
    a.wop3 = wv.wop3;
    hipStream_t s = (hipStream_t)stream;
    int sb = (M + 1023) / 1024;
    if (sb > kStatBlocks) sb = kStatBlocks;
    a.n_stat_blocks = sb;
    // layers and state at most two 32-column blocks wide: one workgroup carries both nets (k_mlpw_step<., true>, fp32 MFMA);
    // anything wider: one net per workgroup on the bf16 pipe (k_mlpw3_step; AURPPO_K7W_VARIANT=2 keeps the fp32-MFMA k_mlpw_step)
    const bool dual = hidden <= 64 && D <= 64;
    const AurppoKnobs& knobs = aurppo_knobs();
    const bool bf3k = !dual && knobs.k7w_variant == 3;
    if (!(tail && tail->chained)) {   // otherwise the previous chained call's optimizer launch has left all of this
        if (bf3k)
            hipLaunchKernelGGL(k_mlpw3_prep, dim3(sb + 96), dim3(256), 0, s, params, a.L, num_layers, D, hidden, wv.wop3, a.rec,
                               a.rec_stride, idx, M, reinterpret_cast<double (*)[2]>(wv.stats), sb, wv.tile_counter);
        else
            hipLaunchKernelGGL(k_mlpw_prep, dim3(sb + 96), dim3(256), 0, s, params, a.L, num_layers, D, hidden, wv.wop, a.rec,
                               a.rec_stride, idx, M, reinterpret_cast<double (*)[2]>(wv.stats), sb, wv.tile_counter);
        AURPPO_LAUNCH_CHECK("k_mlpw_prep");
    }
    static int cus_of[kMaxDevices] = {0};
    const int dslot = aurppo_device_slot();
    if (!cus_of[dslot]) {
        hipDeviceProp_t prop;
        AURPPO_HIP_TRY(hipGetDeviceProperties(&prop, dslot));
        cus_of[dslot] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : kMaxGrid;
    }
    const int n_tiles = (M + R - 1) / R;
    // 8 CUs left to the side stream's shuffle kernels, as K7; a both-net workgroup is built to share its CU with a second one
    a.static_tiles = knobs.static_tiles ? 1 : 0;
    const int spare = knobs.k7_spare_cus >= 0 ? knobs.k7_spare_cus : 8;
    int pairs = dual ? 2 * (cus_of[dslot] - spare) : (cus_of[dslot] - spare) / 2;
    if (pairs > (dual ? kMaxSlabs : kMaxGrid / 2)) pairs = dual ? kMaxSlabs : kMaxGrid / 2;
    if (pairs > n_tiles) pairs = n_tiles;
    if (pairs < 1) pairs = 1;
    static bool attr[kMaxDevices][3][MAXL] = {};
    if (ev_begin) AURPPO_HIP_TRY(hipEventRecord((hipEvent_t)ev_begin, s));
    const int grid = dual ? pairs : 2 * pairs;
    bool* ad = &attr[dslot][bf3k ? 2 : (dual ? 1 : 0)][num_layers - 1];
    if (bf3k) {
        // hidden layers of 113..128 / 81..96 units over <= 64 / <= 128 state floats: the builds whose matrix chains have compile-time
        // trip counts (v = 1..4); every other width runs the build with run-time counts (v = 0)
        static bool attr3[kMaxDevices][5][MAXL] = {};
        const int hk = hidden > 112 ? 8 : ((hidden > 80 && hidden <= 96) ? 6 : 0);
        const int v = hk == 0 ? 0 : (hk == 8 ? 1 : 3) + (D <= 64 ? 0 : 1);
        ad = &attr3[dslot][v][num_layers - 1];
        const size_t lds = (size_t)w3::kBytes;
#define AURPPO_W3_LAUNCH(NLV)                                                                                   \
        switch (v) {                                                                                            \
            case 1: rc = launch_wide(k_mlpw3_step<NLV, 8, 4>, ad, grid, lds, s, a); break;                        \
            case 2: rc = launch_wide(k_mlpw3_step<NLV, 8, 8>, ad, grid, lds, s, a); break;                        \
            case 3: rc = launch_wide(k_mlpw3_step<NLV, 6, 4>, ad, grid, lds, s, a); break;                        \
            case 4: rc = launch_wide(k_mlpw3_step<NLV, 6, 8>, ad, grid, lds, s, a); break;                        \
            default: rc = launch_wide(k_mlpw3_step<NLV, 0, 0>, ad, grid, lds, s, a); break;                       \
        }
        if (num_layers == 1) { AURPPO_W3_LAUNCH(1) }
        else if (num_layers == 2) { AURPPO_W3_LAUNCH(2) }
        else { AURPPO_W3_LAUNCH(3) }
#undef AURPPO_W3_LAUNCH
    } else
    switch (num_layers * 2 + (dual ? 1 : 0)) {
        case 2: rc = launch_wide(k_mlpw_step<1, false>, ad, grid, wide_lds_bytes<false, 1>(1), s, a); break;
        case 3: rc = launch_wide(k_mlpw_step<1, true>, ad, grid, wide_lds_bytes<true, 1>(2), s, a); break;
        case 4: rc = launch_wide(k_mlpw_step<2, false>, ad, grid, wide_lds_bytes<false, 2>(1), s, a); break;
        case 5: rc = launch_wide(k_mlpw_step<2, true>, ad, grid, wide_lds_bytes<true, 2>(2), s, a); break;
        case 6: rc = launch_wide(k_mlpw_step<3, false>, ad, grid, wide_lds_bytes<false, 3>(1), s, a); break;
        default: rc = launch_wide(k_mlpw_step<3, true>, ad, grid, wide_lds_bytes<true, 3>(2), s, a); break;
    }
    if (rc != AURPPO_OK) return rc;
    AURPPO_LAUNCH_CHECK("k_mlpw_step");
    if (ev_end) AURPPO_HIP_TRY(hipEventRecord((hipEvent_t)ev_end, s));
    if (!tail) return launch_mlp_reduce(wv.slabs, wv.loss_part, pairs, n_params, a.h, grads, out_scalars, s);
    // (the reduce clears the tile counters for the launch that follows: the next chained call has no prepare pass to do it)
    double* const bc = reinterpret_cast<double*>(wv.tile_counter + 8);     // Adam's bias corrections, reduce -> optimizer launch
    rc = launch_mlp_reduce(wv.slabs, wv.loss_part, pairs, n_params, a.h, grads, out_scalars, s, wv.sq_part, tail->step_dev,
                           wv.tile_counter, tail->beta1, tail->beta2, bc);
    if (rc != AURPPO_OK) return rc;
    WideCopies wc;
    for (int n = 0; n < 2; ++n)
        for (int l = 0; l < 3; ++l) wc.w[n][l] = l < num_layers ? a.L.w[n][l] : n_params;
    wc.NL = num_layers; wc.Hd = hidden; wc.D = D; wc.wop = wv.wop;
    wc.wop3 = bf3k ? wv.wop3 : nullptr;
    return launch_adam_tail(tail->params_rw, grads, tail->exp_avg, tail->exp_avg_sq, n_params, wv.sq_part, tail->max_norm,
                            tail->lr_dev, tail->step_dev, tail->beta1, tail->beta2, tail->eps, tail->out_norm, s,
                            tail->next_idx ? &wc : nullptr, a.rec, a.rec_stride, tail->next_idx, tail->next_M, wv.stats, bc);
}

extern "C" int aurppo_mlp_wide_ppo_step_f32(const float* obs, const float* actions, const float* rec, const int32_t* idx, int M,
                                            int D, int A, int continuous, int hidden, int num_layers, const float* params,
                                            const int* layout_h, int n_params, float* grads, double clip, double ent_coef,
                                            double vf_coef, int norm_adv, int vloss_mode, float* out_scalars, void* workspace,
                                            void* stream, void* ev_begin, void* ev_end) {
    return wide_step_impl(obs, actions, rec, idx, M, D, A, continuous, hidden, num_layers, params, layout_h, n_params, grads, clip,
                          ent_coef, vf_coef, norm_adv, vloss_mode, out_scalars, workspace, stream, ev_begin, ev_end, nullptr,
                          "aurppo_mlp_wide_ppo_step_f32");
}

extern "C" int aurppo_mlp_wide_ppo_minibatch_f32(const float* obs, const float* actions, const float* rec, const int32_t* idx,
                                                 int M, int D, int A, int continuous, int hidden, int num_layers, float* params,
                                                 const int* layout_h, int n_params, float* grads, double clip, double ent_coef,
                                                 double vf_coef, int norm_adv, int vloss_mode, float* out_scalars,
                                                 float* exp_avg, float* exp_avg_sq, double max_norm, const float* lr_dev,
                                                 float* step_dev, double beta1, double beta2, double eps, float* out_norm,
                                                 const int32_t* next_idx, int next_M, int chained, void* workspace, void* stream) {
    AURPPO_REQUIRE(exp_avg && exp_avg_sq && lr_dev && step_dev && out_norm, AURPPO_EINVAL,
                   "aurppo_mlp_wide_ppo_minibatch_f32: null optimizer pointer");
    AURPPO_REQUIRE(!next_idx || next_M > 0, AURPPO_ESHAPE, "aurppo_mlp_wide_ppo_minibatch_f32: next_M=%d", next_M);
    WideTail t;
    t.params_rw = params; t.exp_avg = exp_avg; t.exp_avg_sq = exp_avg_sq; t.max_norm = max_norm; t.lr_dev = lr_dev;
    t.step_dev = step_dev; t.beta1 = beta1; t.beta2 = beta2; t.eps = eps; t.out_norm = out_norm;
    t.next_idx = next_idx; t.next_M = next_idx ? next_M : 0; t.chained = chained ? 1 : 0;
    return wide_step_impl(obs, actions, rec, idx, M, D, A, continuous, hidden, num_layers, params, layout_h, n_params, grads, clip,
                          ent_coef, vf_coef, norm_adv, vloss_mode, out_scalars, workspace, stream, nullptr, nullptr, &t,
                          "aurppo_mlp_wide_ppo_minibatch_f32");
}

extern "C" int aurppo_mlp_wide_act_f32(const float* obs, const float* noise, int N, int D, int A, int continuous, int hidden,
                                       int num_layers, const float* params, const int* layout_h, int n_params, float* actions,
                                       float* logp, float* value, void* workspace, void* stream) {
    const char* who = "aurppo_mlp_wide_act_f32";
    AURPPO_REQUIRE(obs && params && layout_h && value && workspace, AURPPO_EINVAL, "%s: null pointer", who);
    AURPPO_REQUIRE(!noise || (actions && logp), AURPPO_EINVAL, "%s: sampling needs actions and logp outputs", who);
    int rc = check_shape(D, A, continuous, hidden, num_layers, who);
    if (rc != AURPPO_OK) return rc;
    AURPPO_REQUIRE(N > 0 && n_params > 0, AURPPO_ESHAPE, "%s: N=%d n_params=%d", who, N, n_params);
    AURPPO_REQUIRE(aligned_to(workspace, 64), AURPPO_EINVAL, "%s: workspace not 64-byte aligned", who);
    WideArgs a = {};
    a.obs = obs; a.noise = noise; a.params = params; a.out_actions = actions; a.out_logp = logp; a.out_value = value;
    a.N = N; a.D = D; a.A = A; a.Hd = hidden; a.continuous = continuous ? 1 : 0;
    a.net_base = noise ? 0 : 1;
    a.net_count = noise ? 2 : 1;
    rc = fill_layout(a.L, layout_h, num_layers, continuous, n_params, who);
    if (rc != AURPPO_OK) return rc;
    const WideWs wv = wide_ws(workspace, 0, 0);      // the operand copies sit in front of the slabs
    a.wop = wv.wop;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_mlpw_prep, dim3(96), dim3(256), 0, s, params, a.L, num_layers, D, hidden, wv.wop,
                       (const float4*)nullptr, 0, (const int32_t*)nullptr, 0, (double (*)[2]) nullptr, 0, (unsigned*)nullptr);
    AURPPO_LAUNCH_CHECK("k_mlpw_prep");
    const int grid = ((N + R - 1) / R) * a.net_count;
    static bool attr[kMaxDevices][MAXL] = {};
    bool* ad = &attr[aurppo_device_slot()][num_layers - 1];
    switch (num_layers) {
        case 1: rc = launch_wide(k_mlpw_act<1>, ad, grid, wide_lds_bytes<false, 1>(1), s, a); break;
        case 2: rc = launch_wide(k_mlpw_act<2>, ad, grid, wide_lds_bytes<false, 2>(1), s, a); break;
        default: rc = launch_wide(k_mlpw_act<3>, ad, grid, wide_lds_bytes<false, 3>(1), s, a); break;
    }
    if (rc != AURPPO_OK) return rc;
    AURPPO_LAUNCH_CHECK("k_mlpw_act");
    return AURPPO_OK;
}

#ifdef K7W_STAMPS
extern "C" int aurppo_k7w_stamps_read(unsigned long long* host_out) {      // (kMaxSlabs, 32); diagnostic build only
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_k7w_stamps), sizeof(unsigned long long) * kMaxSlabs * 32) == hipSuccess ? 0 : -1;
}
#endif
