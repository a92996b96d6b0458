// K7, transposed arrangement on the bf16 matrix pipe (k_mlp_step4): the fused PPO minibatch step of k_mlp_step2 / k_mlp_step3
// (src/ppo.py:219-267 over src/models/actor_critic.py:8-51) with ONE WAVE carrying a whole net for its 32 samples and no
// barrier anywhere in the tile loop.
//
// k_mlp_step3's stamps (profiles/r03/k7_stamps_v3.txt) showed what bounds the two-tile-set structure once the matrix work is
// 2.7x cheaper: a tile is a chain of eight phases, each "LDS reads -> MFMA chain -> VALU epilogue -> LDS writes -> barrier",
// and with two waves per SIMD little of that chain overlaps anything (32 k cycles per tile for 4.4 k of matrix work).
// Here the products are formed transposed, Z^T[o][s] = W . X^T:
//   * the sample is on the lane of every accumulator, the feature index in its registers, so tanh / loss / dZ are per-lane
//     code, and an accumulator block converted to bf16 planes in place IS the next layer's B operand (bf16x3.h, kappa order)
//     -- forward and data-gradient passes never touch LDS;
//   * the actor's and the critic's terms of the loss are separable (policy + entropy terms read the actor only, the value
//     term the critic only; the advantage statistics are precomputed), so a wave needs nothing from any other wave: four
//     independent waves per workgroup, one per SIMD with the 512-register budget (two actor, two critic), each with its own
//     tile stream from its net's counter;
//   * LDS holds only what the weight gradients need: wave-private [sample][feature] images of X, H1 -> dZ1, H2 -> dZ2 and
//     dOut (39 KB per wave), both operands of dW = dZ^T . H read across their rows with ds_read_b64_tr_b16;
//   * every weight is an A operand streamed from L2 in operand order (90 KB per net, bf16 planes, kept current by the
//     optimizer kernel); the full weight-gradient accumulators of the net (dW1 64 + dW2 64 + dW3 16 registers) and the bias
//     gradients' per-lane partial sums persist in registers across tiles.
#include <stdlib.h>

#pragma clang fp contract(fast)
#include "bf16x3.h"
#include "mlp_common.h"

using namespace aurppo_mlp;
using namespace bf3;

// Diagnostic build only (tools/mlp_stamps.py): every wave accumulates the shader cycles of each stage of its tiles in
// scalar registers; wave 0 of each workgroup dumps them to the workspace at the end.
#ifdef AURPPO_MLP_STAMPS
#define STAMP4(k)                                                         \
    do {                                                                  \
        const unsigned long long t__ = __builtin_readcyclecounter();      \
        st_acc[k] += t__ - st_last;                                       \
        st_last = t__;                                                    \
    } while (0)
#else
#define STAMP4(k) do { } while (0)
#endif

namespace {

constexpr int kThreads4 = 256;
// dynamic LDS (bytes): per wave X image, H1 image, H2 image (X layout, 3 planes each), dOut image [s 32][a 16] (32-B rows)
constexpr int kDoRow4 = 32, kDoPlane4 = R * kDoRow4;
constexpr int wX = 0, wH1 = wX + 3 * kXPlane, wH2 = wH1 + 3 * kXPlane, wDo = wH2 + 3 * kXPlane;
constexpr int kWaveBytes = wDo + 3 * kDoPlane4;                       // 39 936
// shared tables behind the four wave regions: biases in accumulator order [net][layer 2][blk 2][h 2][16], head bias / log-std
constexpr int oTab = 4 * kWaveBytes;
constexpr int oBt = oTab;                                             // float [2][2][2][2][16]
constexpr int oB3t = oBt + 4 * 2 * 2 * 2 * 2 * 16;                    // float [2 nets][2 h][8]   head bias in accumulator order
constexpr int oLst = oB3t + 4 * 2 * 2 * 8;                            // float [2 h][8] log-std, [2 h][8] 1 / sigma^2
constexpr int kDyn4 = oLst + 4 * 2 * 2 * 8;
static_assert(kWaveBytes % 16 == 0 && oTab % 16 == 0, "alignment");
constexpr int kPersist = 64 + 64 + 16 + 32 + 32 + 8 + 8;              // registers parked at the end: dW1, dW2, dW3, db1, db2, db3, dls
static_assert(2 * kPersist * kWave * 4 <= oTab, "hand-over scratch must fit the dead images");

// block `id` of this wave's net: this lane's 16 bytes of each plane.  `wnet` is wave-uniform (scalar registers), `lane16` the
// lane's byte offset: one scalar base per block + a 32-bit vector offset + an immediate per plane, instead of ninety 64-bit
// vector addresses that would be formed once, hoisted out of the tile loop and spilled.
__device__ __forceinline__ Frag3 w_frag(const char* wnet, int id, int lane16) {
    const char* blk = wnet + id * (3 * 1024);
    Frag3 f;
    f.p[0] = *reinterpret_cast<const bf16x8*>(blk + lane16);
    f.p[1] = *reinterpret_cast<const bf16x8*>(blk + 1024 + lane16);
    f.p[2] = *reinterpret_cast<const bf16x8*>(blk + 2048 + lane16);
    return f;
}

// Epilogues of a 32-row accumulator block (my sample's 16 values of it), eight values = one B fragment at a time.
// tanh(acc + bias) -> the two fragments
__device__ __forceinline__ void tanh_frags(const f32x16& acc, const float* bt, Frag3& f0, Frag3& f1) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tanh_fast(acc[e] + bt[e]);
    f0 = regs_to_frag(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tanh_fast(acc[8 + e] + bt[8 + e]);
    f1 = regs_to_frag(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}
// dZ = dH (1 - h^2) with h read back from block b of its image (exact): the two B fragments of dZ (kappa order)
__device__ __forceinline__ void dz_frags(const f32x16& dh, const char* img, int s, int h, int b, Frag3& f0, Frag3& f1) {
    u32x2 q[4][3];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int o = xoff(s, 32 * b + 8 * gq + 4 * h);
#pragma unroll
        for (int p = 0; p < 3; ++p) q[gq][p] = *reinterpret_cast<const u32x2*>(img + p * kXPlane + o);
    }
    unsigned w[3][8];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const float h0 = join_lo(q[gq][0].x, q[gq][1].x, q[gq][2].x), h1 = join_hi(q[gq][0].x, q[gq][1].x, q[gq][2].x);
        const float h2 = join_lo(q[gq][0].y, q[gq][1].y, q[gq][2].y), h3 = join_hi(q[gq][0].y, q[gq][1].y, q[gq][2].y);
        const float d0 = dh[4 * gq + 0] * (1.0f - h0 * h0), d1 = dh[4 * gq + 1] * (1.0f - h1 * h1);
        const float d2 = dh[4 * gq + 2] * (1.0f - h2 * h2), d3 = dh[4 * gq + 3] * (1.0f - h3 * h3);
        split3(d0, d1, w[0][2 * gq], w[1][2 * gq], w[2][2 * gq]);
        split3(d2, d3, w[0][2 * gq + 1], w[1][2 * gq + 1], w[2][2 * gq + 1]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        f0.p[p] = as_frag(w[p][0], w[p][1], w[p][2], w[p][3]);
        f1.p[p] = as_frag(w[p][4], w[p][5], w[p][6], w[p][7]);
    }
}
// acc += A . 1: the row sums of A (all 32 columns of the result are equal); bf16 1.0 = 0x3F80, planes 1 and 2 of "1" are zero
__device__ __forceinline__ f32x16 mma_ones(const Frag3& a, f32x16 c) {
    const bf16x8 one = as_frag(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], one, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], one, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], one, c, 0, 0, 0);
    return c;
}

// FAST: state rows are 16-byte aligned multiples of four floats AND the records are packed (actions == nullptr): every row
// fetch is a 16-byte load at (scalar base + 32-bit offset).  Otherwise the generic element-wise fetch.
template <bool FAST>
__global__ __launch_bounds__(kThreads4, 1) void k_mlp_step4(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ double s_red[2][kThreads4 / kWave];
    __shared__ double s_lp[4][6];
    __shared__ float s_mean, s_std;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int net = w >> 1, u = w & 1;            // waves 0,1: actor; 2,3: critic
    const int s0 = lane & 31, h0 = lane >> 5;     // my sample of the tile; my half of every accumulator's rows
    const int D = a.D, A = a.A;
    const int AW = a.continuous ? a.A : 1;

    char* const my = lds + w * kWaveBytes;
    char* const sX = my + wX;
    char* const sH1 = my + wH1;                   // H1, later dZ1
    char* const sH2 = my + wH2;                   // H2, later dZ2
    char* const sDo = my + wDo;
    float* const sBt = reinterpret_cast<float*>(lds + oBt);
    float* const sB3t = reinterpret_cast<float*>(lds + oB3t);
    float* const sLst = reinterpret_cast<float*>(lds + oLst);

    // ---- shared tables (accumulator order: register e of lane half h is row (e & 3) + 8 (e >> 2) + 4 h of its block)
    for (int e = tid; e < 2 * 2 * 2 * 2 * 16; e += kThreads4) {
        const int r = e & 15, hh = (e >> 4) & 1, blk = (e >> 5) & 1, layer = (e >> 6) & 1, n = e >> 7;
        const int o = 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * hh;
        sBt[e] = a.params[(layer ? a.L.b2[n] : a.L.b1[n]) + o];
    }
    if (tid < 32) {
        const int r = tid & 7, hh = (tid >> 3) & 1, n = tid >> 4;
        const int k = (r & 3) + 8 * (r >> 2) + 4 * hh;
        sB3t[tid] = k < (n ? 1 : A) ? a.params[a.L.b3[n] + k] : 0.0f;
        if (n == 0) {
            const float ls = (a.continuous && k < A) ? a.params[a.L.logstd + k] : 0.0f;
            const float sd = expf(ls);
            sLst[tid] = ls;
            sLst[16 + tid] = 1.0f / (sd * sd);
        }
    }
    // my images start out zero: X columns >= D, dOut rows of padding actions
    {
        u32x4* z = reinterpret_cast<u32x4*>(my);
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (int e = lane; e < kWaveBytes / 16; e += kWave) z[e] = zero;
    }
    // ---- minibatch advantage statistics from the partials (same order in every workgroup)
    {
        double sm = 0.0, q = 0.0;
        for (int b = tid; b < a.n_stat_blocks; b += kThreads4) {
            sm += a.stats[2 * b];
            q += a.stats[2 * b + 1];
        }
        const double ts = block_sum<kThreads4 / kWave>(sm, s_red[0]);
        const double tq = block_sum<kThreads4 / kWave>(q, s_red[1]);
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
    float ent_gauss = 0.0f;
    if (a.continuous)
        for (int k = 0; k < A; ++k) ent_gauss += (0.5f + 0.9189385332046727f) + a.params[a.L.logstd + k];

    // ---- persistent accumulators
    f32x16 gW1[2][2], gW2[2][2];                  // [out block][in block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) gW1[i][j] = gW2[i][j] = zero16();
    f32x4 gW3[4];                                 // rows = head outputs, columns 16 blk + (lane & 15)
#pragma unroll
    for (int i = 0; i < 4; ++i) gW3[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 gB1[2] = {zero16(), zero16()}, gB2[2] = {zero16(), zero16()};   // bias gradients as products with a column of ones (every column equal)
    float gb3[8], gls[8];                          // head bias / log-std: per-lane partial sums over this lane's samples, accumulator order
#pragma unroll
    for (int e = 0; e < 8; ++e) gb3[e] = gls[e] = 0.0f;
    double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0, l4 = 0.0;    // actor: pg, ent, okl, kl, cf (lane half 0 counts); critic: vl in l0

    const char* const wnet0 = reinterpret_cast<const char*>(a.wop3) + (size_t)net * kW4Blocks * 3 * 1024;
    const int n_tiles = (a.h.M + R - 1) / R;
    unsigned* const ctr = a.tile_counter + net;
    const bool stat = a.static_tiles != 0;
    const int n_w = 2 * (int)gridDim.x, my_w = 2 * (int)blockIdx.x + u;     // waves that share this net's tiles
    const bool packed = a.actions == nullptr;
    // (FAST runs every k-step and both state blocks: beyond D the image and W1's copy are zero -- no branches in the chains)
    const int nks1 = FAST ? 4 : (D + 15) >> 4, ndb = FAST ? 2 : (D + 31) >> 5;

    // ---- the tile queue of this wave: cur, nxt known; the grab for the one after is in flight
    int zero_off = 0;
    asm volatile("" : "+v"(zero_off));
    // Small minibatches (under four tiles per wave): every wave's first tile is fixed (wave q of Q takes tile q) and the counter
    // starts behind them, because consecutive grabs land in the first waves to arrive.
    const bool fixed_start = !stat && n_tiles < 4 * n_w;
    const int dyn_base = fixed_start ? n_w : 0;
    int t_cur, t_nxt, t_raw = 0;
    if (stat) {
        t_cur = my_w;
        t_nxt = my_w + n_w;
    } else if (fixed_start) {
        int g1 = 0;
        if (lane == 0) g1 = (int)atomicAdd(ctr + zero_off, 1u);
        t_cur = my_w;
        t_nxt = dyn_base + __builtin_amdgcn_readfirstlane(g1);
    } else {
        int g2 = 0;
        if (lane == 0) g2 = (int)atomicAdd(ctr + zero_off, 2u);
        t_cur = __builtin_amdgcn_readfirstlane(g2);
        t_nxt = t_cur + 1;
    }
    auto load_idx = [&](int tile) -> int {
        const int m = tile * R + s0;
        const bool ok = tile < n_tiles && m < a.h.M;
        const int v = a.idx[ok ? m : 0];
        return ok ? v : -1;
    };
    // rows of a tile, in flight: the observation as this lane's B-operand elements (natural k order: 16 ks + 8 h + j),
    // the record, the action elements of this lane's head rows
    float xr[32];
    float4 rc = make_float4(0.f, 0.f, 0.f, 0.f);
    float act[8];
    auto fetch = [&](int src, int hh) {     // hh = hh (an opaque copy inside the tile loop: nothing of this is to be hoisted)
        if (FAST) {
            // scalar bases (kernel arguments) + 32-bit byte offsets: no 64-bit vector addresses to keep (and spill) across the loop
            const unsigned row = (unsigned)(src >= 0 ? src : 0);
            const char* const ob = reinterpret_cast<const char*>(a.obs);
            const unsigned xo = row * (unsigned)(4 * D) + 32u * (unsigned)hh;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int c = 16 * ks + 8 * hh + 4 * q;
                    const float4 v = *reinterpret_cast<const float4*>(ob + (c < D ? xo + 64u * ks + 16u * q : 0u));
                    xr[8 * ks + 4 * q + 0] = v.x; xr[8 * ks + 4 * q + 1] = v.y; xr[8 * ks + 4 * q + 2] = v.z; xr[8 * ks + 4 * q + 3] = v.w;
                }
            // the 64-byte record: {old_logp, adv, ret, old_v} + the action row: dims {0-3, 8-11} for lane half 0, {4-7} for half 1
            const char* const rb = reinterpret_cast<const char*>(a.rec);
            const unsigned ro = row * 64u;
            rc = *reinterpret_cast<const float4*>(rb + ro);
            if (net == 0) {
                const float4 p = *reinterpret_cast<const float4*>(rb + ro + 16u + 16u * (unsigned)hh);
                const float4 q = *reinterpret_cast<const float4*>(rb + ro + 48u);
                // raw values only (touching a loaded value here would wait for it -- and for every older load -- on the spot): lane
                // half 1's elements 4..7 stand for dims 12..15, which a packed record does not have and the loss masks (k >= A)
                act[0] = p.x; act[1] = p.y; act[2] = p.z; act[3] = p.w;
                act[4] = q.x; act[5] = q.y; act[6] = q.z; act[7] = q.w;
            }
            return;
        }
        const size_t row = (size_t)(src >= 0 ? src : 0);
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int c = 16 * (e >> 3) + 8 * hh + (e & 7);
            xr[e] = a.obs[c < D ? row * D + c : (size_t)0];
        }
        rc = a.rec[row * a.rec_stride];
        if (net == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = (e & 3) + 8 * (e >> 2) + 4 * hh;
                act[e] = packed ? (k < 12 && k < AW ? reinterpret_cast<const float*>(a.rec)[row * 16 + 4 + k] : 0.0f)
                                : a.actions[k < AW ? row * AW + k : (size_t)0];
            }
        }
    };
    int src_cur = load_idx(t_cur);
    fetch(src_cur, h0);
    Frag3 wa[4], wb[4], wb2[4];                    // the weight stages (see the tile loop)
    const char* wnet = wnet0;
    int lane16 = lane * 16;
    auto load4 = [&](Frag3 (&dst)[4], int id0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = w_frag(wnet, id0 + q, lane16);
    };
    auto load2 = [&](Frag3 (&dst)[4], int id0) {
        dst[0] = w_frag(wnet, id0, lane16);
        dst[1] = w_frag(wnet, id0 + 1, lane16);
    };
    load4(wa, kW4_W1 + 0);
#ifdef AURPPO_MLP_STAMPS
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
    const unsigned long long clk0 = st_last, rt0 = wall_clock64();
    unsigned long long n_my_tiles = 0;
#endif

    while (t_cur < n_tiles) {
        // the tile after next: asked for now, read at the end of this tile
        if (!stat && lane == 0) t_raw = (int)atomicAdd(ctr + zero_off, 1u);
        const int src_nxt = load_idx(t_nxt);
        const bool real = src_cur >= 0;
        // opaque per-tile copies of the lane coordinates: every LDS address below is re-derived from them inside the tile
        // instead of being hoisted out of the loop as dozens of loop-invariant address registers that are then spilled
        int s = s0, h = h0, lane_t = lane;
        asm volatile("" : "+v"(s), "+v"(h), "+v"(lane_t));
        {   // ... and of the weight base (scalar) and the lane's offset into a block
            int wz = 0;                               // (an opaque zero added to the pointer keeps it a GLOBAL pointer)
            asm volatile("" : "+s"(wz));
            wnet = wnet0 + wz;
            lane16 = lane_t * 16;
        }

        // Weight stages.  Every weight fragment is an L2 load; left to itself the scheduler hoists all 90 of a tile's to its
        // top (they depend on nothing) and spills.  So the loads are issued by hand, one stage (4 k-steps of one block, 48
        // registers) ahead of the chain that consumes them, into two buffers that alternate, with scheduling fences between
        // the stages; inside a stage the order is "issue the next stage's loads, run this block's MFMA chain, finish the
        // PREVIOUS block's epilogue under it".
#define SB() __builtin_amdgcn_sched_barrier(0)
        // Interleave hint for a region that holds an MFMA chain and an independent epilogue: n x {1 MFMA, nv VALU, nt TRANS}.  One
        // wave per SIMD issues in order: vector work only overlaps the matrix pipe if it sits BETWEEN the MFMAs in program order,
        // and left alone the scheduler emits the chain back to back and the epilogue behind it (profiles/r03: 30.7 k cycles per tile
        // = the sum of the two).
#define IL(n, nv, nt)                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
            __builtin_amdgcn_sched_group_barrier(0x002, nv, 0);    \
            __builtin_amdgcn_sched_group_barrier(0x400, nt, 0);    \
        }
        // ---- X: this lane's 32 observation elements -> B fragments (4 k-steps) + the [s][d] image dW1 reads
        Frag3 xb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (real && 16 * ks + 8 * h + j < D) ? xr[8 * ks + j] : 0.0f;
            xb[ks] = regs_to_frag(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            if (ks < nks1) {
                const int o = xoff(s, 16 * ks + 8 * h);
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(sX + p * kXPlane + o) = xb[ks].p[p];
            }
        }
        const float4 rec = rc;
        float av[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = act[e];
        // ---- F1: Z1^T = W1 X^T (wa holds W1 block 0 since the end of the previous tile); k-step ks of the first block's chain only
        // needs X's k-step ks: the chain runs under the rest of the split
        Frag3 hb[4];                               // B fragments of H1, k-steps 0..3 (kappa order)
        f32x16 acc0 = zero16(), acc1 = zero16();
        load4(wb, kW4_W1 + 4);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            if (ks < nks1) acc0 = mma32x3(wa[ks], xb[ks], acc0);
        IL(24, 9, 0)
        SB();
        STAMP4(0);
        load4(wa, kW4_W2 + 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            if (ks < nks1) acc1 = mma32x3(wb[ks], xb[ks], acc1);
        tanh_frags(acc0, sBt + (((net * 2 + 0) * 2 + 0) * 2 + h) * 16, hb[0], hb[1]);
        store_frags_x(sH1, s, h, 0, hb[0], hb[1]);
        IL(24, 8, 2)
        SB();
        STAMP4(1);
        // ---- F2 / F3, software-pipelined across the blocks: a chain's k-steps run as soon as the fragments they need exist, the
        // tanh + split epilogue of one block under the MFMAs of the next
        Frag3 h2b[4];
        load4(wb, kW4_W2 + 4);
        acc0 = zero16();
        acc0 = mma32x3(wa[0], hb[0], acc0);
        acc0 = mma32x3(wa[1], hb[1], acc0);
        tanh_frags(acc1, sBt + (((net * 2 + 0) * 2 + 1) * 2 + h) * 16, hb[2], hb[3]);
        store_frags_x(sH1, s, h, 1, hb[2], hb[3]);
        f32x16 acc2 = zero16();
        acc2 = mma32x3(wb[0], hb[0], acc2);
        acc2 = mma32x3(wb[1], hb[1], acc2);
        IL(24, 8, 2)
        SB();
        STAMP4(2);
        acc0 = mma32x3(wa[2], hb[2], acc0);
        acc0 = mma32x3(wa[3], hb[3], acc0);
        load4(wa, kW4_W3);
        acc2 = mma32x3(wb[2], hb[2], acc2);
        acc2 = mma32x3(wb[3], hb[3], acc2);
        tanh_frags(acc0, sBt + (((net * 2 + 1) * 2 + 0) * 2 + h) * 16, h2b[0], h2b[1]);
        store_frags_x(sH2, s, h, 0, h2b[0], h2b[1]);
        IL(12, 0, 0)
        IL(12, 16, 3)
        SB();
        // ---- F3: head (W3 zero-padded to 32 rows)
        load2(wb, kW4_W3T);
        acc1 = zero16();
        acc1 = mma32x3(wa[0], h2b[0], acc1);
        acc1 = mma32x3(wa[1], h2b[1], acc1);
        tanh_frags(acc2, sBt + (((net * 2 + 1) * 2 + 1) * 2 + h) * 16, h2b[2], h2b[3]);
        store_frags_x(sH2, s, h, 1, h2b[2], h2b[3]);
        IL(12, 16, 3)
        SB();
        acc1 = mma32x3(wa[2], h2b[2], acc1);
        acc1 = mma32x3(wa[3], h2b[3], acc1);
        float outv[8];
        {
            const float* b3 = sB3t + (net * 2 + h) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) outv[e] = acc1[e] + b3[e];
        }
        SB();
        STAMP4(3);
        load4(wa, kW4_W2T + 0);                    // for dH1's first block, far ahead: the loss and dW3 / dW2 run meanwhile
        // ---- loss: this net's terms for my sample; head outputs become their gradients (per-lane code: the sample is the lane)
        float dv[8];
        if (net == 1) {
            // critic: the value is row 0 = register 0 of lane half 0
            const float v_new = __shfl(outv[0], s, kWave);
            const PpoSample t = ppo_sample(rec.x, rec.x, rec.y, v_new, rec.w, rec.z, mean, denom, invM, a.h);
            const float gv = (real && h == 0) ? t.g_v : 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) dv[e] = e == 0 ? gv : 0.0f;
            gb3[0] += gv;
            if (real && h == 0) l0 += (double)t.vl;
        } else if (a.continuous) {
            const float* ls = sLst + h * 8;
            const float* iv = sLst + 16 + h * 8;
            float logp = 0.0f, z[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = (e & 3) + 8 * (e >> 2) + 4 * h;
                z[e] = av[e] - outv[e];
                if (k < A) logp += (-(z[e] * z[e]) * (0.5f * iv[e]) - ls[e]) - 0.9189385332046727f;
            }
            logp += __shfl_xor(logp, 32, kWave);
            const PpoSample t = ppo_sample(logp, rec.x, rec.y, rec.w, rec.w, rec.z, mean, denom, invM, a.h);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = (e & 3) + 8 * (e >> 2) + 4 * h;
                const bool on = real && k < A;
                dv[e] = on ? t.g_logp * (z[e] * iv[e]) : 0.0f;
                gb3[e] += dv[e];
                if (on) gls[e] += t.g_logp * (z[e] * z[e] * iv[e] - 1.0f) + g_ent;
            }
            if (real && h == 0) {
                l0 += (double)t.pg; l1 += (double)ent_gauss; l2 += (double)t.okl; l3 += (double)t.kl; l4 += (double)t.cf;
            }
        } else {
            // Categorical(logits): log-softmax over the 16 (A) logits held by this lane and lane ^ 32
            const int ai = (int)__shfl(av[0], s, kWave);          // the action index sits in dim 0 = element 0 of half 0
            float mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = (e & 3) + 8 * (e >> 2) + 4 * h;
                if (k < A) mx = fmaxf(mx, outv[e]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, kWave));
            float se = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = (e & 3) + 8 * (e >> 2) + 4 * h;
                if (k < A) se += expf(outv[e] - mx);
            }
            se += __shfl_xor(se, 32, kWave);
            const float lse = mx + logf(se);
            float lp[8], pr[8], ent = 0.0f, logp = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = (e & 3) + 8 * (e >> 2) + 4 * h;
                lp[e] = k < A ? outv[e] - lse : 0.0f;
                pr[e] = k < A ? expf(lp[e]) : 0.0f;
                ent -= pr[e] * lp[e];
                if (k == ai) logp += lp[e];
            }
            ent += __shfl_xor(ent, 32, kWave);
            logp += __shfl_xor(logp, 32, kWave);
            const PpoSample t = ppo_sample(logp, rec.x, rec.y, rec.w, rec.w, rec.z, mean, denom, invM, a.h);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = (e & 3) + 8 * (e >> 2) + 4 * h;
                dv[e] = (real && k < A) ? t.g_logp * ((k == ai ? 1.0f : 0.0f) - pr[e]) + g_ent * (-pr[e] * (lp[e] + ent)) : 0.0f;
                gb3[e] += dv[e];
            }
            if (real && h == 0) {
                l0 += (double)t.pg; l1 += (double)ent; l2 += (double)t.okl; l3 += (double)t.kl; l4 += (double)t.cf;
            }
        }
        // dOut: B fragment of dH2's single k-step (rows 0..15 of the block = registers 0..7) + the [s][a] image dW3 reads
        const Frag3 dob = regs_to_frag(dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], dv[6], dv[7]);
#pragma unroll
        for (int gq = 0; gq < 2; ++gq)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const u32x4v q = __builtin_bit_cast(u32x4v, dob.p[p]);
                *reinterpret_cast<u32x2*>(sDo + p * kDoPlane4 + s * kDoRow4 + 2 * (8 * gq + 4 * h)) = gq ? u32x2{q.z, q.w} : u32x2{q.x, q.y};
            }
        SB();
        STAMP4(4);
        // ---- dH2^T = W3^T dOut^T (K = 16), then dW3 += dOut^T H2 (16x16x32, one k-step over the 32 samples, both operands read
        // across their images' rows) with the dZ2 epilogues under its chains
        Frag3 dzb[4];
        load4(wb2, kW4_W2T + 4);
        acc0 = zero16();
        acc1 = zero16();
        acc0 = mma32x3(wb[0], dob, acc0);
        acc1 = mma32x3(wb[1], dob, acc1);
        {
            const Frag3 da = plain_cols16(sDo, kDoRow4, kDoPlane4, 0, lane_t);
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) gW3[blk] = mma16x3(da, x_cols16(sH2, 16 * blk, lane_t), gW3[blk]);
        }
        dz_frags(acc0, sH2, s, h, 0, dzb[0], dzb[1]);
        dz_frags(acc1, sH2, s, h, 1, dzb[2], dzb[3]);
        IL(12, 0, 0)
        IL(24, 12, 0)
        SB();       // (dW3 has read the H2 image: dZ2 may now take its place)
        store_frags_x(sH2, s, h, 0, dzb[0], dzb[1]);
        store_frags_x(sH2, s, h, 1, dzb[2], dzb[3]);
        SB();
        STAMP4(5);
        // ---- dH1^T = W2^T dZ2^T (wa = W2^T block 0, loaded before the loss; wb2 = block 1, loaded before dH2), dZ1 kept as
        // fragments: its image takes H1's place only after dW2 has read H1
        acc0 = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc0 = mma32x3(wa[ks], dzb[ks], acc0);
        acc1 = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc1 = mma32x3(wb2[ks], dzb[ks], acc1);
        Frag3 d1b[4];
        dz_frags(acc0, sH1, s, h, 0, d1b[0], d1b[1]);
        IL(24, 0, 0)
        IL(24, 8, 0)
        SB();
        STAMP4(6);
        // ---- The long-latency loads of the tile go out HERE: the next tile's first weight stage, then its rows from HBM.  Returns
        // are counted in issue order, so nothing issued after the row fetch can be used before it has come back: what follows
        // until the next tile's X stage is LDS-and-matrix work only (dW2, dZ1's image, dW1: ~100 MFMAs).
        load4(wa, kW4_W1 + 0);
        fetch(src_nxt, h);
        SB();
        // ---- dW2 += dZ2^T H1;  db2 += dZ2^T 1 (the bias gradient as one more column of the product: accumulators stay in the
        // matrix pipe's registers, no vector adds); dH1's second block gets its epilogue under these chains
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const Frag3 b0 = x_cols(sH1, ks, 0, lane_t), b1 = x_cols(sH1, ks, 32, lane_t);
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const Frag3 za = x_cols(sH2, ks, 32 * ob, lane_t);
                gW2[ob][0] = mma32x3(za, b0, gW2[ob][0]);
                gW2[ob][1] = mma32x3(za, b1, gW2[ob][1]);
                gB2[ob] = mma_ones(za, gB2[ob]);
            }
        }
        dz_frags(acc1, sH1, s, h, 1, d1b[2], d1b[3]);
        IL(60, 4, 0)
        SB();       // (dW2 has read the H1 image: dZ1 may now take its place)
        store_frags_x(sH1, s, h, 0, d1b[0], d1b[1]);
        store_frags_x(sH1, s, h, 1, d1b[2], d1b[3]);
        SB();
        STAMP4(7);
        // ---- dW1 += dZ1^T X;  db1 += dZ1^T 1
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const Frag3 z0 = x_cols(sH1, ks, 0, lane_t), z1 = x_cols(sH1, ks, 32, lane_t);
            gB1[0] = mma_ones(z0, gB1[0]);
            gB1[1] = mma_ones(z1, gB1[1]);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                if (db < ndb) {
                    const Frag3 xbq = x_cols(sX, ks, 32 * db, lane_t);
                    gW1[0][db] = mma32x3(z0, xbq, gW1[0][db]);
                    gW1[1][db] = mma32x3(z1, xbq, gW1[1][db]);
                }
            }
        }
        SB();
        STAMP4(8);
#ifdef AURPPO_MLP_STAMPS
        ++n_my_tiles;
#endif
#undef SB
#undef IL
        // ---- advance the queue
        t_cur = t_nxt;
        src_cur = src_nxt;
        t_nxt = stat ? t_nxt + n_w : dyn_base + __builtin_amdgcn_readfirstlane(t_raw);
    }
#ifdef AURPPO_MLP_STAMPS
    if (w == 0 && lane == 0) {
        for (int k = 0; k < 12; ++k) a.stamps[(size_t)blockIdx.x * 40 + k] = st_acc[k];
        a.stamps[(size_t)blockIdx.x * 40 + 12] = n_my_tiles;
        a.stamps[(size_t)blockIdx.x * 40 + 32] = __builtin_readcyclecounter() - clk0;
        a.stamps[(size_t)blockIdx.x * 40 + 33] = wall_clock64() - rt0;
    }
#endif
    __syncthreads();      // every wave's images are dead: the hand-over below reuses them

    // ---- bias / log-std gradients: per-lane partials -> sums over the 32 lanes of each half
    auto half_sum = [&](float x) {
        x += __shfl_xor(x, 1, kWave);
        x += __shfl_xor(x, 2, kWave);
        x += __shfl_xor(x, 4, kWave);
        x += __shfl_xor(x, 8, kWave);
        x += __shfl_xor(x, 16, kWave);
        return x;
    };
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        gb3[e] = half_sum(gb3[e]);
        gls[e] = half_sum(gls[e]);
    }
    // loss partials of this wave
    {
        double v5[5] = {l0, l1, l2, l3, l4};
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const double x = wave_sum(v5[q]);
            if (lane == 0) s_lp[w][q] = x;
        }
    }
    // ---- hand-over: the second wave of each net parks its accumulators, the first adds them and writes the net's half of the slab
    float* park = reinterpret_cast<float*>(lds) + (size_t)net * kPersist * kWave + lane;
    if (u == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    park[(16 * (2 * i + j) + e) * kWave] = gW1[i][j][e];
                    park[(64 + 16 * (2 * i + j) + e) * kWave] = gW2[i][j][e];
                }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) park[(128 + 4 * i + e) * kWave] = gW3[i][e];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                park[(144 + 16 * b + e) * kWave] = gB1[b][e];
                park[(176 + 16 * b + e) * kWave] = gB2[b][e];
            }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            park[(208 + e) * kWave] = gb3[e];
            park[(216 + e) * kWave] = gls[e];
        }
    }
    __syncthreads();
    if (u == 0) {
        float* slab = a.slabs + (size_t)blockIdx.x * a.L.n_params;
        const int col = lane & 31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = 16 * (2 * i + j);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int o = 32 * i + acc_row(e, lane), c = 32 * j + col;
                    if (c < D) slab[a.L.w1[net] + o * D + c] = gW1[i][j][e] + park[(r + e) * kWave];
                    slab[a.L.w2[net] + o * H + c] = gW2[i][j][e] + park[(64 + r + e) * kWave];
                }
            }
        const int od = net ? 1 : A;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = 4 * (lane >> 4) + e, c = 16 * i + (lane & 15);
                if (o < od) slab[a.L.w3[net] + o * H + c] = gW3[i][e] + park[(128 + 4 * i + e) * kWave];
            }
        if (col == 0) {      // lanes 0 and 32: the sums of their half's rows
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int o = 32 * b + (e & 3) + 8 * (e >> 2) + 4 * h0;
                    slab[a.L.b1[net] + o] = gB1[b][e] + park[(144 + 16 * b + e) * kWave];
                    slab[a.L.b2[net] + o] = gB2[b][e] + park[(176 + 16 * b + e) * kWave];
                }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = (e & 3) + 8 * (e >> 2) + 4 * h0;
                if (k < od) slab[a.L.b3[net] + k] = gb3[e] + park[(208 + e) * kWave];
                if (net == 0 && a.continuous && k < A) slab[a.L.logstd + k] = gls[e] + park[(216 + e) * kWave];
            }
        }
        if (lane == 0) {
            double* lp = a.loss_part + (size_t)blockIdx.x * 8;   // {pg, vl, ent, okl, kl, cf, mean, std}
            if (net == 0) {
                lp[0] = s_lp[0][0] + s_lp[1][0];
                lp[2] = s_lp[0][1] + s_lp[1][1];
                lp[3] = s_lp[0][2] + s_lp[1][2];
                lp[4] = s_lp[0][3] + s_lp[1][3];
                lp[5] = s_lp[0][4] + s_lp[1][4];
            } else {
                lp[1] = s_lp[2][0] + s_lp[3][0];
                lp[6] = (double)mean;
                lp[7] = (double)s_std;
            }
        }
    }
}

// wop4[...] = the bf16 planes of every weight of both nets in A-operand order (bf16x3.h), written destination-first so that
// padding (state columns >= D, head rows >= the head's width) is zero without a clearing pass
__global__ __launch_bounds__(256) void k_mlp4_prep(const float* __restrict__ params, MlpLayout L, int D, int A, unsigned short* __restrict__ wop4) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < 2 * kW4Blocks * 64 * 8; e += gridDim.x * 256) {
        const int j = e & 7, lane = (e >> 3) & 63, id = (e >> 9) % kW4Blocks, net = (e >> 9) / kW4Blocks;
        const int r = lane & 31, h = lane >> 5;
        const int od = net ? 1 : A;
        float v;
        if (id < kW4_W2) {                    // W1, natural k
            const int ob = (id - kW4_W1) >> 2, ks = (id - kW4_W1) & 3, k = 16 * ks + 8 * h + j;
            v = k < D ? params[L.w1[net] + (32 * ob + r) * D + k] : 0.0f;
        } else if (id < kW4_W3) {
            const int ob = (id - kW4_W2) >> 2, ks = (id - kW4_W2) & 3;
            v = params[L.w2[net] + (32 * ob + r) * H + kappa(ks, h, j)];
        } else if (id < kW4_W3T) {
            const int ks = id - kW4_W3;
            v = r < od ? params[L.w3[net] + r * H + kappa(ks, h, j)] : 0.0f;
        } else if (id < kW4_W2T) {
            const int ib = id - kW4_W3T, k = kappa(0, h, j);
            v = k < od ? params[L.w3[net] + k * H + 32 * ib + r] : 0.0f;
        } else {
            const int ib = (id - kW4_W2T) >> 2, ks = (id - kW4_W2T) & 3;
            v = params[L.w2[net] + kappa(ks, h, j) * H + 32 * ib + r];
        }
        unsigned p0, p1, p2;
        split3(v, 0.0f, p0, p1, p2);
        const int at = wop4_index(net, id, 0, lane, j);
        wop4[at] = (unsigned short)p0;
        wop4[at + kWopBlock] = (unsigned short)p1;
        wop4[at + 2 * kWopBlock] = (unsigned short)p2;
    }
}

}  // namespace

namespace aurppo_mlp {

size_t mlp_step4_lds_bytes() { return (size_t)kDyn4; }
size_t mlp_step4_wop_bytes() { return sizeof(unsigned short) * (size_t)kW4Elems; }

int launch_mlp4_prep(const float* params, const MlpLayout& L, int D, int A, void* wop4, hipStream_t s) {
    hipLaunchKernelGGL(k_mlp4_prep, dim3(30), dim3(256), 0, s, params, L, D, A, reinterpret_cast<unsigned short*>(wop4));
    AURPPO_LAUNCH_CHECK("k_mlp4_prep");
    return AURPPO_OK;
}

int launch_mlp_step4(const MlpArgs& a, int grid, hipStream_t s) {
    static bool attr_set[kMaxDevices] = {false};
    const int dslot = aurppo_device_slot();
    if (!attr_set[dslot]) {
        AURPPO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_step4<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)mlp_step4_lds_bytes()));
        AURPPO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_step4<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)mlp_step4_lds_bytes()));
        attr_set[dslot] = true;
    }
    const bool fast = (a.D % 4 == 0) && ((reinterpret_cast<size_t>(a.obs) & 15) == 0) && a.actions == nullptr;
    if (fast) hipLaunchKernelGGL(k_mlp_step4<true>, dim3(grid), dim3(kThreads4), mlp_step4_lds_bytes(), s, a);
    else hipLaunchKernelGGL(k_mlp_step4<false>, dim3(grid), dim3(kThreads4), mlp_step4_lds_bytes(), s, a);
    AURPPO_LAUNCH_CHECK("k_mlp_step4");
    return AURPPO_OK;
}

}  // namespace aurppo_mlp
