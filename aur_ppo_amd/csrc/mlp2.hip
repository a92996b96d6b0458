// K7, two-tile-set variant (k_mlp_step2): the same fused PPO minibatch step as mlp.hip's k_mlp_step
// (src/ppo.py:219-267 over src/models/actor_critic.py:8-51), restructured so the matrix pipe is not idle
// while a wave runs its tanh / loss / staging code.
//
// s_memtime stamps of k_mlp_step (tools/mlp_stamps.py, profiles/r01) put only ~50 % of a workgroup's time in
// MFMA chains: with one wave per SIMD nothing overlaps the epilogues, the loss lanes or the barriers.  Here a
// workgroup is 8 waves = two independent TILE SETS of 4 waves (2 nets x 2 column halves, as before), so
// every SIMD hosts one wave of each set, and each set runs the eight phases of a tile
//     S (land tile)  F1  F2  F3 (head)  L (loss lanes)  B1  B2  B3
// at its own pace: the four waves of a set synchronise on a counter in LDS (set_bar below), never on the
// workgroup barrier, so while one set is in an MFMA chain the other is free to be in VALU / LDS work.
//
// What had to move to make two sets fit one CU (160 KB LDS, 256 VGPRs per wave at 2 waves/SIMD):
//   * W1 is only ever a forward B operand: k_adv_stats_idx lays it out in operand order (32 KB, L2-resident) and
//     each wave streams its 32x64 slice with coalesced loads in B3 of the previous tile; W2 / W3 / biases stay in
//     LDS once per workgroup and are shared by both sets;
//   * dZ2 and dZ1 overwrite H2 and H1 in place (a wave only reads its own column half of them in the phase
//     that produces the gradient), which removes the dZ buffer and one barrier per tile;
//   * the loss phase runs on all 256 lanes of the set (8 lanes per row, 2 action dims per lane) instead of 32;
//     the head-bias / log-std column sums fall out of the same lanes' registers, so the d-logstd tile and the
//     serial column-sum pass are gone;
//   * at the end set 1 hands its accumulators to set 0 through LDS, so a workgroup still writes ONE slab.
#include <stdlib.h>

// The library is built with -ffp-contract=off for K1's bit parity; nothing in this file is held to bit parity (K7 is
// checked to 1e-5 against fp32 autograd), so its VALU code may fuse a*b+c: 1.4 % off the kernel (profiles/r02/ab_fp_contract.txt).
#pragma clang fp contract(fast)
#include "mlp_common.h"

using namespace aurppo_mlp;

// Diagnostic build only (tools/mlp_stamps.py): wave 0 of each set accumulates, per phase, the cycles it spent
// working (slot k) and waiting at the phase's barrier (slot 8 + k) in LDS; dumped to the workspace at the end.
#ifdef AURPPO_MLP_STAMPS
#define STAMP2(k)                                                              \
    do {                                                                       \
        if (w == 0 && lane == 0) {                                             \
            const unsigned long long t__ = __builtin_readcyclecounter();       \
            s_stamp[set][k] += t__ - st_last;                                  \
            st_last = t__;                                                     \
        }                                                                      \
    } while (0)
#else
#define STAMP2(k) do { } while (0)
#endif

namespace {

constexpr int kThreads2 = 512;
constexpr int kSetThreads = 256;
constexpr int kCh = 4;       // operand prefetch depth of the MFMA chains (registers are halved at 2 waves/SIMD)
#ifndef AURPPO_KCH_BIG
#define AURPPO_KCH_BIG 4     // prefetch depth of the 32- and 64-deep chains (A/B knob)
#endif
constexpr int kChB = AURPPO_KCH_BIG;
#ifndef AURPPO_K7_CRITIC_VALU
// Critic head (1 output) on the VALU + actor/critic roles swapped between the sets, so that every SIMD hosts one wave of each.
// Built and measured in round 2: parity green, 5.6 % fewer matrix instructions per SIMD, and 1-2 % SLOWER (146.8 -> 148.5 us,
// three alternating runs; 150 us with single-accumulator dot products): what a tile waits for is not the matrix pipe.  Off.
#define AURPPO_K7_CRITIC_VALU 0
#endif
#ifndef AURPPO_BAR_SLEEP
#define AURPPO_BAR_SLEEP 1   // s_sleep argument of the software barriers' poll loops (A/B knob; 0 = poll back to back)
#endif

// LDS carve-up (floats).  Shared by both sets:
constexpr int oW2 = 0;                       // [2][H][LD]
constexpr int oW3 = oW2 + 2 * H * LD;        // [2][AP][LD]
constexpr int oB1 = oW3 + 2 * AP * LD;       // [2][H]
constexpr int oB2 = oB1 + 2 * H;             // [2][H]
constexpr int oB3 = oB2 + 2 * H;             // [2][AP]
constexpr int oLs = oB3 + 2 * AP;            // [AP]
constexpr int oIvar = oLs + AP;              // [AP]
constexpr int kSharedFloats = oIvar + AP;    // multiple of 4
// per set:
constexpr int pX = 0;                        // [R][LD]
constexpr int pH1 = pX + R * LD;             // [2][R][LD]   H1, later dZ1
constexpr int pH2 = pH1 + 2 * R * LD;        // [2][R][LD]   H2, later dZ2
constexpr int pOut = pH2 + 2 * R * LD;       // [2][R][LDO]  head outputs, then their gradients
constexpr int pRec = pOut + 2 * R * LDO;     // float4[R]    (offset is a multiple of 4 floats)
constexpr int pSrc = pRec + 4 * R;           // int[R]
constexpr int pIdx = pSrc + R;               // int[2][R]
constexpr int kSetFloats = pIdx + 2 * R;     // multiple of 4
static_assert(kSharedFloats % 4 == 0 && pRec % 4 == 0 && kSetFloats % 4 == 0, "float4 alignment of sRec");
constexpr int kAccRegs = 72;                 // gW1 (32) + gW2 (32) + gW3 (2 x 4) per lane
static_assert(kSharedFloats + 2 * kSetFloats >= 4 * kAccRegs * kWave, "hand-over scratch must fit the dead tiles");

// LDS accumulate without reading the result back (ds_add_f64)
__device__ __forceinline__ void lds_add(double* p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(kThreads2, 1) void k_mlp_step2(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double s_red[2][kThreads2 / kWave];
    __shared__ double s_loss[2][6][R];          // per set, per quantity, per tile row: running sums
    __shared__ float s_small[2 * 4][8][5];      // per wave: head-side column sums handed over at the end
    __shared__ float s_gb[4][2][32];            // set 1's bias-gradient column sums, handed over at the end
    __shared__ float s_mean, s_std;
#ifdef AURPPO_MLP_STAMPS
    __shared__ unsigned long long s_stamp[2][16];
    unsigned long long st_last = 0;
#endif
    __shared__ int s_next[2][2];                // per set, per iteration parity: does the set have a tile for the next iteration
    __shared__ int s_first[2];                  // per set: is its first tile real
    __shared__ int s_grab;                      // the workgroup's first grab
    __shared__ int s_bar[2];                    // per set: arrivals at the set's own (software) barriers
    __shared__ int s_pbar[2][2];                // per set and net: arrivals at the barriers only a net's two waves share

#ifdef AURPPO_MLP_STAMPS
    const unsigned long long rt_entry = wall_clock64();   // timeline of the launch in 10-ns ticks (slots 34..38)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps set / net / cb and everything derived from them scalar
    const int set = wave >> 2, w = wave & 3, st = tid & (kSetThreads - 1);
    // Roles.  Wave w of either set sits on SIMD w.  The critic's one-output head is cheaper on the VALU than as padded matrix
    // tiles (F3, dH2 and dW3: 20 of a wave's 180 MFMA-equivalents), which leaves the critic waves with less matrix work than
    // the actor's -- so the sets take opposite roles per SIMD: every SIMD hosts one actor wave and one critic wave.
    const int net = AURPPO_K7_CRITIC_VALU ? ((w >> 1) ^ set) : (w >> 1), cb = w & 1;
    const int wi = net * 2 + cb;                 // role index: W1 operand slice, hand-over slot
    const int D = a.D, A = a.A;
    const int AW = a.continuous ? a.A : 1;
    const int out_dim[2] = {A, 1};

    float* sW2 = lds + oW2;
    float* sW3 = lds + oW3;
    float* sB1 = lds + oB1;
    float* sB2 = lds + oB2;
    float* sB3 = lds + oB3;
    float* sLs = lds + oLs;
    float* sIvar = lds + oIvar;
    float* base = lds + kSharedFloats + set * kSetFloats;
    float* sX = base + pX;
    float* sH1 = base + pH1;
    float* sH2 = base + pH2;
    float* sOut = base + pOut;
    float4* sRec = reinterpret_cast<float4*>(base + pRec);
    int* sSrc = reinterpret_cast<int*>(base + pSrc);
    int* sIdx = reinterpret_cast<int*>(base + pIdx);

    const int n_tiles = (a.h.M + R - 1) / R;
    // Tiles are handed out dynamically (one global counter, reset by k_adv_stats_idx): a workgroup that starts
    // late -- this kernel needs a whole CU's registers, so it waits for CUs that other streams' kernels occupy --
    // or that shares its CU simply takes fewer tiles, instead of holding the whole launch back.
    unsigned* const tile_counter = a.tile_counter;
    // An opaque zero in the address keeps LLVM's atomic optimizer away from the grabs below: it would rewrite each
    // one as a wave-wide scan + readfirstlane of the result, i.e. wait for the atomic's round trip on the spot,
    // while the point of grabbing a tile ahead is that nobody waits for it.
    int zero_off = 0;
    asm volatile("" : "+v"(zero_off));
    int n_idx = -1;
    bool n_ok = false;
    // Global loads below are branch-free (padding lanes read element 0 and discard it): straight-line code lets
    // the compiler count outstanding loads exactly, so a wait for one load does not turn into a wait for all.
    auto load_idx = [&](int tile, int st) -> int {   // wave 0 of the set; lanes >= R mirror lanes < R
        const int m = tile * R + (st & (R - 1));
        const bool ok = tile < n_tiles && m < a.h.M;
        const int v = a.idx[ok ? m : 0];
        return ok ? v : -1;
    };
    // wave 0 of each set owns the set's tile queue: t1 / t2 = tiles of the next two iterations (wave-uniform),
    // t3_raw = lane 0's pending grab for the one after (an atomic issued one tile ahead of its use)
    int t1 = 0, t2 = 0, t3_raw = 0;
    // Priming the queue (tiles of iterations 0..3).  Large minibatches: everything comes from the counter, so a
    // workgroup that starts late owns nothing -- ONE grab of eight consecutive tiles per workgroup, four per set.
    // (Per-set grabs at kernel entry were ~1000 atomics on one address: stamps showed 8-10 us of every launch waiting
    // for them, the weight staging included, since a wait for a load also waits for everything issued before it.)
    // Small minibatches (under four tiles per set): a set's first two tiles are fixed (set s of S takes tiles s and
    // s + S) and the counter starts behind them, because consecutive grabs land in the first sets to arrive (64 tiles
    // on 64 sets took three rounds instead of one).  The grab is issued behind the weight loads.
    const int n_sets = 2 * gridDim.x, my_set = 2 * blockIdx.x + set;
    // Diagnostic (AURPPO_STATIC_TILES): no counter at all -- set s takes tiles s, s + S, s + 2S, ...  Every sum of the launch is
    // then formed in a fixed order and two launches on the same inputs give bit-identical slabs, whatever else the chip is
    // doing (tests/test_determinism.py); in-situ it costs what the static stride cost before the counter.
    const bool stat = a.static_tiles != 0;
    const bool fixed_start = !stat && n_tiles < 4 * n_sets;
    const int dyn_base = fixed_start ? 2 * n_sets : 0;   // tile = dyn_base + counter value
    int grab_raw = 0;
    // ---- stage the shared weights (once per launch).  Every global load of the prologue is issued before the first
    // of them is waited for: section by section (load, wait, store to LDS, next) it was a dozen L2 round trips in a
    // row with 248 workgroups asking for the same 68 KB -- 17 us of a 174 us launch (stamps: tools/mlp_stamps.py).
    {
        float w2r[2][H * H / kThreads2], w3r[2][AP * H / kThreads2], b1r = 0.f, b2r = 0.f, b3r = 0.f, lsr = 0.f;
        static_assert(H * H % kThreads2 == 0 && AP * H % kThreads2 == 0 && 2 * H <= kThreads2, "staging slots");
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int q = 0; q < H * H / kThreads2; ++q) w2r[n][q] = a.params[a.L.w2[n] + tid + q * kThreads2];
#pragma unroll
            for (int q = 0; q < AP * H / kThreads2; ++q) {
                const int e = tid + q * kThreads2, o = e / H;
                w3r[n][q] = a.params[o < out_dim[n] ? a.L.w3[n] + e : a.L.w3[n]];   // rows past the head are zeroed below
            }
        }
        // (selects, not a.L.b1[n]: indexing the argument block with a run-time n turns into a vector load of the kernel
        // arguments and a wait that all the weight loads above queue up behind)
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
        if (tid == 0 && !stat) grab_raw = (int)atomicAdd(tile_counter + zero_off, fixed_start ? 2u : 8u);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int q = 0; q < H * H / kThreads2; ++q) {
                const int e = tid + q * kThreads2;
                sW2[(n * H + e / H) * LD + e % H] = w2r[n][q];
            }
#pragma unroll
            for (int q = 0; q < AP * H / kThreads2; ++q) {
                const int e = tid + q * kThreads2, o = e / H, i = e % H;
                sW3[(n * AP + o) * LD + i] = o < out_dim[n] ? w3r[n][q] : 0.0f;
            }
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
    for (int e = st; e < R * LD; e += kSetThreads) sX[e] = 0.0f;   // columns >= D stay zero for the whole launch
    for (int e = tid; e < 2 * 6 * R; e += kThreads2) (&s_loss[0][0][0])[e] = 0.0;
    if (tid == 0) s_grab = grab_raw;
    __syncthreads();
    if (w == 0) {
        const int g = s_grab;
        const int base = (fixed_start || stat) ? my_set : g + 4 * set;
        t1 = (fixed_start || stat) ? my_set + n_sets : base + 1;
        t2 = stat ? my_set + 2 * n_sets : (fixed_start ? dyn_base + g + set : base + 2);
        if (stat) {
            t3_raw = my_set + 3 * n_sets;
        } else if (fixed_start) {
            if (lane == 0) t3_raw = (int)atomicAdd(tile_counter + zero_off, 1u);   // consumed a tile later
        } else {
            t3_raw = base + 3;
        }
        if (lane == 0) {
            s_first[set] = base < n_tiles ? 1 : 0;
            s_bar[set] = 0;
            s_pbar[set][0] = 0;
            s_pbar[set][1] = 0;
        }
        const int i0 = load_idx(base, st), i1 = load_idx(t1, st);
        n_idx = load_idx(t2, st);
        n_ok = n_idx >= 0;
        if (st < R) {
            sIdx[st] = i0;
            sIdx[R + st] = i1;
        }
    }
    // ---- minibatch advantage statistics from the partials (same order in every workgroup)
    {
        double s = 0.0, q = 0.0;
        for (int b = tid; b < a.n_stat_blocks; b += kThreads2) {
            s += a.stats[2 * b];
            q += a.stats[2 * b + 1];
        }
        const double ts = block_sum<kThreads2 / kWave>(s, s_red[0]);
        const double tq = block_sum<kThreads2 / kWave>(q, s_red[1]);
        if (tid == 0) {
            const double m = ts / (double)a.h.M;
            double var = (tq - ts * m) / (double)(a.h.M - 1);
            if (var < 0.0) var = 0.0;
            s_mean = (float)m;
            s_std = (float)sqrt(var);
        }
    }
    __syncthreads();
    // workgroup-uniform floats are pinned to SGPRs: they are live for the whole tile loop
    auto uniform = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    const float mean = uniform(s_mean), denom = uniform(s_std + 1e-8f);
    const float invM = uniform(1.0f / (float)a.h.M);
    const float g_ent = uniform(-a.h.ent_coef * invM);
    // entropy of the state-independent Gaussian (actor_critic.py:43), summed in action order
    float ent_sum = 0.0f;
    if (a.continuous)
        for (int k = 0; k < A; ++k) ent_sum += (0.5f + 0.9189385332046727f) + sLs[k];
    const float ent_gauss = uniform(ent_sum);

    // ---- persistent accumulators (registers)
    f32x16 gW1[2] = {zero16(), zero16()};
    f32x16 gW2[2] = {zero16(), zero16()};
    f32x4 gW3[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // dW3 as two 16x16 blocks: rows = head outputs, cols cb*32 + 16*q + (lane & 15)
    float gb1 = 0.0f, gb2 = 0.0f;
    float gw3c = 0.0f;      // critic waves (VALU head): dW3[0][cb*32 + lane], lanes 0..31
    float g_b3a[2] = {0.0f, 0.0f}, g_ls[2] = {0.0f, 0.0f}, g_b3c = 0.0f;   // loss-lane (row, j) partial column sums

    const float* __restrict__ w1p = a.w1op + (size_t)wi * 32 * kWave;
    const bool vec4 = (D % 4 == 0) && ((reinterpret_cast<size_t>(a.obs) & 15) == 0);
    // packed records (actions == nullptr): a sample's action row sits behind its 16-B record in one 64-B line
    const float* const act_base = a.actions ? a.actions : reinterpret_cast<const float*>(a.rec) + 4;
    const int act_stride = a.actions ? AW : 16;
    // loss-lane coordinates, also the action staging slots: row lr = st >> 3, action dims lj = st & 7 and lj + 8
    float xr[8];            // next tile's observation elements, in flight
    float ar[2], act_cur[2] = {0.0f, 0.0f};
    float4 p_rec = make_float4(0.f, 0.f, 0.f, 0.f);
    int p_src = -1;

    // Raw values only: padding lanes are zeroed when the tile is landed (S), because touching a loaded value
    // here would wait for it on the spot.
    bool x_ok[2] = {false, false}, a_ok[2] = {false, false};   // vec4: per 16-row pass; otherwise x_ok[0] = row is real
    auto prefetch = [&](const int* sidx, int st) {
        const int lr = st >> 3, lj = st & 7;
        if (vec4) {
            // 16 lanes x float4 per row, 16 rows per pass
            const int c4 = (st & 15) * 4;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int src = sidx[(st >> 4) + 16 * p];
                x_ok[p] = src >= 0 && c4 < D;
                const float4 v = *reinterpret_cast<const float4*>(a.obs + (x_ok[p] ? (size_t)src * D + c4 : (size_t)0));
                xr[4 * p + 0] = v.x; xr[4 * p + 1] = v.y; xr[4 * p + 2] = v.z; xr[4 * p + 3] = v.w;
            }
        } else {
            // 8 lanes per row, columns lj + 8u
            const int src = sidx[lr];
            x_ok[0] = src >= 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) xr[u] = a.obs[(src >= 0 && lj + 8 * u < D) ? (size_t)src * D + lj + 8 * u : (size_t)0];
        }
        {
            const int src = sidx[lr];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                a_ok[u] = src >= 0 && lj + 8 * u < AW;
                ar[u] = act_base[a_ok[u] ? (size_t)src * act_stride + lj + 8 * u : (size_t)0];
            }
        }
        if (w == 0) {   // lanes >= R mirror lanes < R
            p_src = sidx[st & (R - 1)];
            p_rec = a.rec[(size_t)(p_src >= 0 ? p_src : 0) * a.rec_stride];
        }
    };
    auto load_w1 = [&](float (&w1r)[32], int ln) {
#pragma unroll
        for (int m = 0; m < 32; ++m) w1r[m] = w1p[m * kWave + ln];
    };
    __syncthreads();
    prefetch(sIdx, st);
    float w1r[32];   // this wave's W1 slice for the next F1 (reloaded in B3: it only has to live from B3 to F1)
    load_w1(w1r, lane);

#ifdef AURPPO_MLP_STAMPS
    if (tid < 32) (&s_stamp[0][0])[tid] = 0ull;
    __syncthreads();
    st_last = __builtin_readcyclecounter();
    const unsigned long long clk0 = st_last, rt0 = wall_clock64();   // shader cycles vs the 100 MHz wall clock
#endif
    // The two sets do not share barriers inside the tile loop.  A set's four waves meet at a counter in LDS (arrive =
    // one ds_add by lane 0 once the wave's LDS writes have completed, wait = poll until 4 more arrivals than at the
    // previous barrier), so neither set ever waits for the other's longer phase -- with workgroup-wide barriers 38 %
    // of the loop was the tail of barrier intervals where one set finished its epilogue alone -- and a set whose
    // queue is dry simply leaves.  s_barrier is only used before and after the loop.
    int bar_gen = 0;
    auto set_bar = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) (void)__hip_atomic_fetch_add(&s_bar[set], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        bar_gen += 4;
        while (__hip_atomic_load(&s_bar[set], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - bar_gen < 0)
            __builtin_amdgcn_s_sleep(AURPPO_BAR_SLEEP);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // Between the layers of one net only that net's two waves exchange data (each writes its column half of H1 / H2 /
    // dZ2 / dZ1, both read all of it): those four barriers involve two waves, not four, and the actor and critic pairs
    // drift apart between the set-wide barriers around S and L.
    int pbar_gen = 0;
    auto pair_bar = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) (void)__hip_atomic_fetch_add(&s_pbar[set][net], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pbar_gen += 2;
        while (__hip_atomic_load(&s_pbar[set][net], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - pbar_gen < 0)
            __builtin_amdgcn_s_sleep(AURPPO_BAR_SLEEP);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    for (int it = 0; s_first[set] != 0; ++it) {
        // Opaque per-tile copies of the lane coordinates: every LDS address below is re-derived from them inside
        // the phase (one or two VALU ops) instead of being hoisted out of the loop as ~100 loop-invariant
        // address registers that the allocator would then spill and reload in every phase.
        int ln = lane, sl = st;
        asm volatile("" : "+v"(ln), "+v"(sl));
        const int lr = sl >> 3, lj = sl & 7;
        // Order of the global-memory traffic over a tile (a wait for a load also waits for every OLDER load, and
        // the compiler falls back to waiting for all of them around divergent code): S lands rows fetched seven
        // phases ago and issues nothing; F1 consumes the W1 slice fetched in the previous B3, then issues the row
        // fetch of the next tile; L advances the tile queue (atomic + index load); B3 refetches the W1 slice.
        {   // ---- S: land the prefetched tile in LDS
            if (vec4) {
                const int c4 = (sl & 15) * 4;
                if (c4 < D) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        float* d = sX + ((sl >> 4) + 16 * p) * LD + c4;
#pragma unroll
                        for (int q = 0; q < 4; ++q) d[q] = x_ok[p] ? xr[4 * p + q] : 0.0f;
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (lj + 8 * u < D) sX[lr * LD + lj + 8 * u] = x_ok[0] ? xr[u] : 0.0f;
            }
            act_cur[0] = a_ok[0] ? ar[0] : 0.0f;
            act_cur[1] = a_ok[1] ? ar[1] : 0.0f;
            if (sl < R) {
                sSrc[sl] = p_src;
                sRec[sl] = p_rec;
                sIdx[(it & 1) * R + sl] = n_ok ? n_idx : -1;   // indices of tile it+2 replace those of tile it (consumed)
            }
            if (w == 0 && sl == 0) s_next[set][it & 1] = t1 < n_tiles ? 1 : 0;   // t1 = tile of iteration it+1
        }
        STAMP2(0);
        set_bar();
        STAMP2(8);
        {   // ---- F1: H1 = tanh(X W1^T + b1), B operand from registers
            f32x16 acc = zero16();
            const int ij = ln & 31, kk = ln >> 5;
            const float* Xr = sX + ij * LD + kk;
            float av[2][kChB];
#pragma unroll
            for (int u = 0; u < kChB; ++u) av[0][u] = Xr[2 * u];
#pragma unroll
            for (int c = 0; c < H / (2 * kChB); ++c) {
                if (c + 1 < H / (2 * kChB)) {
#pragma unroll
                    for (int u = 0; u < kChB; ++u) av[(c + 1) & 1][u] = Xr[2 * kChB * (c + 1) + 2 * u];
                }
#pragma unroll
                for (int u = 0; u < kChB; ++u)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][u], w1r[kChB * c + u], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            prefetch(sIdx + ((it + 1) & 1) * R, sl);    // rows of tile it+1 (its indices are already in LDS)
            const int col = cb * 32 + ij;
            const float bias = sB1[net * H + col];
#pragma unroll
            for (int e = 0; e < 16; ++e) sH1[(net * R + acc_row(e, ln)) * LD + col] = tanh_fast(acc[e] + bias);
        }
        STAMP2(1);
        pair_bar();
        STAMP2(9);
        {   // ---- F2
            f32x16 acc = zero16();
            const float* W = sW2 + (net * H + cb * 32) * LD;
            const float* Hin = sH1 + net * R * LD;
            mma32<H, kChB, true>(acc, [&](int i, int k) { return Hin[i * LD + k]; }, [&](int k, int j) { return W[j * LD + k]; }, ln);
            const int col = cb * 32 + (ln & 31);
            const float bias = sB2[net * H + col];
#pragma unroll
            for (int e = 0; e < 16; ++e) sH2[(net * R + acc_row(e, ln)) * LD + col] = tanh_fast(acc[e] + bias);
        }
        STAMP2(2);
        pair_bar();
        STAMP2(10);
        if (AURPPO_K7_CRITIC_VALU && net == 1) {
            // ---- F3, critic: v[row] = H2[row] . w3 + b3 for this wave's 16 rows; lane = (k quarter, row): 16 rows x 4 quarters
            // of 16 inputs each (row stride LD is odd, so a half-wave's 16 rows x 2 quarters hit 32 different banks)
            const int r = ln & 15, kq = ln >> 4;
            const float* Hrow = sH2 + (R + cb * 16 + r) * LD + kq * 16;
            const float* Wc = sW3 + AP * LD + kq * 16;             // row 0 of the critic's head
            float hv[16], wv[16], s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) { hv[k] = Hrow[k]; wv[k] = Wc[k]; }
#pragma unroll
            for (int k = 0; k < 16; ++k) s4[k & 3] = fmaf(hv[k], wv[k], s4[k & 3]);
            float sacc = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            sacc += __shfl_xor(sacc, 16, kWave);
            sacc += __shfl_xor(sacc, 32, kWave);
            if (kq == 0) sOut[(R + cb * 16 + r) * LDO] = sacc + sB3[AP];
        } else {   // ---- F3: head (R x AP), each wave of a net takes 16 of the 32 rows
            const float* W = sW3 + net * AP * LD;
            const float* Hin = sH2 + (net * R + cb * 16) * LD;
            const f32x4 acc = mma16<H, true>([&](int i, int k) { return Hin[i * LD + k]; },
                                       [&](int k, int j) { return W[j * LD + k]; }, ln);
            const int col = ln & 15;
            const float bias = sB3[net * AP + col];
#pragma unroll
            for (int e = 0; e < 4; ++e) sOut[(net * R + cb * 16 + 4 * (ln >> 4) + e) * LDO + col] = acc[e] + bias;
        }
        STAMP2(3);
        set_bar();
        STAMP2(11);
        {   // ---- L: distribution + PPO terms, 8 lanes per row; head outputs become their gradients
            if (w == 0) {
                // the grab issued one tile ago is tile it+3: fetch its indices (landed in LDS at the next S), grab again
                const int t3 = dyn_base + __builtin_amdgcn_readfirstlane(t3_raw);
                if (stat) t3_raw = t3 + n_sets;
                else if (ln == 0) t3_raw = (int)atomicAdd(tile_counter + zero_off, 1u);
                // raw index now, validity applied when it is landed at the next S: touching the loaded value here
                // would wait for it, and for the grab issued just above
                const int m = t3 * R + (sl & (R - 1));
                n_ok = t3 < n_tiles && m < a.h.M;
                n_idx = a.idx[n_ok ? m : 0];
                t1 = t2;
                t2 = t3;
            }
            // Everything below is straight-line per lane: all LDS reads go out together, the 8-lane reductions are
            // DPP moves (no LDS round trip), the loss sums are fire-and-forget LDS atomics, and a padding row only
            // masks what is written back.
            float* mu = sOut + (0 * R + lr) * LDO;
            float* vv = sOut + (1 * R + lr) * LDO;
            const bool real = sSrc[lr] >= 0;
            const float4 rc = sRec[lr];
            const float v_new = vv[0];
            const int k0 = lj, k1 = lj + 8;
            const float m0 = mu[k0], m1 = mu[k1];
            float logp = 0.0f, ent = 0.0f, d0, d1;
            PpoSample t;
            if (a.continuous) {
                // Normal(mu, exp(logstd)): log-prob summed over action dims (actor_critic.py:36-43)
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
                // Categorical(logits): log_softmax, log-prob of the stored action, entropy (actor_critic.py:45-50)
                const int ai = (int)sum8(act_cur[0]);   // only lane lj == 0 holds the action index, the others hold 0
                const float z0 = k0 < A ? m0 : -INFINITY, z1 = k1 < A ? m1 : -INFINITY;
                const float mx = max8(fmaxf(z0, z1));
                const float se = sum8((k0 < A ? expf(z0 - mx) : 0.0f) + (k1 < A ? expf(z1 - mx) : 0.0f));
                const float lse = mx + logf(se);
                const float lp0 = k0 < A ? z0 - lse : 0.0f, lp1 = k1 < A ? z1 - lse : 0.0f;
                const float p0 = k0 < A ? expf(lp0) : 0.0f, p1 = k1 < A ? expf(lp1) : 0.0f;
                ent = sum8(-(p0 * lp0) - p1 * lp1);
                logp = sum8((k0 == ai ? lp0 : 0.0f) + (k1 == ai ? lp1 : 0.0f));
                t = ppo_sample(logp, rc.x, rc.y, v_new, rc.w, rc.z, mean, denom, invM, a.h);
                // d logp / d z_k = [k == a] - p_k ;  d H / d z_k = -p_k (log p_k + H)
                d0 = (real && k0 < A) ? t.g_logp * ((k0 == ai ? 1.0f : 0.0f) - p0) + g_ent * (-p0 * (lp0 + ent)) : 0.0f;
                d1 = (real && k1 < A) ? t.g_logp * ((k1 == ai ? 1.0f : 0.0f) - p1) + g_ent * (-p1 * (lp1 + ent)) : 0.0f;
            }
            mu[k0] = d0;
            mu[k1] = d1;
            g_b3a[0] += d0;
            g_b3a[1] += d1;
            if (lj == 0) {
                const float gv = real ? t.g_v : 0.0f;
                vv[0] = gv;
                g_b3c += gv;
                if (real) {
                    double* L = &s_loss[set][0][lr];
                    lds_add(L + 0 * R, (double)t.pg); lds_add(L + 1 * R, (double)t.vl); lds_add(L + 2 * R, (double)ent);
                    lds_add(L + 3 * R, (double)t.okl); lds_add(L + 4 * R, (double)t.kl); lds_add(L + 5 * R, (double)t.cf);
                }
            }
        }
        STAMP2(4);
        set_bar();
        STAMP2(12);
        {   // ---- B1: dH2 -> dZ2 (in place over this wave's half of H2), dW3
            const float* dO = sOut + net * R * LDO;
            const float* W3 = sW3 + net * AP * LD;
            float* H2 = sH2 + net * R * LD;
            f32x16 acc = zero16();
            if (AURPPO_K7_CRITIC_VALU && net == 1) {
                // critic: one output.  dH2[row][col] = g_v[row] * w3[col] (an outer product), dW3[col] += sum_rows g_v[row] * H2[row][col]
                const int cl = cb * 32 + (ln & 31);
                const float w3c = W3[cl];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = dO[acc_row(e, ln) * LDO] * w3c;
                const int r0 = (ln >> 5) * 16;
                float gv[16], hv[16], s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; ++r) { gv[r] = dO[(r0 + r) * LDO]; hv[r] = H2[(r0 + r) * LD + cl]; }
#pragma unroll
                for (int r = 0; r < 16; ++r) s4[r & 3] = fmaf(gv[r], hv[r], s4[r & 3]);
                float sacc = (s4[0] + s4[1]) + (s4[2] + s4[3]);
                sacc += __shfl_xor(sacc, 32, kWave);
                gw3c += sacc;
            } else {
            // dH2 = dO . W3 over the head's outputs: columns past the head width are zero, so K = 8 covers a head of up to
            // 8 outputs (and the critic's single one) with half the matrix instructions of the padded 16
            if (A <= 8 || net == 1)
                mma32<8, kCh, true>(acc, [&](int i, int k) { return dO[i * LDO + k]; }, [&](int k, int j) { return W3[k * LD + cb * 32 + j]; }, ln);
            else
                mma32<AP, kCh, true>(acc, [&](int i, int k) { return dO[i * LDO + k]; }, [&](int k, int j) { return W3[k * LD + cb * 32 + j]; }, ln);
            // dW3 (AP rows) x (in-block cb) as two 16x16 blocks: A = dO^T, B = H2 (still the activations) -- a 32x32 MFMA
            // block would spend half its rows on padding
#pragma unroll
            for (int q = 0; q < 2; ++q)
                gW3[q] += mma16<R, true>([&](int i, int k) { return dO[k * LDO + i]; },
                                         [&](int k, int j) { return H2[k * LD + cb * 32 + 16 * q + j]; }, ln);
            }
            const int col = cb * 32 + (ln & 31);
            float colsum = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float* hp = H2 + acc_row(e, ln) * LD + col;
                const float h = *hp;
                const float dz = acc[e] * (1.0f - h * h);
                *hp = dz;
                colsum += dz;
            }
            colsum += __shfl_xor(colsum, 32, kWave);
            gb2 += colsum;
        }
        STAMP2(5);
        pair_bar();
        STAMP2(13);
        {   // ---- B2: dW2, dH1 -> dZ1 (in place over this wave's half of H1)
            const float* dZ = sH2 + net * R * LD;
            float* H1 = sH1 + net * R * LD;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
                mma32<R, kChB, true>(gW2[ob], [&](int i, int k) { return dZ[k * LD + ob * 32 + i]; },
                         [&](int k, int j) { return H1[k * LD + cb * 32 + j]; }, ln);
            const float* W2 = sW2 + net * H * LD;
            f32x16 acc = zero16();
            mma32<H, kChB, true>(acc, [&](int i, int k) { return dZ[i * LD + k]; }, [&](int k, int j) { return W2[k * LD + cb * 32 + j]; }, ln);
            const int col = cb * 32 + (ln & 31);
            float colsum = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float* hp = H1 + acc_row(e, ln) * LD + col;
                const float h = *hp;
                const float dz = acc[e] * (1.0f - h * h);
                *hp = dz;
                colsum += dz;
            }
            colsum += __shfl_xor(colsum, 32, kWave);
            gb1 += colsum;
        }
        STAMP2(6);
        pair_bar();
        STAMP2(14);
        {   // ---- B3: dW1 (two out-blocks x in-block cb of D)
            if (cb * 32 < D) {
                const float* dZ = sH1 + net * R * LD;
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
                    mma32<R, kChB, true>(gW1[ob], [&](int i, int k) { return dZ[k * LD + ob * 32 + i]; },
                             [&](int k, int j) { return sX[k * LD + cb * 32 + j]; }, ln);
            }
            load_w1(w1r, ln);
        }
        STAMP2(7);
        set_bar();
        STAMP2(15);
        if (!s_next[set][it & 1]) break;     // written by the set's wave 0 in this iteration's S phase
    }
#ifdef AURPPO_MLP_STAMPS
    const unsigned long long rt_loop_end = wall_clock64();
#endif
    __syncthreads();   // the hand-over below reuses the weights' LDS: both sets must have left the loop
#ifdef AURPPO_MLP_STAMPS
    const unsigned long long rt_both_done = wall_clock64();
#endif

    int le = lane, se = st;   // fresh opaque copies: nothing lane-derived has to stay live across the tile loop
    asm volatile("" : "+v"(le), "+v"(se));
    // ---- hand-over: set 1 parks its accumulators in the (dead) tile memory, set 0 adds them and writes the slab
    float* park = lds + (size_t)wi * kAccRegs * kWave + le;   // [role][reg][le]: set 0's wave of the same role reads it
    // head-side column sums: fold the 8 rows a wave's loss lanes cover
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
        if (AURPPO_K7_CRITIC_VALU && net == 1) park[64 * kWave] = gw3c;      // (gW3 is unused by a VALU-head critic wave)
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
        if (AURPPO_K7_CRITIC_VALU && net == 1) {
            if (le < 32) slab[a.L.w3[1] + cb * 32 + le] = gw3c + park[64 * kWave];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = 4 * (le >> 4) + e, c = cb * 32 + (le & 15);   // 16x16 accumulator layout
                if (o < out_dim[net]) {
                    slab[a.L.w3[net] + o * H + c] = gW3[0][e] + park[(64 + e) * kWave];
                    slab[a.L.w3[net] + o * H + c + 16] = gW3[1][e] + park[(68 + e) * kWave];
                }
            }
        }
        if (le < 32) {
            slab[a.L.b1[net] + col] = gb1 + s_gb[wi][0][le];
            slab[a.L.b2[net] + col] = gb2 + s_gb[wi][1][le];
        }
        if (w == 0) {
            // head biases / log-std: le = action dim k (< 16): column sums over all 8 waves, fixed order
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
            // loss partial sums of this workgroup: 2 sets x 32 rows per quantity
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
    __syncthreads();
    if (tid < 32) a.stamps[(size_t)blockIdx.x * 40 + tid] = (&s_stamp[0][0])[tid];
    if (tid == 0) {   // shader cycles and 100 MHz wall-clock ticks of the tile loop + hand-over: the clock the launch ran at
        a.stamps[(size_t)blockIdx.x * 40 + 32] = __builtin_readcyclecounter() - clk0;
        a.stamps[(size_t)blockIdx.x * 40 + 33] = wall_clock64() - rt0;
        a.stamps[(size_t)blockIdx.x * 40 + 34] = rt_entry;          // absolute: launch skew between workgroups
        a.stamps[(size_t)blockIdx.x * 40 + 35] = rt0 - rt_entry;     // prologue (weights, statistics, queue, first fetch)
        a.stamps[(size_t)blockIdx.x * 40 + 36] = rt_loop_end - rt0;  // set 0 / wave 0's tile loop
        a.stamps[(size_t)blockIdx.x * 40 + 37] = rt_both_done - rt_loop_end;   // waiting for the other set
        a.stamps[(size_t)blockIdx.x * 40 + 38] = wall_clock64() - rt_both_done;  // hand-over + slab write (issue)
    }
#endif
}

}  // namespace

namespace aurppo_mlp {

size_t mlp_step2_lds_bytes() { return sizeof(float) * (size_t)(kSharedFloats + 2 * kSetFloats); }

int launch_mlp_step2(const MlpArgs& a, int grid, hipStream_t s) {
    static bool attr_set[kMaxDevices] = {false};
    const int dslot = aurppo_device_slot();
    if (!attr_set[dslot]) {
        AURPPO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_step2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)mlp_step2_lds_bytes()));
        attr_set[dslot] = true;
    }
    hipLaunchKernelGGL(k_mlp_step2, dim3(grid), dim3(kThreads2), mlp_step2_lds_bytes(), s, a);
    AURPPO_LAUNCH_CHECK("k_mlp_step2");
    return AURPPO_OK;
}

}  // namespace aurppo_mlp
