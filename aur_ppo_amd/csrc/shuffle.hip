// K2: numpy-legacy MT19937 + Fisher-Yates shuffle, bit-exact with np.random.shuffle under
// np.random.seed(s)  (src/ppo.py:182,213-217; src/robot_ppo.py:335-338).
//
// The algorithm lives in numpy (third-party dependency of the reference): init_genrand seeding,
// 624-word twist, tempering, random_interval = masked rejection on one 32-bit draw per trial,
// then `for i = n-1..1: swap(x[i], x[j_i])`.
//
// Design (gfx950).  The swap chain is n dependent memory transactions if done literally, so it is
// split into the part that is inherently a stream and the part that is not:
//   1a. k_mt_fill -- one workgroup runs the twist only, as 227-word phases of the recurrence
//      x[m] = x[m-227] ^ mix(x[m-624], x[m-623]): a lane owns one offset of every phase (x[m-227] is its own previous
//      output, a register), two phases share a barrier, and eight helper waves store the words -- UNTEMPERED: the consumers
//      temper what they read -- to a ring in HBM one barrier behind the four that twist.  It runs one shuffle ahead of its
//      consumer on the handle's own stream.
//   1b. k_fy_accept3 (default; k_fy_accept is the one-workgroup build it grew from, AURPPO_K2_ACCEPT=1) -- turns the draws into
//      accept/reject decisions.  Whether draw p is accepted depends on the index i it is tried against, i.e. on how
//      many earlier draws were accepted -- a triangular system.  A thread resolves its own consecutive draws exactly
//      given its starting index; inside a workgroup the starting indices are the fixed point of
//      "i_t = i0 - (#accepts of earlier threads)", iterated with prefix sums until no count changes; BETWEEN workgroups
//      only the index at a chunk boundary is handed on, and a workgroup has solved its 16 384-draw chunk from a guessed
//      index -- and listed the few hundred draws the true one can change -- before its predecessor's word arrives.
//      Output: the swap targets j[1..n).
//   2. k_fy_link / k_fy_resolve -- given j, the final content of every position is found in
//      parallel with no swaps at all (link behind the accept; resolve on the fill stream, beside the NEXT accept).  Step s writes old x[s] into position j_s, so "what sits in
//      position q just before step t" is "what step min{s>t : j_s=q} put there", recursively.
//      Linked lists per target (atomicExch) give those predecessor sets; chains are O(log n) and
//      almost always empty, so each element resolves with a handful of L2-resident loads.
// The generator state (key[624], pos) stays on the device between calls, like numpy's global stream.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

struct aurppo_rng {
    uint32_t* d_state;   // key[624] + pos: the generator at the stream ORIGIN (last seed / set_state)
    uint32_t* d_last;    // [624] untempered form of the last block written to the ring
    uint32_t* d_ring;    // ring of tempered draws; stream word g lives at d_ring[g % ring_cap]
    long long* d_pos;    // [0] words written (stream offset), [1] words consumed (read cursor), [2] sticky error
    // Shuffle e's accept + link run on the caller's stream while shuffle e-1's resolve runs on the fill stream, so the
    // per-shuffle scratch is multi-buffered: swap targets and list links by parity, list heads three deep (the accept
    // of shuffle e clears the heads shuffle e+1 will link into while shuffle e-1's resolve may still read its own).
    int32_t* d_j[2];     // swap targets
    int32_t* d_head[3];  // list head per target position
    int32_t* d_next[2];  // list link per step
    int head_clean[3];   // how many leading entries of d_head[k] are known to be -1 when its next user starts
    int32_t* d_tmp;      // out-of-place result for the in-place API
    int32_t* d_meta;     // accept scratch
    int max_n;
    size_t head_cap;     // entries per d_head buffer (max_n rounded up to whole clear chunks)
    size_t ring_cap;     // words, a multiple of 624
    hipStream_t fill_stream;   // the twist runs one shuffle ahead of its consumer on this stream; resolves follow it there
    hipStream_t post_stream;   // AURPPO_K2_POST_STREAM=1: link + resolve run here, beside the next accept AND the next fill
    int use_post;
    hipEvent_t ev_fill[2], ev_acc[2], ev_link[2], ev_post[2], ev_res, ev_sync;
    unsigned long long* d_relay;   // k_fy_accept3: one word per chunk of draws {launch tag, index after the chunk} + the `done` word
    int relay_cap;
    unsigned acc_gen;    // launch tag of the next k_fy_accept3 (never reset: a stale word of an earlier shuffle must not match)
    long seq;            // shuffles issued since the last (re)seed
    double primed_need;  // words-per-shuffle the look-ahead fill was sized for (0: nothing in flight)
};

namespace {

constexpr int kMtN = 624, kMtM = 397, kMtD = kMtN - kMtM;  // 227

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__global__ void k_mt_seed(uint32_t* state, uint32_t seed) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t x = seed;
    state[0] = x;
    for (int i = 1; i < kMtN; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
        state[i] = x;
    }
    state[kMtN] = kMtN;  // pos: next draw regenerates
}

constexpr int kFillThreads = 256;
// The first kRingMirror words of the ring are repeated after its end, so a reader may take up to kRingMirror
// consecutive words from any slot with one straight (vector) access.
constexpr int kRingMirror = 64;

// Append freshly twisted words to the ring so that (words written - words consumed) reaches `target` words (whole
// 624-word blocks, at most nblk_max).  Continues from d_last; all bookkeeping is on the device.
//
// The recurrence is x[m] = x[m-227] ^ mix(x[m-624], x[m-623]).  Taken 227 words at a time ("phase"), with thread
// t owning offset t of EVERY phase, the x[m-227] term is the thread's own previous output -- a register.  The two
// other inputs were written at least two phases earlier, so they come from a small circular window in LDS.  What is
// left on the critical path of a phase is one XOR, one LDS write and the barrier that publishes it -- and two phases
// share one barrier (below); the (coalesced) ring store hangs off the side.  The block formulation
// this replaces needed three barrier-separated passes with an LDS round trip each per 624 words.
constexpr int kFillWin = 2048;   // LDS window over the untempered stream (power of two, >= 624 + 2 * 454)
constexpr int kFillProd = 256;   // waves 0-3: the twist (227 lanes carry a word each)
#ifndef AURPPO_FILL_ALL
#define AURPPO_FILL_ALL 768
#endif
constexpr int kFillAll = AURPPO_FILL_ALL;    // waves 4..: ring stores of the pair produced one barrier earlier (768: eight waves, a word each)
constexpr int kPair = 2 * kMtD;  // words per barrier interval

// Roles.  What a barrier interval must contain is: LDS reads of the mix inputs -> two XORs -> LDS writes -> barrier.
// Everything else a word needs (the ring slot arithmetic, the global store and the mirrored head; until round 3 also the
// tempering, 13 VALU ops) is off that chain: the producer waves hand the untempered pair to more waves of the same workgroup
// through the LDS window they write anyway, and those store it during the NEXT interval, in the issue slots the producers
// leave free while they wait for LDS.  (First version: every lane tempered and stored its own
// words between the LDS writes and the barrier -- ~95 instructions per wave on the chain, 750 cycles per interval.
// With four helper waves taking two words each, one after the other, the helpers were the last to reach the barrier:
// eight of them take one word each.  Round 3: twelve waves share four SIMDs, and the helpers' instructions -- not the barrier --
// paced the interval (three phases per barrier with fifteen waves: 411 us per shuffle's draws against 360; four or two helper
// waves taking two / four words each: 1.7x / 2.6x slower).  So the tempering moved to the CONSUMERS (k_fy_accept*: 13 more
// operations per draw on six CUs) and the helpers only store: 319 us.  Without helpers at all -- each producer lane storing its
// own two words behind the barrier -- it was 429 us: the stores' address arithmetic and exec branches back on the chain.)
__global__ __launch_bounds__(kFillAll) void k_mt_fill(uint32_t* __restrict__ last, uint32_t* __restrict__ ring,
                                                      long long ring_cap, long long* __restrict__ posv,
                                                      long long target, int nblk_max, int cur_slot) {
    __shared__ uint32_t win[kFillWin + kWave];      // + a dummy slot per lane for writes that are masked off
    const int tid = threadIdx.x;
    const long long S = posv[0];
    // Cursor as of the end of a SPECIFIC earlier shuffle (slot 4/5 by parity; slot 1 = live value for the
    // first fills after a restart).  Reading a fixed shuffle's cursor, not whatever a concurrently running
    // consumer has published, makes every look-ahead fill produce exactly one shuffle's consumption.
    const long long cur = posv[cur_slot];
    long long want = target - (S - cur);
    int nblk = want > 0 ? (int)((want + kMtN - 1) / kMtN) : 0;
    if (nblk > nblk_max) nblk = nblk_max;
    if (nblk == 0) return;
    // local numbering: x[0 .. 624) is the last block of the previous call, new words are x[624 ..  624 + total)
    const int total = nblk * kMtN;
    const int m_end = kMtN + total;
    const int cap = (int)ring_cap;
    const int at0 = (int)(S % ring_cap);                           // ring slot of x[624]
    for (int k = tid; k < kMtN; k += kFillAll) win[k] = last[k];
    __syncthreads();
    const int npair = (total + kPair - 1) / kPair;
    const bool producer = tid < kFillProd;                         // wave-uniform
    if (producer) {
        const bool lane_on = tid < kMtD;
        const int dummy = kFillWin + (tid & (kWave - 1));
        int m = kMtN + tid;                                         // my word of the pair's first phase
        uint32_t prev = lane_on ? win[kMtM + tid] : 0u;             // x[m - 227]
        // Two phases per barrier: phase p reads words written in phases p-3 and p-2, so both phases of a pair only need
        // what was complete at the previous barrier; x[m - 227] is this lane's own previous output -- a register.
        uint32_t a0 = win[(m - kMtN) & (kFillWin - 1)], b0 = win[(m - kMtN + 1) & (kFillWin - 1)];
        uint32_t a1 = win[(m - kMtM) & (kFillWin - 1)], b1 = win[(m - kMtM + 1) & (kFillWin - 1)];
        for (int pr = 0; pr < npair; ++pr) {
            const uint32_t n0 = prev ^ mt_mix(a0, b0);          // x[m]       = x[m - 227] ^ ...
            const uint32_t n1 = n0 ^ mt_mix(a1, b1);            // x[m + 227] = x[m]       ^ ...
            prev = n1;
            win[(lane_on && m < m_end) ? (m & (kFillWin - 1)) : dummy] = n0;
            win[(lane_on && m + kMtD < m_end) ? ((m + kMtD) & (kFillWin - 1)) : dummy] = n1;
            m += kPair;
            __syncthreads();
            a0 = win[(m - kMtN) & (kFillWin - 1)];
            b0 = win[(m - kMtN + 1) & (kFillWin - 1)];
            a1 = win[(m - kMtM) & (kFillWin - 1)];
            b1 = win[(m - kMtM + 1) & (kFillWin - 1)];
        }
    } else {
        constexpr int kHelp = kFillAll - kFillProd;                 // helper lanes; each takes words c, c + kHelp, ... of a pair
        const int c0 = tid - kFillProd;
        for (int pr = 0; pr <= npair; ++pr) {
            if (pr > 0) {
#pragma unroll
                for (int c = c0; c < kPair; c += kHelp) {
                    const int w = kMtN + (pr - 1) * kPair + c;      // in the pair finished one barrier ago
                    const uint32_t t = win[w & (kFillWin - 1)];     // untempered: the consumers temper what they read
                    int at = at0 + (w - kMtN);
                    at -= at >= cap ? cap : 0;
                    if (w < m_end) {
                        ring[at] = t;
                        if (at < kRingMirror) ring[cap + at] = t;
                    }
                }
            }
            if (pr < npair) __syncthreads();
        }
    }
    __syncthreads();
    for (int k = tid; k < kMtN; k += kFillAll) last[k] = win[(m_end - kMtN + k) & (kFillWin - 1)];
    if (tid == 0) posv[0] = S + total;
}

// (Re)start the stream at a generator state: ring[0 .. 624-pos) = temper(key[pos..]), last block = key
__global__ __launch_bounds__(kFillThreads) void k_mt_origin(const uint32_t* __restrict__ state,
                                                            uint32_t* __restrict__ last, uint32_t* __restrict__ ring,
                                                            long long ring_cap, long long* __restrict__ posv) {
    const int tid = threadIdx.x;
    const int pos = (int)state[kMtN];
    for (int k = tid; k < kMtN; k += kFillThreads) {
        last[k] = state[k];
        if (k >= pos) {
            const uint32_t t = state[k];
            ring[k - pos] = t;
            if (k - pos < kRingMirror) ring[ring_cap + (k - pos)] = t;
        }
    }
    if (tid == 0) {
        posv[0] = kMtN - pos;
        posv[1] = 0;
        posv[2] = 0;
        posv[4] = 0;
        posv[5] = 0;
    }
}

// Generator state at the read cursor, numpy's (key, pos): the untempered 624-word block holding the
// cursor (pos = 624 when the cursor sits exactly on a block boundary: numpy twists lazily).
__global__ __launch_bounds__(kFillThreads) void k_state_at_cursor(const uint32_t* __restrict__ state,
                                                                  const uint32_t* __restrict__ ring,
                                                                  long long ring_cap,
                                                                  const long long* __restrict__ posv,
                                                                  uint32_t* __restrict__ out) {
    const int tid = threadIdx.x;
    const int pos0 = (int)state[kMtN];
    const long long p = (long long)pos0 + posv[1];
    long long blk = p / kMtN;
    int pos = (int)(p % kMtN);
    if (pos == 0 && p > 0) {
        blk -= 1;
        pos = kMtN;
    }
    for (int k = tid; k < kMtN; k += kFillThreads)
        out[k] = blk > 0 ? ring[(blk * kMtN - pos0 + k) % ring_cap] : state[k];
    if (tid == 0) out[kMtN] = (uint32_t)pos;
}

// Diagnostic build only (tools/accept_stamps.py, -DAURPPO_ACC_STAMPS): thread 0 accumulates cycles per section of
// a step into posv[8..]; the product library is built without it.
#ifdef AURPPO_ACC_STAMPS
#define ASTAMP(k)                                                      \
    do {                                                               \
        if (tid == 0) {                                                \
            const unsigned long long t__ = __builtin_readcyclecounter(); \
            st_acc[k] += t__ - st_last;                                \
            st_last = t__;                                             \
        }                                                              \
    } while (0)
#else
#define ASTAMP(k) do { } while (0)
#endif

// Wave-wide inclusive prefix sum with DPP moves only (no LDS): row_shr 1/2/4/8 inside each row of 16 lanes, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add(int v) {
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int row_incl_scan(int v) {   // within rows of 16 lanes
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    return v;
}
__device__ __forceinline__ int wave_incl_scan(int v) {
    v = row_incl_scan(v);
    v = dpp_add<0x142, 0xa>(v);
    v = dpp_add<0x143, 0xc>(v);
    return v;
}

// kWpt = draws per thread per step.
//
// One step: every thread owns kWpt consecutive draws.  Whether a draw is accepted depends on the index it is
// tried against, i.e. on the number of accepts before it; a thread resolves its own draws exactly given its
// starting index i0, and the starting indices are the fixed point of "i0 = i_cur - (#accepts of earlier threads)".
// Round: publish the per-wave accept counts (+ "did any of my counts change in the last evaluation"), ONE
// barrier, every wave rebuilds its base from the 16 wave sums; when nobody changed, (excl, cnt) is the solution.
// Common case per evaluation (count_fast): all of a thread's draws sit in one mask octave and none falls in the
// window (i0 - kWpt, i0] the index moves through, so "#draws <= i0 - kWpt" is the count -- ~5 VALU ops per draw.
// Anything else (octave edge, tail of the shuffle, a draw inside the window) sends the whole wave through the
// literal per-draw loop behind a wave-uniform branch, so the fast path never pays for it.
template <int kAccThreads, int kWpt>
__global__ __launch_bounds__(kAccThreads) void k_fy_accept(const uint32_t* __restrict__ ring, long long ring_cap,
                                                           int32_t* __restrict__ j, int n,
                                                           long long* __restrict__ posv, int done_slot) {
    constexpr int kAccStep = kAccThreads * kWpt;
    constexpr int kWaves = kAccThreads / kWave;   // <= 16: one row of lanes reads all wave sums
    static_assert(kWaves >= 1 && kWaves <= 16, "wave sums are combined inside one DPP row");
    __shared__ int s_wsum[2][kWaves];
    __shared__ int s_chg[2][kWaves];
    __shared__ int s_end;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long cursor = posv[1];               // stream offset of the next unread draw
    const long long avail = posv[0];          // draws written so far (the fill for this shuffle has completed)
    long long rpos = cursor % ring_cap;       // cursor's slot in the ring, kept incrementally
    int i_cur = n - 1;                        // next index to draw a target for
    float rate = 0.72f;                        // acceptance rate guess, refreshed every step
    int par = 0;                              // exchange-buffer parity (runs on across steps)
    // this step's draws are fetched one step ahead (every step but the last consumes exactly kAccStep)
    uint32_t ynext[kWpt];
    // Branch-free: draws past `avail` are fetched from some valid ring slot and never looked at (nhave below),
    // so the loads are straight-line code and the wait at the top of the next step is for exactly these loads.
    // A thread's kWpt consecutive words come as 16-byte vector loads (4-byte aligned: the cursor is arbitrary);
    // dword loads at this stride cost one cache-line access per lane per word -- the vector memory pipe, not the
    // arithmetic, was what a step waited for.  The ring's mirrored head makes the run contiguous at the wrap.
    static_assert(kWpt % 4 == 0 && kWpt <= kRingMirror, "draws are fetched four at a time from a mirrored ring");
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
    auto fetch = [&](long long rp) {
        long long r0 = rp + (long long)tid * kWpt;
        if (r0 >= ring_cap) r0 -= ring_cap;
        if (r0 >= ring_cap) r0 = 0;
        const u32x4u* src = reinterpret_cast<const u32x4u*>(ring + r0);
#pragma unroll
        for (int q = 0; q < kWpt / 4; ++q) {
            const u32x4u v = src[q];
            ynext[4 * q + 0] = v.x; ynext[4 * q + 1] = v.y; ynext[4 * q + 2] = v.z; ynext[4 * q + 3] = v.w;
        }
    };
    fetch(rpos);
#ifdef AURPPO_ACC_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter(), n_steps = 0, n_iters = 0;
#endif
    while (i_cur >= 1 && cursor < avail) {
        const long long base = cursor + (long long)tid * kWpt;
        uint32_t y[kWpt];
        int nhave = 0;                             // my draws that exist (a prefix of the kWpt)
#pragma unroll
        for (int u = 0; u < kWpt; ++u) {
            y[u] = mt_temper(ynext[u]);     // the ring holds the untempered words
            nhave += (base + u) < avail ? 1 : 0;
        }
        ASTAMP(0);   // draws of this step in registers (waits for the fetch issued one step ago)
        {
            long long rn = rpos + kAccStep;
            if (rn >= ring_cap) rn -= ring_cap;
            fetch(rn);
        }
        ASTAMP(1);   // next fetch issued
        // the literal rule for my draws from starting index i0: #accepted; optionally emits targets / #consumed
        auto walk = [&](int i0, bool emit, int& consumed) -> int {
            int i = i0, acc = 0;
            consumed = 0;
#pragma unroll
            for (int u = 0; u < kWpt; ++u) {
                if (u < nhave && i >= 1) {
                    const uint32_t v = y[u] & (0xffffffffu >> __clz(i));
                    ++consumed;
                    if (v <= (uint32_t)i) {
                        if (emit) j[i] = (int32_t)v;
                        --i;
                        ++acc;
                    }
                }
            }
            return acc;
        };
        // Shortcut evaluation at starting index i0.  Returns #draws <= i0 - kWpt and the slack of that count: it
        // stays the same (and the shortcut stays valid) while i0 moves up by at most `up - kWpt` or down by at most
        // `dn` -- the distances from i0 - kWpt to the nearest draw above / at-or-below it.
        uint32_t dn = 0, up = 0;
        bool sure = false;
        int ref_i0 = 0;          // where (cnt, dn, up) were evaluated
        bool wave_fast = false;  // every lane of this wave is on the shortcut
        auto eval = [&](int i0) -> int {
            const bool shape = nhave == kWpt && i0 > kWpt && __clz(i0) == __clz(i0 - kWpt);
            const uint32_t mask = 0xffffffffu >> __clz(i0 | 1);
            const uint32_t lo = (uint32_t)(i0 - kWpt);
            int c = 0;
            uint32_t d = 0xffffffffu, p = 0xffffffffu;
#pragma unroll
            for (int u = 0; u < kWpt; ++u) {
                const uint32_t v = y[u] & mask;
                const bool le = v <= lo;
                c += le ? 1 : 0;
                d = le ? min(d, lo - v) : d;
                p = le ? p : min(p, v - lo - 1u);
            }
            dn = d;
            up = p;
            sure = shape && p >= (uint32_t)kWpt;      // no draw inside the window (i0 - kWpt, i0] the index moves through
            ref_i0 = i0;
            wave_fast = __builtin_amdgcn_ballot_w64(!sure) == 0ull;
            if (!wave_fast) {   // wave-uniform and rare: the literal rule for everybody
                int consumed;
                c = walk(i0, false, consumed);
            }
            return c;
        };
        // is (cnt, shortcut) evaluated at ref_i0 still right at i0?
        auto still_ok = [&](int i0) -> bool {
            const int delta = i0 - ref_i0;
            const bool shape = i0 > kWpt && __clz(i0) == __clz(i0 - kWpt) && __clz(i0) == __clz(ref_i0);
            const bool room = delta >= 0 ? up >= (uint32_t)(kWpt + delta) : dn >= (uint32_t)(-delta);
            return delta == 0 || (sure && shape && room);
        };
        int cur_i0 = i_cur - (int)(rate * (float)(tid * kWpt));   // first guess of my starting index
        int cnt = eval(cur_i0);
        int incl = wave_incl_scan(cnt);
        bool wchg = true;                                // forces the first exchange
        int total = 0;
        ASTAMP(2);   // first guess
        for (;;) {
#ifdef AURPPO_ACC_STAMPS
            ++n_iters;
#endif
            if (lane == kWave - 1) {
                s_wsum[par][wave] = incl;
                s_chg[par][wave] = wchg ? 1 : 0;
            }
            ASTAMP(6);   // (inside the loop) publish
            __syncthreads();
            ASTAMP(7);   // (inside the loop) barrier
            const int ws = lane < kWaves ? s_wsum[par][lane] : 0;
            const int wc = lane < kWaves ? s_chg[par][lane] : 0;
            par ^= 1;
            const int pre = row_incl_scan(ws);                          // lanes 0..15: prefix over the wave sums
            total = __builtin_amdgcn_readlane(pre, kWaves - 1);
            const int wbase = wave > 0 ? __builtin_amdgcn_readlane(pre, wave > 0 ? wave - 1 : 0) : 0;
            const bool any = __builtin_amdgcn_ballot_w64(wc != 0) != 0ull;
            if (!any) break;     // no wave's counts moved in the last evaluation: (cur_i0, cnt) is the fixed point
            cur_i0 = i_cur - (wbase + incl - cnt);
            // A lane recounts only if its index left the slack of its last evaluation; a wave whose lanes all
            // stay inside keeps its counts, its scan and its published sum (typical: one or two waves per step redo).
            if (__builtin_amdgcn_ballot_w64(!still_ok(cur_i0)) == 0ull) {
                wchg = false;
            } else {
                const int cnt2 = eval(cur_i0);
                wchg = __builtin_amdgcn_ballot_w64(cnt2 != cnt) != 0ull;
                cnt = cnt2;
                incl = wave_incl_scan(cnt);
            }
        }
        ASTAMP(3);   // fixed point reached
        // decisions are final: emit the targets
        int consumed = kWpt;
        if (!wave_fast) {
            (void)walk(cur_i0, true, consumed);
        } else {
            const uint32_t mask = 0xffffffffu >> __clz(cur_i0 | 1);
            const uint32_t lo = (uint32_t)(cur_i0 - kWpt);
            int i = cur_i0;
#pragma unroll
            for (int u = 0; u < kWpt; ++u) {
                const uint32_t v = y[u] & mask;
                if (v <= lo) j[i--] = (int32_t)v;
            }
        }
        ASTAMP(4);   // targets emitted
        long long step_words = avail - cursor < kAccStep ? avail - cursor : kAccStep;
        if (total >= i_cur) {
            // last step: words consumed = everything up to and including the draw that filled i = 1
            const int my_end_i = cur_i0 - cnt;          // index after my draws
            if (tid == 0) s_end = -1;
            __syncthreads();
            if (cnt > 0 && my_end_i == 0) s_end = tid * kWpt + consumed;   // unique thread
            __syncthreads();
            step_words = s_end;
        }
        rate = step_words > 0 ? (float)total / (float)step_words : rate;
        i_cur -= total;
        cursor += step_words;
        rpos += step_words;
        if (rpos >= ring_cap) rpos -= ring_cap;
        ASTAMP(5);   // step bookkeeping
#ifdef AURPPO_ACC_STAMPS
        ++n_steps;
#endif
    }
#ifdef AURPPO_ACC_STAMPS
    if (tid == 0) {
        for (int k = 0; k < 8; ++k) posv[8 + k] = (long long)st_acc[k];
        posv[16] = (long long)n_steps;
        posv[17] = (long long)n_iters;
    }
#endif
    if (tid == 0) {
        posv[1] = cursor;
        posv[done_slot] = cursor;      // this shuffle's final cursor, for the fill two shuffles later
        if (i_cur >= 1) posv[2] = 1;   // ran out of draws before the shuffle finished: sticky error
    }
}

// k_fy_accept3 -- a relay between WORKGROUPS, with k_fy_accept's fixed point inside each.  (A relay between the WAVES of one
// workgroup, k_fy_accept2, was bit-exact and slower -- 515 us per shuffle against 394 -- and was retired in round 4: DESIGN 4.2c.)
//
// k_fy_accept is one workgroup walking the draw stream 8192 draws at a time: ~89 steps of ~4.6 us per 524 288-element shuffle,
// every one on the chain (in-situ 437 us per shuffle, four per update: the longest pipeline of the update in round 3).  What a
// stretch of draws needs from everything before it is ONE number, the index it starts at; where it starts in the stream is
// known beforehand, because every chunk but the last consumes all of its draws.  So:
//   * G workgroups (one per spare CU) own chunks g, g + G, ... of 1024 x kWpt consecutive draws (32 768 at kWpt = 32: 23 chunks
//     per shuffle of 524 288);
//   * a workgroup solves its chunk AHEAD of time from a guessed starting index (expected trajectory from the latest chunk anybody
//     has confirmed), with k_fy_accept's rounds -- one workgroup barrier each --, and forms the window [LO, HI] of starting indices
//     for which every lane's count stands;
//   * thread 0 then polls its predecessor's word {launch tag, index after the chunk} (coherent loads); inside the window it
//     publishes its own word at once -- the chain per chunk is then one cross-workgroup hop --, otherwise the workgroup re-runs
//     the rounds from the true index (the lanes whose slack was exceeded recount) and publishes after;
//   * targets are emitted after the word is out.  The workgroup that ends the shuffle (or runs out of draws) writes the cursor
//     and raises the `done` word, on which everybody still waiting leaves.
// Decisions and the words consumed are exactly numpy's: the same literal rule decides wherever a shortcut does not apply.
template <int kAccThreads, int kWpt>
__global__ __launch_bounds__(kAccThreads) void k_fy_accept3(const uint32_t* __restrict__ ring, long long ring_cap,
                                                            int32_t* __restrict__ j, int n, long long* __restrict__ posv,
                                                            int done_slot, unsigned long long* __restrict__ relay, int relay_cap,
                                                            unsigned gen) {
    constexpr int kChunk = kAccThreads * kWpt;
    constexpr int kWaves = kAccThreads / kWave;
    static_assert(kWaves >= 1 && kWaves <= 16, "wave sums are combined inside one DPP row");
    static_assert(kWpt % 4 == 0 && kWpt <= kRingMirror, "draws are fetched four at a time from a mirrored ring");
    __shared__ int s_wsum[2][kWaves];
    __shared__ int s_chg[2][kWaves];
    __shared__ int s_lo[kWaves], s_hi[kWaves];
    __shared__ int s_ref[2];
    __shared__ int s_start, s_fast;
    constexpr int kListCap = 2048;
    __shared__ unsigned long long s_list[kListCap];
    // the chunk's targets, in index order, before they go to memory as whole lines (the dynamic allocation: kChunk words.  Stored
    // straight from the lanes they were 4-byte stores ~90 B apart -- one line per lane and instruction, ~24 k of them per chunk,
    // whose drain the next chunk's first wait then sat behind: 58 k cycles per chunk)
    extern __shared__ int32_t s_stage[];
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const long long cursor0 = posv[1];                // stream offset of the first unread draw
    const long long avail = posv[0];                  // draws written so far (the fill for this shuffle has completed)
    unsigned long long* const done_word = relay + relay_cap;
    auto pack = [&](int i) -> unsigned long long { return ((unsigned long long)gen << 32) | (unsigned long long)(unsigned)i; };
    auto tagged = [&](unsigned long long e) -> bool { return (unsigned)(e >> 32) == gen; };
    auto cload = [](const unsigned long long* p) -> unsigned long long { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto cstore = [](unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
    uint32_t ynext[kWpt];
    long long r_next = (cursor0 + (long long)blockIdx.x * kChunk + (long long)tid * kWpt) % ring_cap;
    const long long r_step = ((long long)G * kChunk) % ring_cap;
    auto fetch = [&]() {             // branch-free: draws past `avail` come from some valid slot and are never looked at
        const long long r0 = r_next;
        r_next += r_step;
        if (r_next >= ring_cap) r_next -= ring_cap;
        const u32x4u* src = reinterpret_cast<const u32x4u*>(ring + r0);
#pragma unroll
        for (int q = 0; q < kWpt / 4; ++q) {
            const u32x4u v = src[q];
            ynext[4 * q + 0] = v.x; ynext[4 * q + 1] = v.y; ynext[4 * q + 2] = v.z; ynext[4 * q + 3] = v.w;
        }
    };
    fetch();
    int par = 0;                                      // exchange-buffer parity (runs on across chunks)
#ifdef AURPPO_ACC_STAMPS   // tools/accept_stamps.py: thread 0 of workgroup 1 (a steady-state member of the relay), cycles per section
    unsigned long long t_cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_readcyclecounter();
    unsigned long long t_chunks = 0, t_fast = 0, t_list = 0, t_ent = 0, t_delta = 0, t_rounds = 0, t_w = 0;
#define TST(k) do { if (tid == 0) { const unsigned long long t__ = __builtin_readcyclecounter(); t_cyc[k] += t__ - t_last; t_last = t__; } } while (0)
#else
#define TST(k) do { } while (0)
#endif
    for (int c = blockIdx.x; c < relay_cap; c += G) {
        const long long cbase = cursor0 + (long long)c * kChunk;
        uint32_t y[kWpt];
        int nhave = 0;                                // my draws that exist (a prefix of the kWpt)
#pragma unroll
        for (int u = 0; u < kWpt; ++u) {
            y[u] = mt_temper(ynext[u]);     // the ring holds the untempered words
            nhave += (cbase + (long long)tid * kWpt + u) < avail ? 1 : 0;
        }
        fetch();
        int stage_top = 0;                            // the chunk's first index (set before anything is emitted)
        // the literal rule for my draws from starting index i0: #accepted; optionally stages targets / #consumed
        auto walk = [&](int i0, bool emit, int& consumed) -> int {
            int i = i0, acc = 0;
            consumed = 0;
#pragma unroll
            for (int u = 0; u < kWpt; ++u) {
                if (u < nhave && i >= 1) {
                    const uint32_t v = y[u] & (0xffffffffu >> __clz(i));
                    ++consumed;
                    if (v <= (uint32_t)i) {
                        if (emit) s_stage[stage_top - i] = (int32_t)v;
                        --i;
                        ++acc;
                    }
                }
            }
            return acc;
        };
        uint32_t dn = 0, up = 0;
        bool sure = false;
        int ref_i0 = 0;
        bool wave_fast = false;
        auto eval = [&](int i0) -> int {          // as in k_fy_accept: #draws <= i0 - kWpt, and the slack of that count
            const bool shape = nhave == kWpt && i0 > kWpt && __clz(i0) == __clz(i0 - kWpt);
            const uint32_t mask = 0xffffffffu >> __clz(i0 | 1);
            const uint32_t lo = (uint32_t)(i0 - kWpt);
            int cc = 0;
            uint32_t d = 0xffffffffu, pp = 0xffffffffu;
#pragma unroll
            for (int u = 0; u < kWpt; ++u) {
                const uint32_t v = y[u] & mask;
                const bool le = v <= lo;
                cc += le ? 1 : 0;
                d = le ? min(d, lo - v) : d;
                pp = le ? pp : min(pp, v - lo - 1u);
            }
            dn = d;
            up = pp;
            sure = shape && pp >= (uint32_t)kWpt;
            ref_i0 = i0;
            wave_fast = __builtin_amdgcn_ballot_w64(!sure) == 0ull;
            if (!wave_fast) {
                int consumed;
                cc = walk(i0, false, consumed);
            }
            return cc;
        };
        auto still_ok = [&](int i0) -> bool {
            const int delta = i0 - ref_i0;
            const bool shape = i0 > kWpt && __clz(i0) == __clz(i0 - kWpt) && __clz(i0) == __clz(ref_i0);
            const bool room = delta >= 0 ? up >= (uint32_t)(kWpt + delta) : dn >= (uint32_t)(-delta);
            return delta == 0 || (sure && shape && room);
        };
        // ---- where to start from: the chunk before mine if it is confirmed already, else a guess from the latest confirmed one
        int i_start = n - 1;
        bool have_start = c == 0;
        int c_ref_kept = c - 1;
        if (c > 0) {
            if (wave == 0) {
                const int cc = c - 1 - lane;
                const unsigned long long e = cc >= 0 ? cload(&relay[cc]) : 0ull;
                const unsigned long long dw = cload(done_word);
                const unsigned long long ok = __builtin_amdgcn_ballot_w64(cc >= 0 && tagged(e));
                int c_ref = -1, i_ref = n - 1;
                if (ok) {
                    const int L = __builtin_ctzll(ok);
                    c_ref = c - 1 - L;
                    i_ref = __builtin_amdgcn_readlane((int)(unsigned)e, L);
                }
                if (tagged(dw)) { c_ref = c - 1; i_ref = 0; }   // the shuffle is over
                if (lane == 0) { s_ref[0] = c_ref; s_ref[1] = i_ref; }
            }
            __syncthreads();
            const int c_ref = s_ref[0], i_ref = s_ref[1];
            __syncthreads();          // (s_ref is rewritten for the next chunk)
            if (i_ref < 1) break;     // the shuffle ended before this chunk
            c_ref_kept = c_ref;
            if (c_ref == c - 1) {
                i_start = i_ref;
                have_start = true;
            } else {
                // expected trajectory: di/dp = -(i + 1) / (mask + 1), octave by octave (the mask halves where i crosses a power of two)
                float fi = (float)(i_ref + 1), between = (float)(c - 1 - c_ref) * (float)kChunk;
                float m1 = (float)(0xffffffffu >> __clz(i_ref | 1)) + 1.0f;
                for (int k = 0; k < 32; ++k) {
                    const float to_edge = m1 * __logf(fi / (0.5f * m1));      // draws until i + 1 reaches the bottom of this octave
                    if (between <= to_edge || m1 <= 2.0f) {
                        fi *= __expf(-between / m1);
                        break;
                    }
                    between -= to_edge;
                    fi = 0.5f * m1;
                    m1 *= 0.5f;
                }
                i_start = (int)fi - 1;
            }
        }
        TST(0);      // draws in registers, next fetch issued, reference for the guess read
        // ---- k_fy_accept's fixed point over the chunk, from starting index i_cur (first call: from the expected trajectory)
        int cur_i0, cnt, incl, total = 0, wbase = 0;
        bool wchg = true;
        auto rounds = [&](int i_cur) {
            for (;;) {
#ifdef AURPPO_ACC_STAMPS
                ++t_rounds;
#endif
                if (lane == kWave - 1) {
                    s_wsum[par][wave] = incl;
                    s_chg[par][wave] = wchg ? 1 : 0;
                }
                __syncthreads();
                const int ws = lane < kWaves ? s_wsum[par][lane] : 0;
                const int wc = lane < kWaves ? s_chg[par][lane] : 0;
                par ^= 1;
                const int pre = row_incl_scan(ws);                          // lanes 0..15: prefix over the wave sums
                total = __builtin_amdgcn_readlane(pre, kWaves - 1);
                wbase = wave > 0 ? __builtin_amdgcn_readlane(pre, wave > 0 ? wave - 1 : 0) : 0;
                const bool any = __builtin_amdgcn_ballot_w64(wc != 0) != 0ull;
                if (!any) break;     // no wave's counts moved in the last evaluation: (cur_i0, cnt) is the fixed point
                cur_i0 = i_cur - (wbase + incl - cnt);
                if (__builtin_amdgcn_ballot_w64(!still_ok(cur_i0)) == 0ull) {
                    wchg = false;
                } else {
                    const int cnt2 = eval(cur_i0);
                    wchg = __builtin_amdgcn_ballot_w64(cnt2 != cnt) != 0ull;
                    cnt = cnt2;
                    incl = wave_incl_scan(cnt);
                }
            }
        };
        {
            // expected index at my first draw (same trajectory, tid * kWpt draws on)
            float fi = (float)(i_start + 1), between = (float)(tid * kWpt);
            float m1 = (float)(0xffffffffu >> __clz(i_start | 1)) + 1.0f;
            for (int k = 0; k < 32; ++k) {
                const float to_edge = m1 * __logf(fi / (0.5f * m1));
                if (between <= to_edge || m1 <= 2.0f) {
                    fi *= __expf(-between / m1);
                    break;
                }
                between -= to_edge;
                fi = 0.5f * m1;
                m1 *= 0.5f;
            }
            cur_i0 = i_start >= 1 ? (int)fi - 1 : i_start;
            cnt = eval(cur_i0);
            incl = wave_incl_scan(cnt);
            rounds(i_start);
        }
        TST(1);      // solved from the guess (or from the confirmed start)
        if (!have_start) {
            // ---- what the true starting index can change: the SENSITIVE draws.  Along the guessed solution draw p is tried against
            // index idx(p); with the chunk starting `c` higher or lower, |c| <= W, it is tried against idx(p) + c.  A draw whose
            // decision is the same over that whole range (same mask octave, value outside [idx - W, idx + W]) cannot change; the
            // others go, in stream order, into a list {raw word, idx(p), decision} in LDS.  Given the true start, the index after
            // the chunk then follows from a walk over the list alone (the literal rule per entry, the offset c carried along) --
            // one wave, no barrier -- which is all the chain has to wait for; the lanes' own counts are redone after the word is out.
            const int sigma_draws = (c - 1 - c_ref_kept) * kChunk + kChunk;
            const int W = min(max((int)(2.0f * __fsqrt_rn((float)sigma_draws)), 64), 4096);   // 4 sigma of the guess (sigma^2 <= draws / 4); beyond it: the slow path
            unsigned long long ent[kWpt / 4];      // (a lane keeps at most kWpt / 4 entries in registers; more: overflow)
            int n_ent = 0;
            {
                int i = cur_i0;
#pragma unroll
                for (int u = 0; u < kWpt; ++u) {
                    const bool live = u < nhave;
                    const bool pos = i >= 1;
                    const uint32_t mask = 0xffffffffu >> __clz(i | 1);
                    const uint32_t v = y[u] & mask;
                    const bool acc = live && pos && v <= (uint32_t)i;
                    const bool same_oct = i - W >= 1 && __clz(i - W) == __clz(i + W);
                    const bool fixed = same_oct && (v + (uint32_t)W <= (uint32_t)i || v > (uint32_t)(i + W));
                    if (live && !fixed) {
                        const unsigned long long e = ((unsigned long long)y[u] << 32) | ((unsigned long long)(unsigned)(i & 0x7fffffff) << 1) | (acc ? 1ull : 0ull);
                        // i may be <= 0 past the end of the shuffle: kept as a 31-bit two's complement
#pragma unroll
                        for (int q = 0; q < kWpt / 4; ++q)
                            if (q == n_ent) ent[q] = e;
                        ++n_ent;
                    }
                    i -= acc ? 1 : 0;
                }
            }
            const int e_incl = wave_incl_scan(n_ent);
            if (lane == kWave - 1) s_lo[wave] = e_incl;
            const bool lane_over = n_ent > kWpt / 4;
            if (__builtin_amdgcn_ballot_w64(lane_over) != 0ull && lane == 0) s_hi[wave] = 1; else if (lane == 0) s_hi[wave] = 0;
            __syncthreads();
            int e_base = 0, e_total = 0, over = 0;
            for (int w = 0; w < kWaves; ++w) {
                const int t = s_lo[w];
                e_base += w < wave ? t : 0;
                e_total += t;
                over |= s_hi[w];
            }
            const bool use_list = !over && e_total <= kListCap;
            if (use_list) {
                const int at = e_base + e_incl - n_ent;
#pragma unroll
                for (int q = 0; q < kWpt / 4; ++q)
                    if (q < n_ent) s_list[at + q] = ent[q];
            }
            __syncthreads();
            TST(2);      // sensitive draws listed
#ifdef AURPPO_ACC_STAMPS
            t_list += use_list ? 1 : 0;
            t_ent += (unsigned long long)e_total;
            t_w += (unsigned long long)W;
#endif
            if (wave == 0) {
                // ---- the relay: my predecessor's word.  From here to the store of my own is the critical path of the whole shuffle
                __builtin_amdgcn_s_setprio(3);
                int i_exact = 0;
                if (lane == 0) {
                    for (;;) {
                        const unsigned long long e = cload(&relay[c - 1]);
                        const unsigned long long dw = cload(done_word);
                        if (tagged(e)) { i_exact = (int)(unsigned)e; break; }
                        if (tagged(dw)) { i_exact = 0; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                i_exact = __builtin_amdgcn_readfirstlane(i_exact);
                TST(3);      // waited for the predecessor
#ifdef AURPPO_ACC_STAMPS
                t_delta += (unsigned long long)abs(i_exact - i_start);
#endif
                int fast = 0;
                if (i_exact >= 1 && use_list) {
                    // walk the list, 64 entries at a time: within a block the offsets are the fixed point of
                    // "c_l = c_in + (decisions flipped by earlier lanes)", iterated to rest (a flip is rare)
                    int cin = i_exact - i_start;
                    bool in_range = abs(cin) <= W;
                    for (int b0 = 0; b0 < e_total && in_range; b0 += kWave) {
                        const bool have = b0 + lane < e_total;
                        const unsigned long long e = have ? s_list[b0 + lane] : 0ull;
                        const uint32_t yy = (uint32_t)(e >> 32);
                        const int i_sp = ((int)(((unsigned)e >> 1) << 1)) >> 1;      // sign-extend the 31-bit index
                        const int a_sp = (int)(e & 1ull);
                        int cl = cin, flip = 0;
                        for (;;) {
                            const int i_now = i_sp + cl;
                            const uint32_t v = yy & (0xffffffffu >> __clz(i_now | 1));
                            const int a_now = (have && i_now >= 1 && v <= (uint32_t)i_now) ? 1 : 0;
                            const int f2 = have ? a_sp - a_now : 0;             // the index after this draw moves by this much
                            const bool moved = f2 != flip;
                            flip = f2;
                            const int incl = wave_incl_scan(flip);
                            cl = cin + incl - flip;
                            if (__builtin_amdgcn_ballot_w64(moved) == 0ull) {
                                const int lo_c = min(cl, cl + flip), hi_c = max(cl, cl + flip);
                                in_range = __builtin_amdgcn_ballot_w64(have && (lo_c < -W || hi_c > W)) == 0ull;
                                cin += __builtin_amdgcn_readlane(incl, kWave - 1);
                                break;
                            }
                        }
                    }
                    if (in_range) {
                        const int i_end = (i_start - total) + cin;
                        const bool last = i_end <= 0;                            // every remaining index gets its target in this chunk
                        const bool dry = !last && cbase + kChunk >= avail;       // ... or the draws run out first
                        if (lane == 0) {
                            cstore(&relay[c], pack((last || dry) ? 0 : i_end));
                            if (last || dry) cstore(done_word, pack(1));
                        }
                        fast = 1;
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                TST(4);      // list walked, word published (fast path)
#ifdef AURPPO_ACC_STAMPS
                t_fast += fast ? 1 : 0;
#endif
                if (lane == 0) {
                    s_start = i_exact;
                    s_fast = fast;
                }
            }
            __syncthreads();
            const int i_exact = s_start;
            const bool fast = s_fast != 0;
            if (i_exact < 1) break;                   // the shuffle ended before this chunk
            if (i_exact != i_start) {
                cur_i0 = i_exact - (wbase + incl - cnt);
                if (__builtin_amdgcn_ballot_w64(!still_ok(cur_i0)) == 0ull) {
                    wchg = false;
                } else {
                    const int cnt2 = eval(cur_i0);
                    wchg = __builtin_amdgcn_ballot_w64(cnt2 != cnt) != 0ull;
                    cnt = cnt2;
                    incl = wave_incl_scan(cnt);
                }
                rounds(i_exact);
            }
            i_start = i_exact;
            have_start = !fast;      // (fast: the word is out already)
        }
        TST(5);      // counts redone from the true start
#ifdef AURPPO_ACC_STAMPS
        ++t_chunks;
#endif
        const bool last = total >= i_start;
        const bool dry = !last && cbase + kChunk >= avail;
        if (have_start && tid == 0) {
            cstore(&relay[c], pack((last || dry) ? 0 : i_start - total));
            if (last || dry) cstore(done_word, pack(1));
        }
        // ---- off the chain: the targets, staged in index order, then stored as whole lines
        int consumed = kWpt;
        stage_top = i_start;
        if (!wave_fast || last) {
            (void)walk(cur_i0, true, consumed);
        } else {
            const uint32_t mask = 0xffffffffu >> __clz(cur_i0 | 1);
            const uint32_t lo = (uint32_t)(cur_i0 - kWpt);
            int i = cur_i0;
#pragma unroll
            for (int u = 0; u < kWpt; ++u) {
                const uint32_t v = y[u] & mask;
                if (v <= lo) s_stage[stage_top - (i--)] = (int32_t)v;
            }
        }
        __syncthreads();
        {
            const int n_out = last ? i_start : total;         // indices i_start .. i_start - n_out + 1
            for (int q = tid; q < n_out; q += kAccThreads) j[i_start - q] = s_stage[q];
        }
        TST(6);      // targets emitted
        if (last) {
            // words consumed = everything up to and including the draw that filled i = 1 (unique thread)
            if (cnt > 0 && cur_i0 - cnt == 0) {
                const long long end = cbase + (long long)tid * kWpt + consumed;
                posv[1] = end;
                posv[done_slot] = end;
            }
            break;
        }
        if (dry) {
            if (tid == 0) {
                const long long end = cbase + (avail - cbase < kChunk ? avail - cbase : kChunk);
                posv[1] = end;
                posv[done_slot] = end;
                posv[2] = 1;      // ran out of draws before the shuffle finished: sticky error
            }
            break;
        }
    }
#ifdef AURPPO_ACC_STAMPS
    if (tid == 0 && blockIdx.x == (G > 1 ? 1 : 0)) {
        for (int q = 0; q < 7; ++q) posv[8 + q] = (long long)t_cyc[q];
        posv[15] = (long long)t_chunks; posv[16] = (long long)t_fast; posv[17] = (long long)t_list; posv[18] = (long long)t_ent;
        posv[19] = (long long)t_delta; posv[20] = (long long)t_rounds; posv[21] = (long long)t_w;
    }
#endif
#undef TST
}

// Side job: clr[0 .. clr_n) = -1 (clr_n a multiple of 4, 16-B aligned) -- the list heads the NEXT shuffle links into.
// (It was a memset between accept and link: 5 us of work that had to find a free CU beside K7, ~100 us in-situ; as
// stores inside the single-workgroup accept kernel it cost that kernel 20-40 us.)
__global__ void k_fy_link(const int32_t* __restrict__ j, int32_t* __restrict__ head, int32_t* __restrict__ next,
                          int n, int32_t* __restrict__ clr, int clr_n, const long long* __restrict__ posv) {
    // grid-stride, four positions per pass: their swaps are in flight together (a memory-side atomic takes ~1 us)
    constexpr int kB = 4;
    const int stride = gridDim.x * blockDim.x;
    {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const i32x4 m1 = {-1, -1, -1, -1};
        i32x4* dst = reinterpret_cast<i32x4*>(clr);
        for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < clr_n / 4; q += stride) dst[q] = m1;
    }
    // a shuffle that ran out of draws (sticky flag) left targets unwritten: nothing may be indexed with them
    if (posv[2] != 0) return;
    for (int s0 = blockIdx.x * blockDim.x + threadIdx.x; s0 < n; s0 += kB * stride) {
        int js[kB], old[kB];
#pragma unroll
        for (int u = 0; u < kB; ++u) {
            const int s = s0 + u * stride;
            js[u] = (s >= 1 && s < n) ? j[s] : s;
        }
#pragma unroll
        for (int u = 0; u < kB; ++u) {
            const int s = s0 + u * stride;
            old[u] = (s >= 1 && s < n && js[u] != s) ? atomicExch(&head[js[u]], s) : 0;
        }
#pragma unroll
        for (int u = 0; u < kB; ++u) {
            const int s = s0 + u * stride;
            if (s >= 1 && s < n && js[u] != s) next[s] = old[u];
        }
    }
}

// out[i] = in[src(i)] (in == nullptr: identity), src(i) = position whose ORIGINAL content ends at i.
__global__ void k_fy_resolve(const int32_t* __restrict__ j, const int32_t* __restrict__ head,
                             const int32_t* __restrict__ next, const int32_t* __restrict__ in,
                             int32_t* __restrict__ out, int n, const long long* __restrict__ posv) {
  // grid-stride: a bounded grid (AURPPO_K2_RESOLVE_WGS) keeps the kernel on the few CUs K7 leaves free instead of queueing
  // thousands of short workgroups behind it
  const bool broken = posv[2] != 0;   // the shuffle ran out of draws: hand back the input order (valid indices; the flag says it is not a shuffle)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (broken) {
        out[i] = in ? in[i] : i;
        continue;
    }
    const int kNone = 0x7fffffff;
    int src;
    int cur = i;
    bool chase = true;
    if (i >= 1) {
        const int ji = j[i];
        if (ji != i) {
            // position ji just before step i holds what the most recent earlier writer (smallest s > i
            // with j_s == ji) put there, else its original content
            int m = kNone;
            for (int p = head[ji]; p >= 0; p = next[p])
                if (p > i && p < m) m = p;
            if (m == kNone) {
                src = ji;
                chase = false;
            } else {
                cur = m;
            }
        }
    }
    if (chase) {
        // content of position cur just before step cur: written by the smallest s > cur with j_s == cur
        for (;;) {
            int m = kNone;
            for (int p = head[cur]; p >= 0; p = next[p])
                if (p < m) m = p;
            if (m == kNone) break;
            cur = m;
        }
        src = cur;
    }
    out[i] = in ? in[src] : src;
  }
}

// sticky "a shuffle ran out of draws" flag as a float, for the trainer's once-per-update scalar read
__global__ void k_rng_status(const long long* __restrict__ posv, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = posv[2] != 0 ? 1.0f : 0.0f;
}

__global__ void k_arange(int32_t* idx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = i;
}

// Expected number of 32-bit draws numpy's masked rejection needs for indices n-1..1
static double expected_draws(int n) {
    double e = 0.0;
    long lo = 1;
    while (lo <= (long)n - 1) {                 // octave [lo, 2lo): mask + 1 = 2lo
        const long hi = (2 * lo - 1 < (long)n - 1) ? 2 * lo - 1 : (long)n - 1;
        e += 2.0 * (double)lo * log(((double)hi + 1.5) / ((double)lo + 0.5));
        lo *= 2;
    }
    return e;
}

// Draws one shuffle of n may need: expectation + 12 sigma (sigma <= sqrt(2n)); exceeding it trips the
// sticky error flag instead of producing a wrong permutation.
static double need_words(int n) { return expected_draws(n) + 12.0 * sqrt(2.0 * (double)n) + 2.0 * kMtN; }

// k_mt_fill and k_fy_accept are single-workgroup latency chains that run side by side for most of an update.  The
// dispatcher is free to put both on the same CU, where the 16 accept waves take four issue slots in five from the
// twist (rocprof in-situ: k_mt_fill 640 us alone, 850-910 us next to the accept kernel).  Each therefore asks for
// more than half a CU's LDS (unused), which no two of them can get together.
#ifndef AURPPO_ACC_WPT
#define AURPPO_ACC_WPT 8      // draws per thread per accept step (A/B knob)
#endif
constexpr int kAccWpt = AURPPO_ACC_WPT;
#ifndef AURPPO_ACC3_WPT
#define AURPPO_ACC3_WPT 16    // draws per thread and chunk of the workgroup relay (k_fy_accept3): 16 384 per chunk (32: the kernel spills)
#endif
constexpr int kAcc3Wpt = AURPPO_ACC3_WPT, kAcc3Chunk = 1024 * kAcc3Wpt;
constexpr size_t kOwnCuLds = 81 * 1024;
static hipError_t own_cu_setup() {
    static bool done_of[kMaxDevices] = {false};
    bool& done = done_of[aurppo_device_slot()];
    if (done) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mt_fill), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kOwnCuLds);

    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fy_accept<1024, kAccWpt>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kOwnCuLds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fy_accept3<1024, kAcc3Wpt>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(int32_t) * kAcc3Chunk));
    done = e == hipSuccess;
    return e;
}

// Diagnostic knob AURPPO_K2_ONE_STREAM=1: the twist and the resolves run on the caller's stream too (everything serial) -- used
// to tell how much of the main stream's slowdown comes from HOW MANY streams are busy rather than from what runs on them.
static hipStream_t fill_stream_of(aurppo_rng* rng, hipStream_t s) {
    return aurppo_knobs().k2_one_stream == 1 ? s : rng->fill_stream;
}

// Inventory the twist is asked to keep: two shuffles' worth.  Diagnostic knob AURPPO_TEST_K2_STARVE=p (tests only): p per cent
// of ONE shuffle's need instead, so that a shuffle runs out of draws -- the sticky-error path no real run reaches (12 sigma).
static long long fill_target(double need) {
    const int starve = aurppo_knobs().k2_starve;
    return (long long)(starve > 0 ? need * (double)starve / 100.0 : 2.0 * need);
}

static int enqueue_fill(aurppo_rng* rng, double need, hipStream_t after, int slot, int cur_slot) {
    const hipStream_t fs = fill_stream_of(rng, after);
    // inventory target 2*need: one shuffle may be consuming while the next one's draws are produced
    AURPPO_HIP_TRY(hipEventRecord(rng->ev_sync, after));
    AURPPO_HIP_TRY(hipStreamWaitEvent(fs, rng->ev_sync, 0));
    const int nblk_max = (int)(2.0 * need / kMtN) + 2;
    AURPPO_HIP_TRY(own_cu_setup());
    hipLaunchKernelGGL(k_mt_fill, dim3(1), dim3(kFillAll), kOwnCuLds, fs, rng->d_last, rng->d_ring,
                       (long long)rng->ring_cap, rng->d_pos, fill_target(need), nblk_max, cur_slot);
    AURPPO_LAUNCH_CHECK("k_mt_fill");
    AURPPO_HIP_TRY(hipEventRecord(rng->ev_fill[slot], fs));
    return AURPPO_OK;
}

constexpr int kClrChunk = 4096;         // head buffers are sized and cleared in whole chunks

int permute_once(aurppo_rng* rng, const int32_t* in, int32_t* out, int n, hipStream_t s) {
    const hipStream_t fs = fill_stream_of(rng, s);
    const double need = need_words(n);
    if (2.0 * need + 4.0 * kMtN > (double)rng->ring_cap) {
        aurppo_set_error("shuffle: word ring too small for n=%d", n);
        return AURPPO_ESHAPE;
    }
    const int slot = (int)(rng->seq & 1);
    const int h = (int)(rng->seq % 3), hn = (int)((rng->seq + 1) % 3);
    if (rng->primed_need < need) {
        // nothing (or too little) in flight for this size: produce this shuffle's draws now
        // (ordered after everything on `s`, so the live cursor in slot 1 is final)
        int rc = enqueue_fill(rng, need, s, slot, 1);
        if (rc != AURPPO_OK) return rc;
        rng->primed_need = need;
    }
    AURPPO_HIP_TRY(hipStreamWaitEvent(s, rng->ev_fill[slot], 0));
    AURPPO_HIP_TRY(own_cu_setup());
    // where link + resolve run: behind the accept on the caller's stream / the fill stream (default), or both on a third
    // stream of the handle, so that the accept chain and the fill chain carry nothing else
    hipStream_t ls = rng->use_post ? rng->post_stream : s;
    if (rng->use_post && rng->seq >= 2)       // this accept overwrites the swap targets shuffle seq-2's resolve read
        AURPPO_HIP_TRY(hipStreamWaitEvent(s, rng->ev_post[slot], 0));
    if (rng->head_clean[h] < n) {
        // first shuffle after a (re)start, or a larger n than the previous call prepared: clear here (rare path)
        AURPPO_HIP_TRY(hipMemsetAsync(rng->d_head[h], 0xff, sizeof(int32_t) * (size_t)n, ls));
        rng->head_clean[h] = n;
    }
    if (aurppo_knobs().k2_accept == 3) {
        // G workgroups, one per CU (each asks for more than half a CU's LDS): as many as K7 leaves free beside the twist, and no
        // more than the shuffle has chunks
        int G = aurppo_knobs().k2_accept3_wgs;
        const int chunks = (int)(need / kAcc3Chunk) + 1;
        if (G > chunks) G = chunks;
        if (G < 1) G = 1;
        rng->acc_gen += 1;
        if (rng->acc_gen == 0) rng->acc_gen = 1;     // (0 is what the buffer is cleared to)
        hipLaunchKernelGGL((k_fy_accept3<1024, kAcc3Wpt>), dim3(G), dim3(1024), sizeof(int32_t) * kAcc3Chunk, s, rng->d_ring, (long long)rng->ring_cap,
                           rng->d_j[slot], n, rng->d_pos, 4 + slot, rng->d_relay, rng->relay_cap, rng->acc_gen);
    } else
        hipLaunchKernelGGL((k_fy_accept<1024, kAccWpt>), dim3(1), dim3(1024), kOwnCuLds, s, rng->d_ring, (long long)rng->ring_cap,
                           rng->d_j[slot], n, rng->d_pos, 4 + slot);
    AURPPO_LAUNCH_CHECK("k_fy_accept");
    AURPPO_HIP_TRY(hipEventRecord(rng->ev_acc[slot], s));
    // the link kernel also clears the heads of the NEXT shuffle (same n assumed; a larger one takes the path above).
    // Their last reader, the resolve of shuffle seq-2, precedes the fill this shuffle's accept waited for.
    const int clr_n = (int)(((size_t)n + kClrChunk - 1) / kClrChunk * kClrChunk);
    rng->head_clean[hn] = n;
    rng->head_clean[h] = 0;      // about to be linked into
    const int grid = (n + 255) / 256;
    // k_fy_link strides over the positions with a bounded grid.  It runs beside K7, whose workgroups fill the register
    // file of every CU but the spare ones, so a few dozen workgroups are all that is ever resident; one workgroup per
    // 256 positions (2048 of them at B = 524 288) cost every K7 launch it overlapped 8-10 us (rocprof,
    // tools/k7_overlap.sh): 2.73 -> 2.68 ms per update with 48.  AURPPO_K2_LINK_WGS overrides (0 = one per 256).
    const int link_wgs = aurppo_knobs().k2_link_wgs;
    if (rng->use_post) AURPPO_HIP_TRY(hipStreamWaitEvent(ls, rng->ev_acc[slot], 0));
    hipLaunchKernelGGL(k_fy_link, dim3(link_wgs > 0 && link_wgs < grid ? link_wgs : grid), dim3(256), 0, ls, rng->d_j[slot],
                       rng->d_head[h], rng->d_next[slot], n, rng->d_head[hn], clr_n, rng->d_pos);
    AURPPO_LAUNCH_CHECK("k_fy_link");
    AURPPO_HIP_TRY(hipEventRecord(rng->ev_link[slot], ls));
    // Fill stream, in this order: the NEXT shuffle's draws (assumed the same size; ordered after the PREVIOUS accept,
    // ev_acc of the other slot, so the cursor it reads is at most one shuffle stale -- what the 2*need target
    // covers), then THIS shuffle's resolve.  The caller's stream is then free for the next accept while the resolve
    // runs: per shuffle the two streams carry {accept, link} and {fill, resolve} instead of {accept, memset, link,
    // resolve} and {fill}.
    if (rng->seq > 0) AURPPO_HIP_TRY(hipStreamWaitEvent(fs, rng->ev_acc[slot ^ 1], 0));
    {
        const int nblk_max = (int)(2.0 * need / kMtN) + 2;
        hipLaunchKernelGGL(k_mt_fill, dim3(1), dim3(kFillAll), kOwnCuLds, fs, rng->d_last, rng->d_ring,
                           (long long)rng->ring_cap, rng->d_pos, fill_target(need), nblk_max,
                           rng->seq > 0 ? 4 + (slot ^ 1) : 1);   // cursor after the PREVIOUS shuffle (done: waited above)
        AURPPO_LAUNCH_CHECK("k_mt_fill");
        AURPPO_HIP_TRY(hipEventRecord(rng->ev_fill[slot ^ 1], fs));
    }
    hipStream_t rs = rng->use_post ? rng->post_stream : fs;
    if (!rng->use_post) AURPPO_HIP_TRY(hipStreamWaitEvent(rs, rng->ev_link[slot], 0));
    // default 256 workgroups striding over the positions (0 = one per 256 positions, 2048 of them at B = 524 288).
    // Round 1 had no slack on the shuffle streams to pay for a slower resolve; with 0.5 ms of it (round 2) a bounded
    // grid is affordable: 64 / 128 / 192 / 256 / 512 workgroups -> main stream +17 % (shuffle-bound) / -1.2 % (slack 0)
    // / -0.9 % (slack 0.17 ms) / -0.6 % (0.28 ms) / -0.6 % (0.37 ms), alternating runs on one box.
    const int resolve_wgs = aurppo_knobs().k2_resolve_wgs;
    hipLaunchKernelGGL(k_fy_resolve, dim3(resolve_wgs > 0 && resolve_wgs < grid ? resolve_wgs : grid), dim3(256), 0, rs,
                       rng->d_j[slot], rng->d_head[h], rng->d_next[slot], in, out, n, rng->d_pos);
    AURPPO_LAUNCH_CHECK("k_fy_resolve");
    AURPPO_HIP_TRY(hipEventRecord(rng->ev_res, rs));
    if (rng->use_post) AURPPO_HIP_TRY(hipEventRecord(rng->ev_post[slot], rs));
    rng->seq += 1;
    return AURPPO_OK;
}

// The caller's stream sees the permutations only after the last resolve (which ran on the fill stream).
static int join_resolves(aurppo_rng* rng, hipStream_t s) {
    AURPPO_HIP_TRY(hipStreamWaitEvent(s, rng->ev_res, 0));
    return AURPPO_OK;
}

// Drain the look-ahead and restart the stream at d_state (after seed / set_state wrote it on `s`).
static int restart_stream(aurppo_rng* rng, hipStream_t s) {
    AURPPO_HIP_TRY(hipStreamSynchronize(rng->fill_stream));
    if (rng->post_stream) AURPPO_HIP_TRY(hipStreamSynchronize(rng->post_stream));
    hipLaunchKernelGGL(k_mt_origin, dim3(1), dim3(kFillThreads), 0, s, rng->d_state, rng->d_last, rng->d_ring,
                       (long long)rng->ring_cap, rng->d_pos);
    AURPPO_LAUNCH_CHECK("k_mt_origin");
    rng->seq = 0;
    rng->primed_need = 0.0;
    for (int k = 0; k < 3; ++k) rng->head_clean[k] = 0;
    return AURPPO_OK;
}

}  // namespace

extern "C" int aurppo_mt19937_create(aurppo_rng** out, uint32_t seed, int max_n, void* stream) {
    AURPPO_REQUIRE(out, AURPPO_EINVAL, "aurppo_mt19937_create: null out");
    AURPPO_REQUIRE(max_n > 0, AURPPO_ESHAPE, "aurppo_mt19937_create: max_n=%d must be positive", max_n);
    aurppo_rng* r = new aurppo_rng();
    memset(r, 0, sizeof(*r));
    r->max_n = max_n;
    // 2*need in flight + one maximal fill + slack, rounded to whole blocks
    const double need = need_words(max_n);
    r->ring_cap = ((size_t)(4.0 * need) / kMtN + 8) * kMtN;
    const size_t nb = sizeof(int32_t) * (size_t)max_n;
    r->head_cap = ((size_t)max_n + kClrChunk - 1) / kClrChunk * kClrChunk;
    hipError_t e = hipMalloc(&r->d_state, sizeof(uint32_t) * (kMtN + 1));
    if (e == hipSuccess) e = hipMalloc(&r->d_last, sizeof(uint32_t) * kMtN);
    if (e == hipSuccess) e = hipMalloc(&r->d_ring, sizeof(uint32_t) * (r->ring_cap + kRingMirror));
    if (e == hipSuccess) e = hipMalloc(&r->d_pos, sizeof(long long) * 32);
    r->relay_cap = (int)(r->ring_cap / kAcc3Chunk) + 4;
    if (e == hipSuccess) e = hipMalloc(&r->d_relay, sizeof(unsigned long long) * (size_t)(r->relay_cap + 1));
    if (e == hipSuccess) e = hipMemset(r->d_relay, 0, sizeof(unsigned long long) * (size_t)(r->relay_cap + 1));
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipMalloc(&r->d_j[k], nb);
        if (e == hipSuccess) e = hipMalloc(&r->d_next[k], nb);
    }
    for (int k = 0; k < 3 && e == hipSuccess; ++k) e = hipMalloc(&r->d_head[k], sizeof(int32_t) * r->head_cap);
    if (e == hipSuccess) e = hipMalloc(&r->d_tmp, nb);
    if (e == hipSuccess) e = hipMalloc(&r->d_meta, sizeof(uint32_t) * (kMtN + 1));
    if (e == hipSuccess) {
        // highest priority: the twist is a chain of LDS round trips that needs few issue slots, but needs them
        // promptly when it shares a CU with MFMA-heavy waves
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e = hipStreamCreateWithPriority(&r->fill_stream, hipStreamNonBlocking, hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&r->post_stream, hipStreamNonBlocking, hi);
        r->use_post = aurppo_knobs().k2_post_stream == 1 ? 1 : 0;
    }
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipEventCreateWithFlags(&r->ev_fill[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_acc[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_link[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_post[k], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_res, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_sync, hipEventDisableTiming);
    if (e != hipSuccess) {
        aurppo_set_error("aurppo_mt19937_create: HIP resource allocation failed: %s", hipGetErrorString(e));
        aurppo_mt19937_destroy(r);
        return AURPPO_EHIP;
    }
    *out = r;
    return aurppo_mt19937_seed(r, seed, stream);
}

extern "C" int aurppo_mt19937_destroy(aurppo_rng* rng) {
    if (!rng) return AURPPO_OK;
    if (rng->fill_stream) (void)hipStreamSynchronize(rng->fill_stream);
    if (rng->post_stream) (void)hipStreamSynchronize(rng->post_stream);
    (void)hipFree(rng->d_state);
    (void)hipFree(rng->d_last);
    (void)hipFree(rng->d_ring);
    (void)hipFree(rng->d_pos);
    (void)hipFree(rng->d_relay);
    for (int k = 0; k < 2; ++k) {
        (void)hipFree(rng->d_j[k]);
        (void)hipFree(rng->d_next[k]);
    }
    for (int k = 0; k < 3; ++k) (void)hipFree(rng->d_head[k]);
    (void)hipFree(rng->d_tmp);
    (void)hipFree(rng->d_meta);
    for (int k = 0; k < 2; ++k) {
        if (rng->ev_fill[k]) (void)hipEventDestroy(rng->ev_fill[k]);
        if (rng->ev_acc[k]) (void)hipEventDestroy(rng->ev_acc[k]);
        if (rng->ev_link[k]) (void)hipEventDestroy(rng->ev_link[k]);
        if (rng->ev_post[k]) (void)hipEventDestroy(rng->ev_post[k]);
    }
    if (rng->ev_res) (void)hipEventDestroy(rng->ev_res);
    if (rng->ev_sync) (void)hipEventDestroy(rng->ev_sync);
    if (rng->fill_stream) (void)hipStreamDestroy(rng->fill_stream);
    if (rng->post_stream) (void)hipStreamDestroy(rng->post_stream);
    delete rng;
    return AURPPO_OK;
}

extern "C" int aurppo_mt19937_seed(aurppo_rng* rng, uint32_t seed, void* stream) {
    AURPPO_REQUIRE(rng, AURPPO_EINVAL, "aurppo_mt19937_seed: null handle");
    AURPPO_HIP_TRY(hipStreamSynchronize(rng->fill_stream));   // nothing may still be appending to the ring
    if (rng->post_stream) AURPPO_HIP_TRY(hipStreamSynchronize(rng->post_stream));
    hipLaunchKernelGGL(k_mt_seed, dim3(1), dim3(64), 0, (hipStream_t)stream, rng->d_state, seed);
    AURPPO_LAUNCH_CHECK("k_mt_seed");
    return restart_stream(rng, (hipStream_t)stream);
}

#ifdef AURPPO_ACC_STAMPS
extern "C" int aurppo_debug_accept_stamps(aurppo_rng* rng, long long* out24) {
    return hipMemcpy(out24, rng->d_pos + 8, sizeof(long long) * 24, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int aurppo_mt19937_get_state(aurppo_rng* rng, uint32_t* key_h, int32_t* pos_h, void* stream) {
    AURPPO_REQUIRE(rng && key_h && pos_h, AURPPO_EINVAL, "aurppo_mt19937_get_state: null pointer");
    hipStream_t s = (hipStream_t)stream;
    AURPPO_HIP_TRY(hipStreamSynchronize(rng->fill_stream));
    if (rng->post_stream) AURPPO_HIP_TRY(hipStreamSynchronize(rng->post_stream));
    hipLaunchKernelGGL(k_state_at_cursor, dim3(1), dim3(kFillThreads), 0, s, rng->d_state, rng->d_ring,
                       (long long)rng->ring_cap, rng->d_pos, reinterpret_cast<uint32_t*>(rng->d_meta));
    AURPPO_LAUNCH_CHECK("k_state_at_cursor");
    uint32_t buf[kMtN + 1];
    long long posv[4];
    AURPPO_HIP_TRY(hipMemcpyAsync(buf, rng->d_meta, sizeof(buf), hipMemcpyDeviceToHost, s));
    AURPPO_HIP_TRY(hipMemcpyAsync(posv, rng->d_pos, sizeof(posv), hipMemcpyDeviceToHost, s));
    AURPPO_HIP_TRY(hipStreamSynchronize(s));
    AURPPO_REQUIRE(posv[2] == 0, AURPPO_EHIP, "a shuffle ran out of draws (generator state is invalid)");
    for (int i = 0; i < kMtN; ++i) key_h[i] = buf[i];
    *pos_h = (int32_t)buf[kMtN];
    return AURPPO_OK;
}

extern "C" int aurppo_mt19937_set_state(aurppo_rng* rng, const uint32_t* key_h, int32_t pos_h, void* stream) {
    AURPPO_REQUIRE(rng && key_h, AURPPO_EINVAL, "aurppo_mt19937_set_state: null pointer");
    AURPPO_REQUIRE(pos_h >= 0 && pos_h <= kMtN, AURPPO_EINVAL, "aurppo_mt19937_set_state: pos=%d out of [0,624]",
                   pos_h);
    hipStream_t s = (hipStream_t)stream;
    uint32_t buf[kMtN + 1];
    for (int i = 0; i < kMtN; ++i) buf[i] = key_h[i];
    buf[kMtN] = (uint32_t)pos_h;
    AURPPO_HIP_TRY(hipStreamSynchronize(rng->fill_stream));
    if (rng->post_stream) AURPPO_HIP_TRY(hipStreamSynchronize(rng->post_stream));
    AURPPO_HIP_TRY(hipStreamSynchronize(s));   // in-flight shuffles still read the old stream
    AURPPO_HIP_TRY(hipMemcpyAsync(rng->d_state, buf, sizeof(buf), hipMemcpyHostToDevice, s));
    AURPPO_HIP_TRY(hipStreamSynchronize(s));   // buf is a stack temporary
    return restart_stream(rng, s);
}

extern "C" int aurppo_mt19937_status_f32(aurppo_rng* rng, float* out, void* stream) {
    AURPPO_REQUIRE(rng && out, AURPPO_EINVAL, "aurppo_mt19937_status_f32: null pointer");
    hipLaunchKernelGGL(k_rng_status, dim3(1), dim3(64), 0, (hipStream_t)stream, rng->d_pos, out);
    AURPPO_LAUNCH_CHECK("k_rng_status");
    return AURPPO_OK;
}

extern "C" int aurppo_arange_i32(int32_t* idx, int n, void* stream) {
    AURPPO_REQUIRE(idx, AURPPO_EINVAL, "aurppo_arange_i32: null pointer");
    AURPPO_REQUIRE(n >= 0, AURPPO_ESHAPE, "aurppo_arange_i32: n=%d negative", n);
    if (n == 0) return AURPPO_OK;
    hipLaunchKernelGGL(k_arange, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, n);
    AURPPO_LAUNCH_CHECK("k_arange");
    return AURPPO_OK;
}

// The shuffle entry points keep host-side state per call (sequence number, multi-buffer slot, the relay's launch tag, what the
// twist has been asked to stock) and work on the handle's own streams: a captured call would replay with a stale tag and stale
// slots -- a wrong permutation with no error flag.  They therefore refuse a capturing stream (aurppo.h says so).
static int refuse_capture(hipStream_t s, const char* who) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return AURPPO_OK;          // (the legacy NULL stream cannot be queried while another stream captures: nothing to refuse)
    }
    AURPPO_REQUIRE(st == hipStreamCaptureStatusNone, AURPPO_EINVAL,
                   "%s: the stream is being captured into a hipGraph; the mt19937 / shuffle entry points must run eagerly", who);
    return AURPPO_OK;
}

extern "C" int aurppo_shuffle_i32(aurppo_rng* rng, int32_t* idx, int n, void* stream) {
    AURPPO_REQUIRE(rng, AURPPO_EINVAL, "aurppo_shuffle_i32: null handle");
    if (int rc = refuse_capture((hipStream_t)stream, "aurppo_shuffle_i32")) return rc;
    if (n == 0) return AURPPO_OK;
    AURPPO_REQUIRE(idx, AURPPO_EINVAL, "aurppo_shuffle_i32: null pointer");
    AURPPO_REQUIRE(n >= 0 && n <= rng->max_n, AURPPO_ESHAPE, "aurppo_shuffle_i32: n=%d outside [0, max_n=%d]", n,
                   rng->max_n);
    if (n <= 1) return AURPPO_OK;  // numpy draws nothing for n <= 1
    hipStream_t s = (hipStream_t)stream;
    int rc = permute_once(rng, idx, rng->d_tmp, n, s);
    if (rc != AURPPO_OK) return rc;
    rc = join_resolves(rng, s);
    if (rc != AURPPO_OK) return rc;
    AURPPO_HIP_TRY(hipMemcpyAsync(idx, rng->d_tmp, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, s));
    return AURPPO_OK;
}

extern "C" int aurppo_shuffle_epochs_i32(aurppo_rng* rng, int32_t* out, int n, int epochs, void* stream) {
    AURPPO_REQUIRE(rng && out, AURPPO_EINVAL, "aurppo_shuffle_epochs_i32: null pointer");
    AURPPO_REQUIRE(n >= 0 && n <= rng->max_n, AURPPO_ESHAPE, "aurppo_shuffle_epochs_i32: n=%d outside [0, max_n=%d]",
                   n, rng->max_n);
    AURPPO_REQUIRE(epochs >= 0, AURPPO_ESHAPE, "aurppo_shuffle_epochs_i32: epochs=%d negative", epochs);
    if (int rc = refuse_capture((hipStream_t)stream, "aurppo_shuffle_epochs_i32")) return rc;
    if (n == 0 || epochs == 0) return AURPPO_OK;
    hipStream_t s = (hipStream_t)stream;
    if (n == 1) {
        AURPPO_HIP_TRY(hipMemsetAsync(out, 0, sizeof(int32_t) * (size_t)epochs, s));
        return AURPPO_OK;
    }
    for (int e = 0; e < epochs; ++e) {
        const int32_t* in = e ? out + (size_t)(e - 1) * n : nullptr;
        int rc = permute_once(rng, in, out + (size_t)e * n, n, s);
        if (rc != AURPPO_OK) return rc;
    }
    return join_resolves(rng, s);
}
