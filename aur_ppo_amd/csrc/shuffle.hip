// K2: numpy-legacy MT19937 + Fisher-Yates shuffle, bit-exact with np.random.shuffle under
// np.random.seed(s)  (src/ppo.py:182,213-217; src/robot_ppo.py:335-338).
//
// The algorithm lives in numpy (third-party dependency of the reference): init_genrand seeding,
// 624-word twist, tempering, random_interval = masked rejection on one 32-bit draw per trial,
// then `for i = n-1..1: swap(x[i], x[j_i])`.
//
// Design (gfx950).  The swap chain is n dependent memory transactions if done literally, so it is
// split into the part that is inherently a stream and the part that is not:
//   1a. k_mt_fill -- one 256-thread workgroup runs the twist only: 624-word blocks ping-pong between
//      two LDS buffers (3 barriers per block, the recurrence's three dependency phases) and the
//      tempered words stream out to a word buffer in HBM.
//   1b. k_fy_accept -- one 1024-thread workgroup turns 8192 draws per step into accept/reject
//      decisions.  Whether draw p is accepted depends on the index i it is tried against, i.e. on how
//      many earlier draws were accepted -- a triangular system.  Each thread resolves its own 8
//      consecutive draws exactly given its starting index; the starting indices are the fixed point of
//      "i_t = i0 - (#accepts of earlier threads)", iterated with ballot-free prefix sums until no
//      count changes (2-3 rounds: a draw is ambiguous only if its value lands within the error of the
//      guess, and the first guess uses the acceptance rate of the previous step).  ~90 sequential
//      steps per 524288-element shuffle instead of ~1170.  Output: the swap targets j[1..n).
//   2. k_fy_link / k_fy_resolve -- given j, the final content of every position is found in
//      parallel with no swaps at all.  Step s writes old x[s] into position j_s, so "what sits in
//      position q just before step t" is "what step min{s>t : j_s=q} put there", recursively.
//      Linked lists per target (atomicExch) give those predecessor sets; chains are O(log n) and
//      almost always empty, so each element resolves with a handful of L2-resident loads.
// The generator state (key[624], pos) stays on the device between calls, like numpy's global stream.
#include <math.h>

#include "common.h"

struct aurppo_rng {
    uint32_t* d_state;  // key[624] then pos
    int32_t* d_j;       // swap targets
    int32_t* d_head;    // list head per target position
    int32_t* d_next;    // list link per step
    int32_t* d_tmp;     // out-of-place result for the in-place API
    uint32_t* d_words;  // tempered draws for the shuffle in flight
    int32_t* d_meta;    // [0] words consumed so far, [1] next index i to fill, [2] sticky error, [3] words in buffer
    int max_n;
    size_t words_cap;
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

__device__ __forceinline__ uint32_t mt_untemper(uint32_t y) {
    y ^= y >> 18;
    y ^= (y << 15) & 0xefc60000u;
    uint32_t t = y;                       // invert y ^= (y << 7) & 0x9d2c5680
    for (int k = 0; k < 4; ++k) t = y ^ ((t << 7) & 0x9d2c5680u);
    y = t;
    t = y;                                // invert y ^= y >> 11
    t = y ^ (t >> 11);
    t = y ^ (t >> 11);
    return t;
}

constexpr int kFillThreads = 256;

// Tempered words of the stream from the generator's current position on:
//   words[0 .. 624-pos) = temper(state[pos ..]), then `nblk` freshly twisted blocks.  The generator state
// itself is NOT advanced here; k_fy_commit re-derives it from the words actually consumed.
__global__ __launch_bounds__(kFillThreads) void k_mt_fill(const uint32_t* __restrict__ state,
                                                          uint32_t* __restrict__ words, int nblk,
                                                          int32_t* __restrict__ meta, int first) {
    __shared__ uint32_t buf[2][kMtN];
    const int tid = threadIdx.x;
    if (!first && meta[1] < 1) return;   // continuation launch, nothing left to draw
    for (int k = tid; k < kMtN; k += kFillThreads) buf[0][k] = state[k];
    const int pos = (int)state[kMtN];
    __syncthreads();
    size_t out = first ? 0 : (size_t)meta[3];
    if (first) {
        for (int k = pos + tid; k < kMtN; k += kFillThreads) words[k - pos] = mt_temper(buf[0][k]);
        out = (size_t)(kMtN - pos);
    } else {
        // continuation: resume from the last block written (its untempered form is rebuilt from the words)
        for (int k = tid; k < kMtN; k += kFillThreads) buf[0][k] = mt_untemper(words[out - kMtN + k]);
        __syncthreads();
    }
    int cur = 0;
    for (int b = 0; b < nblk; ++b) {
        const uint32_t* o = buf[cur];
        uint32_t* w = buf[cur ^ 1];
        // phase 1: k in [0, 227)
        if (tid < kMtD) w[tid] = o[tid + kMtM] ^ mt_mix(o[tid], o[tid + 1]);
        __syncthreads();
        // phase 2: k in [227, 454)
        if (tid < kMtD) {
            const int k = tid + kMtD;
            w[k] = w[k - kMtD] ^ mt_mix(o[k], o[k + 1]);
        }
        __syncthreads();
        // phase 3: k in [454, 624)
        if (tid < kMtN - 2 * kMtD) {
            const int k = tid + 2 * kMtD;
            w[k] = w[k - kMtD] ^ mt_mix(o[k], k == kMtN - 1 ? w[0] : o[k + 1]);
        }
        __syncthreads();
        for (int k = tid; k < kMtN; k += kFillThreads) words[out + k] = mt_temper(w[k]);
        out += kMtN;
        cur ^= 1;
    }
    if (tid == 0) meta[3] = (int32_t)out;
}

constexpr int kAccThreads = 1024;
constexpr int kWpt = 8;                       // draws per thread per step
constexpr int kAccStep = kAccThreads * kWpt;  // 8192 draws per step

__global__ __launch_bounds__(kAccThreads) void k_fy_accept(const uint32_t* __restrict__ words,
                                                           int32_t* __restrict__ j, int n,
                                                           int32_t* __restrict__ meta, int first) {
    __shared__ int s_wsum[kAccThreads / kWave];
    __shared__ int s_changed[2];
    __shared__ int s_end;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    if (first) {
        if (tid == 0) {
            meta[0] = 0;
            meta[1] = n - 1;
            meta[2] = 0;
        }
    }
    if (tid == 0) s_changed[0] = s_changed[1] = 0;
    __syncthreads();
    long cursor = first ? 0 : meta[0];        // words consumed so far
    int i_cur = first ? n - 1 : meta[1];      // next index to draw a target for
    const long avail = meta[3];
    float rate = 0.72f;                        // acceptance rate guess, refreshed every step
    // this step's draws are fetched one step ahead (every step but the last consumes exactly kAccStep)
    uint32_t ynext[kWpt];
    auto fetch = [&](long cur) {
        const long b0 = cur + (long)tid * kWpt;
#pragma unroll
        for (int u = 0; u < kWpt; ++u) ynext[u] = (b0 + u) < avail ? words[b0 + u] : 0u;
    };
    fetch(cursor);
    while (i_cur >= 1 && cursor < avail) {
        const long base = cursor + (long)tid * kWpt;
        uint32_t y[kWpt];
        int nhave = 0;                             // my draws that exist (a prefix of the 8)
#pragma unroll
        for (int u = 0; u < kWpt; ++u) {
            y[u] = ynext[u];
            nhave += (base + u) < avail ? 1 : 0;
        }
        fetch(cursor + kAccStep);
        // My draws resolved exactly from a starting index i0; returns #accepted (and #consumed).
        // Fast path: with i0 > 8 in one mask octave, a draw v is accepted iff v <= i, and i only moves
        // inside (i0-8, i0]; if no draw lands in that window, "v <= i0" decides all eight at once.
        auto run = [&](int i0, bool emit, int& consumed) -> int {
            consumed = 0;
            if (i0 < 1) return 0;
            if (nhave == kWpt && i0 > kWpt && __clz(i0) == __clz(i0 - kWpt)) {
                const uint32_t mask = 0xffffffffu >> __clz(i0);
                const uint32_t lo = (uint32_t)(i0 - kWpt);
                int acc = 0;
                bool fragile = false;
#pragma unroll
                for (int u = 0; u < kWpt; ++u) {
                    const uint32_t v = y[u] & mask;
                    fragile |= (v > lo) && (v <= (uint32_t)i0);
                    acc += v <= lo ? 1 : 0;
                }
                if (!fragile) {
                    consumed = kWpt;
                    if (emit) {
                        int i = i0;
#pragma unroll
                        for (int u = 0; u < kWpt; ++u) {
                            const uint32_t v = y[u] & mask;
                            if (v <= lo) j[i--] = (int32_t)v;
                        }
                    }
                    return acc;
                }
            }
            int i = i0, acc = 0;
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
        int excl = (int)(rate * (float)(tid * kWpt));   // first guess of #accepts before my draws
        int consumed = 0;
        int cnt = run(i_cur - excl, false, consumed);
        int total = 0;
        for (int it = 0;; ++it) {
            // block-wide exclusive prefix sum of cnt
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const int t = __shfl_up(incl, off, kWave);
                if (lane >= off) incl += t;
            }
            if (lane == kWave - 1) s_wsum[wave] = incl;
            if (tid == 0) s_changed[(it + 1) & 1] = 0;
            __syncthreads();
            int wbase = 0;
            total = 0;
#pragma unroll
            for (int w = 0; w < kAccThreads / kWave; ++w) {
                const int c = s_wsum[w];
                total += c;
                wbase += (w < wave) ? c : 0;
            }
            const int new_excl = wbase + incl - cnt;
            const int cnt2 = run(i_cur - new_excl, false, consumed);
            // (new_excl, cnt2) is the solution as soon as no thread's count moved: prefix(cnt2) == new_excl
            if (cnt2 != cnt) s_changed[it & 1] = 1;
            excl = new_excl;
            cnt = cnt2;
            __syncthreads();
            if (!s_changed[it & 1]) break;
        }
        (void)run(i_cur - excl, true, consumed);   // decisions are final: emit the targets
        // words consumed this step: everything up to and including the draw that filled i = 1
        const int my_end_i = i_cur - excl - cnt;    // index after my draws
        if (tid == 0) s_end = -1;
        __syncthreads();
        if (cnt > 0 && my_end_i == 0) s_end = tid * kWpt + consumed;   // unique thread: filled i = 1
        __syncthreads();
        long step_words = avail - cursor < kAccStep ? avail - cursor : kAccStep;
        if (total >= i_cur) step_words = s_end;
        rate = step_words > 0 ? (float)total / (float)step_words : rate;
        i_cur -= total;
        cursor += step_words;
        __syncthreads();
    }
    if (tid == 0) {
        meta[0] = (int32_t)cursor;
        meta[1] = i_cur;
    }
}

// After a shuffle: move the generator to where numpy's would be -- the 624-word block holding the
// read cursor (untempered back from the word buffer) and the offset inside it; flag an exhausted buffer.
__global__ __launch_bounds__(kFillThreads) void k_fy_commit(uint32_t* __restrict__ state,
                                                            const uint32_t* __restrict__ words,
                                                            int32_t* __restrict__ meta) {
    const int tid = threadIdx.x;
    const int pos0 = (int)state[kMtN];
    const long consumed = meta[0];
    if (meta[1] >= 1) {           // ran out of words before the shuffle finished (astronomically unlikely)
        if (tid == 0) meta[2] = 1;
        return;
    }
    // stream position measured from the start of the generator's current block
    const long p = (long)pos0 + consumed;
    long blk = p / kMtN;
    int pos = (int)(p % kMtN);
    if (pos == 0 && p > 0) {      // numpy sits at pos = 624 of the previous block until the next draw
        blk -= 1;
        pos = kMtN;
    }
    __syncthreads();
    if (blk > 0) {
        // words[] starts at offset pos0 of block 0, so block b >= 1 begins at word (b*624 - pos0)
        const long w0 = blk * kMtN - pos0;
        for (int k = tid; k < kMtN; k += kFillThreads) state[k] = mt_untemper(words[w0 + k]);
    }
    if (tid == 0) state[kMtN] = (uint32_t)pos;
}

__global__ void k_fy_link(const int32_t* __restrict__ j, int32_t* __restrict__ head, int32_t* __restrict__ next,
                          int n) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < 1 || s >= n) return;
    const int js = j[s];
    if (js != s) next[s] = atomicExch(&head[js], s);
}

// out[i] = in[src(i)] (in == nullptr: identity), src(i) = position whose ORIGINAL content ends at i.
__global__ void k_fy_resolve(const int32_t* __restrict__ j, const int32_t* __restrict__ head,
                             const int32_t* __restrict__ next, const int32_t* __restrict__ in,
                             int32_t* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
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
        // sum_{i=lo..hi} 2lo / (i+1)  ~  2lo * ln((hi+1.5)/(lo+0.5))
        e += 2.0 * (double)lo * log(((double)hi + 1.5) / ((double)lo + 0.5));
        lo *= 2;
    }
    return e;
}

int permute_once(aurppo_rng* rng, const int32_t* in, int32_t* out, int n, hipStream_t s) {
    // stage 1 covers the expected draw count + 12 sigma (sigma <= sqrt(2n)); stage 2 (early-exits on the
    // device when stage 1 finished) doubles the margin; beyond that the sticky error flag is raised.
    const double e = expected_draws(n);
    const double sigma = sqrt(2.0 * (double)n);
    int nblk1 = (int)((e + 12.0 * sigma) / kMtN) + 2;
    int nblk2 = (int)((0.05 * e + 40.0 * sigma) / kMtN) + 2;
    const size_t need = (size_t)(nblk1 + nblk2 + 1) * kMtN;
    if (need > rng->words_cap) {
        aurppo_set_error("shuffle: word buffer too small for n=%d (%zu > %zu)", n, need, rng->words_cap);
        return AURPPO_ESHAPE;
    }
    for (int stage = 0; stage < 2; ++stage) {
        hipLaunchKernelGGL(k_mt_fill, dim3(1), dim3(kFillThreads), 0, s, rng->d_state, rng->d_words,
                           stage ? nblk2 : nblk1, rng->d_meta, stage == 0);
        AURPPO_LAUNCH_CHECK("k_mt_fill");
        hipLaunchKernelGGL(k_fy_accept, dim3(1), dim3(kAccThreads), 0, s, rng->d_words, rng->d_j, n, rng->d_meta,
                           stage == 0);
        AURPPO_LAUNCH_CHECK("k_fy_accept");
    }
    hipLaunchKernelGGL(k_fy_commit, dim3(1), dim3(kFillThreads), 0, s, rng->d_state, rng->d_words, rng->d_meta);
    AURPPO_LAUNCH_CHECK("k_fy_commit");
    AURPPO_HIP_TRY(hipMemsetAsync(rng->d_head, 0xff, sizeof(int32_t) * (size_t)n, s));
    const int grid = (n + 255) / 256;
    hipLaunchKernelGGL(k_fy_link, dim3(grid), dim3(256), 0, s, rng->d_j, rng->d_head, rng->d_next, n);
    AURPPO_LAUNCH_CHECK("k_fy_link");
    hipLaunchKernelGGL(k_fy_resolve, dim3(grid), dim3(256), 0, s, rng->d_j, rng->d_head, rng->d_next, in, out, n);
    AURPPO_LAUNCH_CHECK("k_fy_resolve");
    return AURPPO_OK;
}

}  // namespace

extern "C" int aurppo_mt19937_create(aurppo_rng** out, uint32_t seed, int max_n, void* stream) {
    AURPPO_REQUIRE(out, AURPPO_EINVAL, "aurppo_mt19937_create: null out");
    AURPPO_REQUIRE(max_n > 0, AURPPO_ESHAPE, "aurppo_mt19937_create: max_n=%d must be positive", max_n);
    aurppo_rng* r = new aurppo_rng();
    r->max_n = max_n;
    r->d_state = nullptr;
    r->d_j = r->d_head = r->d_next = r->d_tmp = nullptr;
    r->d_words = nullptr;
    r->d_meta = nullptr;
    // worst-case expected draws are < 2n; see permute_once for the per-call sizing
    r->words_cap = (size_t)(2.2 * (double)max_n + 80.0 * sqrt(2.0 * (double)max_n)) + 16 * kMtN;
    const size_t nb = sizeof(int32_t) * (size_t)max_n;
    hipError_t e = hipMalloc(&r->d_state, sizeof(uint32_t) * (kMtN + 1));
    if (e == hipSuccess) e = hipMalloc(&r->d_j, nb);
    if (e == hipSuccess) e = hipMalloc(&r->d_head, nb);
    if (e == hipSuccess) e = hipMalloc(&r->d_next, nb);
    if (e == hipSuccess) e = hipMalloc(&r->d_tmp, nb);
    if (e == hipSuccess) e = hipMalloc(&r->d_words, sizeof(uint32_t) * r->words_cap);
    if (e == hipSuccess) e = hipMalloc(&r->d_meta, sizeof(int32_t) * 8);
    if (e == hipSuccess) e = hipMemsetAsync(r->d_meta, 0, sizeof(int32_t) * 8, (hipStream_t)stream);
    if (e != hipSuccess) {
        aurppo_set_error("aurppo_mt19937_create: hipMalloc failed: %s", hipGetErrorString(e));
        aurppo_mt19937_destroy(r);
        return AURPPO_EHIP;
    }
    *out = r;
    return aurppo_mt19937_seed(r, seed, stream);
}

extern "C" int aurppo_mt19937_destroy(aurppo_rng* rng) {
    if (!rng) return AURPPO_OK;
    (void)hipFree(rng->d_state);
    (void)hipFree(rng->d_j);
    (void)hipFree(rng->d_head);
    (void)hipFree(rng->d_next);
    (void)hipFree(rng->d_tmp);
    (void)hipFree(rng->d_words);
    (void)hipFree(rng->d_meta);
    delete rng;
    return AURPPO_OK;
}

extern "C" int aurppo_mt19937_seed(aurppo_rng* rng, uint32_t seed, void* stream) {
    AURPPO_REQUIRE(rng, AURPPO_EINVAL, "aurppo_mt19937_seed: null handle");
    hipLaunchKernelGGL(k_mt_seed, dim3(1), dim3(64), 0, (hipStream_t)stream, rng->d_state, seed);
    AURPPO_LAUNCH_CHECK("k_mt_seed");
    AURPPO_HIP_TRY(hipMemsetAsync(rng->d_meta, 0, sizeof(int32_t) * 8, (hipStream_t)stream));
    return AURPPO_OK;
}

extern "C" int aurppo_mt19937_get_state(aurppo_rng* rng, uint32_t* key_h, int32_t* pos_h, void* stream) {
    AURPPO_REQUIRE(rng && key_h && pos_h, AURPPO_EINVAL, "aurppo_mt19937_get_state: null pointer");
    uint32_t buf[kMtN + 1];
    AURPPO_HIP_TRY(hipMemcpyAsync(buf, rng->d_state, sizeof(buf), hipMemcpyDeviceToHost, (hipStream_t)stream));
    AURPPO_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < kMtN; ++i) key_h[i] = buf[i];
    *pos_h = (int32_t)buf[kMtN];
    int32_t meta[4];
    AURPPO_HIP_TRY(hipMemcpy(meta, rng->d_meta, sizeof(meta), hipMemcpyDeviceToHost));
    AURPPO_REQUIRE(meta[2] == 0, AURPPO_EHIP, "a shuffle exhausted its word buffer (generator state is invalid)");
    return AURPPO_OK;
}

extern "C" int aurppo_mt19937_set_state(aurppo_rng* rng, const uint32_t* key_h, int32_t pos_h, void* stream) {
    AURPPO_REQUIRE(rng && key_h, AURPPO_EINVAL, "aurppo_mt19937_set_state: null pointer");
    AURPPO_REQUIRE(pos_h >= 0 && pos_h <= kMtN, AURPPO_EINVAL, "aurppo_mt19937_set_state: pos=%d out of [0,624]",
                   pos_h);
    uint32_t buf[kMtN + 1];
    for (int i = 0; i < kMtN; ++i) buf[i] = key_h[i];
    buf[kMtN] = (uint32_t)pos_h;
    AURPPO_HIP_TRY(hipMemcpyAsync(rng->d_state, buf, sizeof(buf), hipMemcpyHostToDevice, (hipStream_t)stream));
    AURPPO_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));  // buf is a stack temporary
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

extern "C" int aurppo_shuffle_i32(aurppo_rng* rng, int32_t* idx, int n, void* stream) {
    AURPPO_REQUIRE(rng, AURPPO_EINVAL, "aurppo_shuffle_i32: null handle");
    if (n == 0) return AURPPO_OK;
    AURPPO_REQUIRE(idx, AURPPO_EINVAL, "aurppo_shuffle_i32: null pointer");
    AURPPO_REQUIRE(n >= 0 && n <= rng->max_n, AURPPO_ESHAPE, "aurppo_shuffle_i32: n=%d outside [0, max_n=%d]", n,
                   rng->max_n);
    if (n <= 1) return AURPPO_OK;  // numpy draws nothing for n <= 1
    hipStream_t s = (hipStream_t)stream;
    int rc = permute_once(rng, idx, rng->d_tmp, n, s);
    if (rc != AURPPO_OK) return rc;
    AURPPO_HIP_TRY(hipMemcpyAsync(idx, rng->d_tmp, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, s));
    return AURPPO_OK;
}

extern "C" int aurppo_shuffle_epochs_i32(aurppo_rng* rng, int32_t* out, int n, int epochs, void* stream) {
    AURPPO_REQUIRE(rng && out, AURPPO_EINVAL, "aurppo_shuffle_epochs_i32: null pointer");
    AURPPO_REQUIRE(n >= 0 && n <= rng->max_n, AURPPO_ESHAPE, "aurppo_shuffle_epochs_i32: n=%d outside [0, max_n=%d]",
                   n, rng->max_n);
    AURPPO_REQUIRE(epochs >= 0, AURPPO_ESHAPE, "aurppo_shuffle_epochs_i32: epochs=%d negative", epochs);
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
    return AURPPO_OK;
}
