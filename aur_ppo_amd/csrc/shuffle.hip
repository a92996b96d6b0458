// K2: numpy-legacy MT19937 + Fisher-Yates shuffle, bit-exact with np.random.shuffle under
// np.random.seed(s)  (src/ppo.py:182,213-217; src/robot_ppo.py:335-338).
//
// The algorithm lives in numpy (third-party dependency of the reference): init_genrand seeding,
// 624-word twist, tempering, random_interval = masked rejection on one 32-bit draw per trial,
// then `for i = n-1..1: swap(x[i], x[j_i])`.
//
// Design (gfx950).  The swap chain is n dependent memory transactions if done literally, so it is
// split into the part that is inherently a stream and the part that is not:
//   1. k_fy_targets  -- one 640-thread workgroup walks the MT19937 stream: the 624-word twist runs
//      624-wide out of LDS (three dependency phases), and a whole block of draws is turned into
//      accept/reject decisions at once.  Whether draw p is accepted depends on the index i it is
//      tried against, i.e. on how many earlier draws were accepted -- a triangular system that is
//      solved by iterating "i_p = i0 - (#accepts before p)" to its unique fixed point with
//      ballot/popcount prefix sums (converges in 1-2 rounds: a draw is ambiguous only if its value
//      lands within 624 of i).  Output: the swap targets j[1..n).
//   2. k_fy_link / k_fy_resolve -- given j, the final content of every position is found in
//      parallel with no swaps at all.  Step s writes old x[s] into position j_s, so "what sits in
//      position q just before step t" is "what step min{s>t : j_s=q} put there", recursively.
//      Linked lists per target (atomicExch) give those predecessor sets; chains are O(log n) and
//      almost always empty, so each element resolves with a handful of L2-resident loads.
// The generator state (key[624], pos) stays on the device between calls, like numpy's global stream.
#include "common.h"

struct aurppo_rng {
    uint32_t* d_state;  // key[624] then pos
    int32_t* d_j;       // swap targets
    int32_t* d_head;    // list head per target position
    int32_t* d_next;    // list link per step
    int32_t* d_tmp;     // out-of-place result for the in-place API
    int max_n;
};

namespace {

constexpr int kMtN = 624, kMtM = 397, kMtD = kMtN - kMtM;  // 227
constexpr int kFyThreads = 640;                            // 10 waves, one draw per lane

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

__global__ __launch_bounds__(kFyThreads) void k_fy_targets(uint32_t* __restrict__ state, int32_t* __restrict__ j,
                                                           int n) {
    __shared__ uint32_t mt[kMtN];
    __shared__ int s_wcnt[kFyThreads / kWave];
    __shared__ int s_changed[2];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    if (tid < kMtN) mt[tid] = state[tid];
    if (tid == 0) s_changed[0] = s_changed[1] = 0;
    int pos = (int)state[kMtN];
    int i_cur = n - 1;  // next index to draw a target for (uniform across the workgroup)
    __syncthreads();

    while (i_cur >= 1) {
        if (pos >= kMtN) {  // twist: mt[k] = mt[k+397] ^ mix(mt[k], mt[k+1]), 624-wide in three phases
            uint32_t a = 0, b = 0, c = 0;
            if (tid < kMtN) {
                a = mt[tid];
                if (tid < kMtN - 1) b = mt[tid + 1];
                if (tid < kMtD) c = mt[tid + kMtM];
            }
            __syncthreads();
            if (tid < kMtD) mt[tid] = c ^ mt_mix(a, b);
            __syncthreads();
            if (tid >= kMtD && tid < 2 * kMtD) mt[tid] = mt[tid - kMtD] ^ mt_mix(a, b);
            __syncthreads();
            if (tid >= 2 * kMtD && tid < kMtN - 1) mt[tid] = mt[tid - kMtD] ^ mt_mix(a, b);
            if (tid == kMtN - 1) mt[tid] = mt[kMtM - 1] ^ mt_mix(a, mt[0]);
            __syncthreads();
            pos = 0;
        }
        const bool have = (pos + tid) < kMtN;
        const uint32_t y = have ? mt_temper(mt[pos + tid]) : 0u;
        // fixed point of: acc_p = [ (y_p & mask(i_p)) <= i_p ],  i_p = i_cur - #{q<p : acc_q}
        auto decide = [&](int i_mine, uint32_t& v_out) -> bool {
            if (!(have && i_mine >= 1)) return false;
            uint32_t mask = (uint32_t)i_mine;
            mask |= mask >> 1;
            mask |= mask >> 2;
            mask |= mask >> 4;
            mask |= mask >> 8;
            mask |= mask >> 16;
            v_out = y & mask;
            return v_out <= (uint32_t)i_mine;
        };
        int excl = tid;  // first guess: every earlier draw accepted
        uint32_t v = 0;
        int my_i = i_cur - excl;
        bool acc = decide(my_i, v);
        int total = 0;
        for (int it = 0;; ++it) {
            const unsigned long long bal = __ballot(acc);
            if (lane == 0) s_wcnt[wave] = __popcll(bal);
            if (tid == 0) s_changed[(it + 1) & 1] = 0;
            __syncthreads();
            int base = 0;
            total = 0;
#pragma unroll
            for (int w = 0; w < kFyThreads / kWave; ++w) {  // independent LDS reads, no serial chain
                const int c = s_wcnt[w];
                total += c;
                base += (w < wave) ? c : 0;
            }
            excl = base + __popcll(bal & ((1ull << lane) - 1ull));
            // re-decide under the prefix these decisions imply; if nobody flips, they are the solution
            my_i = i_cur - excl;
            const bool acc2 = decide(my_i, v);
            if (acc2 != acc) s_changed[it & 1] = 1;
            acc = acc2;
            __syncthreads();
            if (!s_changed[it & 1]) break;
        }
        if (acc) j[my_i] = (int32_t)v;
        int consumed = kMtN - pos;
        if (total >= i_cur) {  // the shuffle ends inside this block: find the draw that filled i = 1
            if (acc && my_i == 1) s_last = tid + 1;
            __syncthreads();
            consumed = s_last;
        }
        i_cur -= total;
        pos += consumed;
        __syncthreads();  // s_wcnt / s_last / mt reads complete before the next round rewrites them
    }
    if (tid < kMtN) state[tid] = mt[tid];
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

int permute_once(aurppo_rng* rng, const int32_t* in, int32_t* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_fy_targets, dim3(1), dim3(kFyThreads), 0, s, rng->d_state, rng->d_j, n);
    AURPPO_LAUNCH_CHECK("k_fy_targets");
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
    const size_t nb = sizeof(int32_t) * (size_t)max_n;
    hipError_t e = hipMalloc(&r->d_state, sizeof(uint32_t) * (kMtN + 1));
    if (e == hipSuccess) e = hipMalloc(&r->d_j, nb);
    if (e == hipSuccess) e = hipMalloc(&r->d_head, nb);
    if (e == hipSuccess) e = hipMalloc(&r->d_next, nb);
    if (e == hipSuccess) e = hipMalloc(&r->d_tmp, nb);
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
    delete rng;
    return AURPPO_OK;
}

extern "C" int aurppo_mt19937_seed(aurppo_rng* rng, uint32_t seed, void* stream) {
    AURPPO_REQUIRE(rng, AURPPO_EINVAL, "aurppo_mt19937_seed: null handle");
    hipLaunchKernelGGL(k_mt_seed, dim3(1), dim3(64), 0, (hipStream_t)stream, rng->d_state, seed);
    AURPPO_LAUNCH_CHECK("k_mt_seed");
    return AURPPO_OK;
}

extern "C" int aurppo_mt19937_get_state(aurppo_rng* rng, uint32_t* key_h, int32_t* pos_h, void* stream) {
    AURPPO_REQUIRE(rng && key_h && pos_h, AURPPO_EINVAL, "aurppo_mt19937_get_state: null pointer");
    uint32_t buf[kMtN + 1];
    AURPPO_HIP_TRY(hipMemcpyAsync(buf, rng->d_state, sizeof(buf), hipMemcpyDeviceToHost, (hipStream_t)stream));
    AURPPO_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < kMtN; ++i) key_h[i] = buf[i];
    *pos_h = (int32_t)buf[kMtN];
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
