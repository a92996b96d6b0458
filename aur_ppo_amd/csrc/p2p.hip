// One-shot all-reduce of the flat gradient bucket over peer memory (SURVEY 8e's plan B: "for <= 1 MB prefer one-shot / direct
// all-reduce over ring"; no reference counterpart -- upstream is single-process, src/ppo.py:266-269 is where the exchange sits).
//
// The bucket of the MLP policy is 68 KB and an update makes 16 exchanges of it: every one is pure latency.  A ring or tree
// collective pays a launch plus several dependent hops over xGMI; here every rank PUBLISHES its gradient in a buffer of its own
// that every peer has mapped (hipIpcOpenMemHandle), raises a flag, and READS the W - 1 peers' buffers directly -- one hop, all
// seven links of a GPU in use at once, no collective library on the path -- and sums the W vectors in RANK ORDER, so every rank
// forms the same bits without a broadcast.
//
//   one launch (k_p2p_allreduce, ~70 workgroups x 256 threads, one element per thread):
//     1. own element -> own exchange buffer, slot (seq & 1), system-scope (write-through) store
//     2. every storing wave drains its stores; the workgroup's lane 0 makes a system-scope release and takes a ticket; the
//        workgroup whose ticket is the last stores flag[slot] = seq (system scope): everything this rank published is in memory
//     3. lanes 0 .. W-1 of every workgroup poll one peer's flag[slot] each for == seq (system-scope loads, s_sleep between
//        polls, bounded by a wall-clock timeout that raises a sticky error word instead of hanging the grid)
//     4. the element of every rank in rank order (own value from the register, peers' through system-scope loads), summed,
//        x 1/W, stored into the local gradient; per-workgroup partial sums of squares for the clip that follows.
//   seq is Adam's step count of the minibatch (a device scalar every rank advances identically): it survives hipGraph replay and
//   needs no host involvement.  Two slots: a rank can be at most one exchange ahead of a peer still reading (it cannot pass
//   step 3 of exchange s + 1 before that peer has raised ITS flag for s + 1, i.e. has finished reading exchange s).
//
// Only workgroups of OTHER ranks are ever waited for, and their flag does not depend on anything this rank does after step 2:
// no circular wait.  The grid is small (<= 128 workgroups) so that two ranks sharing one GPU (the test set-up) are co-resident.
#include <stdio.h>
#include <string.h>

#include "common.h"

namespace {
constexpr int kP2PMaxWorld = 16;
constexpr int kP2PThreads = 256;
constexpr int kFlagStride = 32;            // unsigned words: every flag / counter on a 128-B line of its own
// tail of the allocation, after the two slots: flag[2], ticket, error (one line each)
constexpr int kTailWords = 4 * kFlagStride;

struct P2PView {
    int rank, world, n_pad;
    float* base[kP2PMaxWorld];             // each rank's allocation as mapped into THIS process ([rank] = the local one)
};

__device__ __forceinline__ unsigned long long p2p_wall_clock() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz

__global__ __launch_bounds__(kP2PThreads) void k_p2p_allreduce(float* __restrict__ grads, int n, P2PView v,
                                                               const float* __restrict__ step_dev, float inv_w,
                                                               double* __restrict__ sq_part, unsigned long long timeout_ticks) {
    const unsigned seq = (unsigned)(*step_dev);
    const int slot = (int)(seq & 1u);
    float* const mine = v.base[v.rank] + (size_t)slot * v.n_pad;
    unsigned* const tail = reinterpret_cast<unsigned*>(v.base[v.rank] + 2 * (size_t)v.n_pad);
    unsigned* const my_flag = tail + slot * kFlagStride;
    unsigned* const ticket = tail + 2 * kFlagStride;
    unsigned* const err = tail + 3 * kFlagStride;
    const int i = blockIdx.x * kP2PThreads + threadIdx.x;
    float own = 0.0f;
    if (i < n) {
        own = grads[i];
        __hip_atomic_store(mine + i, own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave: its stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // for the next launch
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(my_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if ((int)threadIdx.x < v.world && (int)threadIdx.x != v.rank) {
        const unsigned* f = reinterpret_cast<const unsigned*>(v.base[threadIdx.x] + 2 * (size_t)v.n_pad) + slot * kFlagStride;
        const unsigned long long t0 = p2p_wall_clock();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(8);
            if (p2p_wall_clock() - t0 > timeout_ticks) {       // a peer that never arrives: say so and let the grid drain
                __hip_atomic_store(err, 1u + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    float t = 0.0f;
    if (i < n) {
        float s = 0.0f;
        for (int r = 0; r < v.world; ++r) {
            const float x = r == v.rank ? own
                                        : __hip_atomic_load(v.base[r] + (size_t)slot * v.n_pad + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            s = r == 0 ? x : s + x;
        }
        t = s * inv_w;
        grads[i] = t;
    }
    if (sq_part) {
        __shared__ double sc[kP2PThreads / kWave];
        const double q = block_sum<kP2PThreads / kWave>((double)t * (double)t, sc);
        if (threadIdx.x == 0) sq_part[blockIdx.x] = q;
    }
}
}  // namespace

struct aurppo_p2p {
    int rank, world, n_cap, n_pad, device;
    float* base;                 // local allocation: [2][n_pad] floats + kTailWords
    size_t bytes;
    void* peer[kP2PMaxWorld];    // opened mappings ([rank] = base); nullptr = not opened
    bool opened;
};

extern "C" int aurppo_p2p_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

extern "C" int aurppo_p2p_parts(int n) { return n > 0 ? (n + kP2PThreads - 1) / kP2PThreads : 0; }

extern "C" int aurppo_p2p_create(aurppo_p2p** out, int rank, int world, int max_floats, void* stream) {
    AURPPO_REQUIRE(out, AURPPO_EINVAL, "aurppo_p2p_create: null pointer");
    AURPPO_REQUIRE(world >= 1 && world <= kP2PMaxWorld && rank >= 0 && rank < world, AURPPO_ESHAPE,
                   "aurppo_p2p_create: rank %d of %d (at most %d ranks)", rank, world, kP2PMaxWorld);
    AURPPO_REQUIRE(max_floats > 0 && max_floats <= 128 * kP2PThreads, AURPPO_ESHAPE,
                   "aurppo_p2p_create: max_floats=%d (1..%d: one launch of at most 128 workgroups, one element per thread -- larger "
                   "buckets belong on RCCL's ring)", max_floats, 128 * kP2PThreads);
    aurppo_p2p* x = new aurppo_p2p();
    x->rank = rank; x->world = world; x->n_cap = max_floats;
    x->n_pad = ((max_floats + 63) / 64) * 64;
    x->bytes = sizeof(float) * (2 * (size_t)x->n_pad + kTailWords);
    x->opened = false;
    for (int r = 0; r < kP2PMaxWorld; ++r) x->peer[r] = nullptr;
    if (hipGetDevice(&x->device) != hipSuccess) x->device = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&x->base), x->bytes);
    if (e == hipSuccess) e = hipMemsetAsync(x->base, 0, x->bytes, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);      // peers may map and poll it as soon as they have the handle
    if (e != hipSuccess) {
        aurppo_set_error("aurppo_p2p_create: %s", hipGetErrorString(e));
        if (x->base) (void)hipFree(x->base);
        delete x;
        return AURPPO_EHIP;
    }
    x->peer[rank] = x->base;
    *out = x;
    return AURPPO_OK;
}

extern "C" int aurppo_p2p_get_handle(aurppo_p2p* x, void* handle_h) {
    AURPPO_REQUIRE(x && handle_h, AURPPO_EINVAL, "aurppo_p2p_get_handle: null pointer");
    hipIpcMemHandle_t h;
    AURPPO_HIP_TRY(hipIpcGetMemHandle(&h, x->base));
    memcpy(handle_h, &h, sizeof(h));
    return AURPPO_OK;
}

extern "C" int aurppo_p2p_open_peers(aurppo_p2p* x, const void* handles_h) {
    AURPPO_REQUIRE(x && handles_h, AURPPO_EINVAL, "aurppo_p2p_open_peers: null pointer");
    AURPPO_REQUIRE(!x->opened, AURPPO_EINVAL, "aurppo_p2p_open_peers: already opened");
    const char* hs = reinterpret_cast<const char*>(handles_h);
    for (int r = 0; r < x->world; ++r) {
        if (r == x->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, hs + (size_t)r * sizeof(h), sizeof(h));
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            aurppo_set_error("aurppo_p2p_open_peers: hipIpcOpenMemHandle for rank %d failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 must be "
                             "set in every rank's environment on this driver)", r, hipGetErrorString(e));
            return AURPPO_EHIP;
        }
        x->peer[r] = p;
    }
    x->opened = true;
    return AURPPO_OK;
}

extern "C" int aurppo_p2p_destroy(aurppo_p2p* x) {
    if (!x) return AURPPO_OK;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < x->world; ++r)
        if (r != x->rank && x->peer[r]) (void)hipIpcCloseMemHandle(x->peer[r]);
    if (x->base) (void)hipFree(x->base);
    delete x;
    return AURPPO_OK;
}

// 0 = every exchange so far met its peers; 1 + r = the flag of rank r did not arrive within the timeout at least once (sticky).
// Synchronises `stream`.
extern "C" int aurppo_p2p_status(aurppo_p2p* x, int* status_h, void* stream) {
    AURPPO_REQUIRE(x && status_h, AURPPO_EINVAL, "aurppo_p2p_status: null pointer");
    unsigned v = 0;
    AURPPO_HIP_TRY(hipMemcpyAsync(&v, reinterpret_cast<unsigned*>(x->base + 2 * (size_t)x->n_pad) + 3 * kFlagStride, sizeof(v),
                                  hipMemcpyDeviceToHost, (hipStream_t)stream));
    AURPPO_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    *status_h = (int)v;
    return AURPPO_OK;
}

extern "C" int aurppo_p2p_allreduce_mean_f32(aurppo_p2p* x, float* grads, int n, const float* step_dev, double* sq_part,
                                             double timeout_s, void* stream) {
    AURPPO_REQUIRE(x && grads && step_dev, AURPPO_EINVAL, "aurppo_p2p_allreduce_mean_f32: null pointer");
    AURPPO_REQUIRE(n > 0 && n <= x->n_cap, AURPPO_ESHAPE, "aurppo_p2p_allreduce_mean_f32: n=%d outside 1..%d", n, x->n_cap);
    AURPPO_REQUIRE(x->opened || x->world == 1, AURPPO_EINVAL, "aurppo_p2p_allreduce_mean_f32: peers not opened");
    P2PView v;
    v.rank = x->rank; v.world = x->world; v.n_pad = x->n_pad;
    for (int r = 0; r < kP2PMaxWorld; ++r) v.base[r] = reinterpret_cast<float*>(r < x->world ? x->peer[r] : nullptr);
    const unsigned long long ticks = (unsigned long long)((timeout_s > 0.0 ? timeout_s : 10.0) * 1e8);
    hipLaunchKernelGGL(k_p2p_allreduce, dim3((n + kP2PThreads - 1) / kP2PThreads), dim3(kP2PThreads), 0, (hipStream_t)stream, grads, n,
                       v, step_dev, 1.0f / (float)x->world, sq_part, ticks);
    AURPPO_LAUNCH_CHECK("k_p2p_allreduce");
    return AURPPO_OK;
}
