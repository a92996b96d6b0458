// K3: fused minibatch gather -- b_obs[mb], b_actions[mb], b_logprobs[mb], b_advantages[mb],
// b_returns[mb], b_values[mb] in one launch (src/ppo.py:219-220,225,236,251-257;
// src/robot_ppo.py:341-345).
//
// Design (gfx950).  This is the dominant HBM stream of the update: every sample row is read once
// at a random row offset and written once sequentially, (8D + 8A + 36) B per sample per epoch.
// A workgroup owns a tile of consecutive OUTPUT rows and walks all streams for it, so the index
// slice is fetched once and stays in L1/L2.  Within a stream, a row of C 16-byte chunks is served
// by LPR = min(64, pow2ceil(C)) adjacent lanes (a 256-B observation row = 16 lanes x float4, four
// rows per wave-instruction; scalar streams = one lane per row), four independent rows in flight
// per lane before the first store.  No LDS: nothing is reused, registers are the staging buffer.
#include "common.h"

namespace {

struct GatherArgs {
    const float* src[AURPPO_MAX_STREAMS];
    float* dst[AURPPO_MAX_STREAMS];
    int chunks[AURPPO_MAX_STREAMS];  // vector chunks per row
    int lpr_log2[AURPPO_MAX_STREAMS];
    int vec_log2[AURPPO_MAX_STREAMS];  // 0: float, 1: float2, 2: float4
    int n_streams;
    int M;
    int rows_per_wg;
};

template <typename V>
__device__ __forceinline__ void gather_stream(const int32_t* __restrict__ idx, const V* __restrict__ src,
                                              V* __restrict__ dst, int chunks, int lpr_log2, int row0, int row1) {
    constexpr int UNROLL = 4;
    const int lpr = 1 << lpr_log2;
    const int sub = threadIdx.x & (lpr - 1);
    const int rows_per_pass = blockDim.x >> lpr_log2;
    const int r_in_pass = threadIdx.x >> lpr_log2;
    for (int rbase = row0; rbase < row1; rbase += rows_per_pass * UNROLL) {
        int row[UNROLL];
        size_t so[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            row[u] = rbase + u * rows_per_pass + r_in_pass;
            so[u] = row[u] < row1 ? (size_t)idx[row[u]] * chunks : 0;
        }
        for (int part = sub; part < chunks; part += lpr) {
            V val[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (row[u] < row1) val[u] = src[so[u] + part];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (row[u] < row1) dst[(size_t)row[u] * chunks + part] = val[u];
        }
    }
}

__global__ __launch_bounds__(256) void k_gather(const int32_t* __restrict__ idx, GatherArgs a) {
    const int row0 = blockIdx.x * a.rows_per_wg;
    const int row1 = min(a.M, row0 + a.rows_per_wg);
    for (int s = 0; s < a.n_streams; ++s) {
        switch (a.vec_log2[s]) {
            case 2:
                gather_stream<float4>(idx, reinterpret_cast<const float4*>(a.src[s]),
                                      reinterpret_cast<float4*>(a.dst[s]), a.chunks[s], a.lpr_log2[s], row0, row1);
                break;
            case 1:
                gather_stream<float2>(idx, reinterpret_cast<const float2*>(a.src[s]),
                                      reinterpret_cast<float2*>(a.dst[s]), a.chunks[s], a.lpr_log2[s], row0, row1);
                break;
            default:
                gather_stream<float>(idx, a.src[s], a.dst[s], a.chunks[s], a.lpr_log2[s], row0, row1);
        }
    }
}

}  // namespace

extern "C" int aurppo_gather_f32(const int32_t* idx, int M, const float* const* src_h, float* const* dst_h,
                                 const int* row_elems_h, int n_streams, void* stream) {
    AURPPO_REQUIRE(idx && src_h && dst_h && row_elems_h, AURPPO_EINVAL, "aurppo_gather_f32: null pointer");
    AURPPO_REQUIRE(n_streams >= 1 && n_streams <= AURPPO_MAX_STREAMS, AURPPO_ESHAPE,
                   "aurppo_gather_f32: n_streams=%d outside [1,%d]", n_streams, AURPPO_MAX_STREAMS);
    AURPPO_REQUIRE(M >= 0, AURPPO_ESHAPE, "aurppo_gather_f32: M=%d negative", M);
    if (M == 0) return AURPPO_OK;
    GatherArgs a;
    a.n_streams = n_streams;
    a.M = M;
    size_t row_bytes = 0;
    for (int s = 0; s < n_streams; ++s) {
        AURPPO_REQUIRE(src_h[s] && dst_h[s], AURPPO_EINVAL, "aurppo_gather_f32: null stream %d", s);
        const int re = row_elems_h[s];
        AURPPO_REQUIRE(re >= 1, AURPPO_ESHAPE, "aurppo_gather_f32: row_elems[%d]=%d", s, re);
        int vl = 0;
        if (re % 4 == 0 && aligned_to(src_h[s], 16) && aligned_to(dst_h[s], 16)) vl = 2;
        else if (re % 2 == 0 && aligned_to(src_h[s], 8) && aligned_to(dst_h[s], 8)) vl = 1;
        a.src[s] = src_h[s];
        a.dst[s] = dst_h[s];
        a.vec_log2[s] = vl;
        a.chunks[s] = re >> vl;
        int l = 0;
        while ((1 << l) < a.chunks[s] && l < 6) ++l;
        a.lpr_log2[s] = l;
        row_bytes += (size_t)re * 4;
    }
    // Output-row tile per workgroup: four row slots per lane for the widest stream (its 4-deep unroll is
    // what keeps loads in flight), halved while that would leave fewer than 512 workgroups.
    int max_lpr_log2 = 0;
    for (int s = 0; s < n_streams; ++s) max_lpr_log2 = a.lpr_log2[s] > max_lpr_log2 ? a.lpr_log2[s] : max_lpr_log2;
    if (max_lpr_log2 < 4) max_lpr_log2 = 4;
    int rows = (256 >> max_lpr_log2) * 4;
    while (rows > 16 && (M + rows - 1) / rows < 512) rows >>= 1;
    (void)row_bytes;
    a.rows_per_wg = rows;
    const int grid = (M + rows - 1) / rows;
    hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, (hipStream_t)stream, idx, a);
    AURPPO_LAUNCH_CHECK("k_gather");
    return AURPPO_OK;
}
