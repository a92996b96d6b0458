// K3: fused minibatch gather -- b_obs[mb], b_actions[mb], b_logprobs[mb], b_advantages[mb],
// b_returns[mb], b_values[mb] in one launch (src/ppo.py:219-220,225,236,251-257;
// src/robot_ppo.py:341-345).
//
// Design (gfx950).  This is the dominant HBM stream of the update: every sample row is read once
// at a random row offset and written once sequentially, (8D + 8A + 36) B per sample per epoch.
//   * Streams are split on the host into ROW streams (more than one float per sample: observations,
//     actions) and SCALAR streams (one float per sample: log-prob, advantage, return, value).
//   * Row streams, row-centric: LPR = min(64, pow2ceil(widest row in chunks)) adjacent lanes own one
//     OUTPUT row and serve every row stream of it (a 256-B observation row = 16 lanes x float4, so a
//     wave-instruction moves four whole rows); UNROLL rows per lane are in flight at once.
//   * Scalar streams, thread-per-row: one lane fetches the sample's index once and all its scalars.
//   * Every load of the tile -- scalars first, then all row streams -- is issued before the first
//     store, so a workgroup pays ONE idx -> row latency chain, not one per stream.
//   * Stream loops are unrolled at compile time (NV row streams is a template parameter): stream
//     descriptors stay in SGPRs and the staging buffer is registers only -- no LDS, nothing is reused.
// Rows wider than LPR chunks (image observations) continue in a chunk-strided loop.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kMaxRowStreams = 3;

struct GatherArgs {
    // row streams (chunks >= 1 vector chunk per row, any vector width)
    const float* vsrc[kMaxRowStreams];
    float* vdst[kMaxRowStreams];
    int vchunks[kMaxRowStreams];
    int vvec[kMaxRowStreams];  // log2 of floats per chunk: 0 float, 1 float2, 2 float4
    // scalar streams (exactly one float per row)
    const float* ssrc[AURPPO_MAX_STREAMS];
    float* sdst[AURPPO_MAX_STREAMS];
    int n_scalar;
    int M;
    int lpr_log2;  // lanes per row for the row streams
    int rows_per_wg;
};

__device__ __forceinline__ float4 load_chunk(const float* __restrict__ base, size_t chunk, int vl) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vl == 2) {
        v = reinterpret_cast<const float4*>(base)[chunk];
    } else if (vl == 1) {
        const float2 t = reinterpret_cast<const float2*>(base)[chunk];
        v.x = t.x;
        v.y = t.y;
    } else {
        v.x = base[chunk];
    }
    return v;
}

__device__ __forceinline__ void store_chunk(float* __restrict__ base, size_t chunk, int vl, float4 v) {
    if (vl == 2) {
        reinterpret_cast<float4*>(base)[chunk] = v;
    } else if (vl == 1) {
        reinterpret_cast<float2*>(base)[chunk] = make_float2(v.x, v.y);
    } else {
        base[chunk] = v.x;
    }
}

template <int UNROLL, int NV>
__global__ __launch_bounds__(256) void k_gather(const int32_t* __restrict__ idx, const GatherArgs a) {
    const int row0 = blockIdx.x * a.rows_per_wg;
    const int row1 = min(a.M, row0 + a.rows_per_wg);

    // ---- scalar streams: issue loads (rows_per_wg <= 256: one lane per row)
    const int srow = row0 + (int)threadIdx.x;
    const bool s_on = a.n_scalar > 0 && srow < row1;
    float sval[AURPPO_MAX_STREAMS];
    if (s_on) {
        const int si = idx[srow];
#pragma unroll
        for (int s = 0; s < AURPPO_MAX_STREAMS; ++s)
            if (s < a.n_scalar) sval[s] = a.ssrc[s][si];
    }

    // ---- row streams
    if (NV > 0) {
        const int lpr = 1 << a.lpr_log2;
        const int sub = threadIdx.x & (lpr - 1);
        const int r_in_pass = threadIdx.x >> a.lpr_log2;
        const int rows_per_pass = 256 >> a.lpr_log2;
        for (int rbase = row0; rbase < row1; rbase += rows_per_pass * UNROLL) {
            int row[UNROLL];
            int src_row[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                row[u] = rbase + u * rows_per_pass + r_in_pass;
                src_row[u] = row[u] < row1 ? idx[row[u]] : -1;
            }
            float4 val[NV > 0 ? NV : 1][UNROLL];
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                if (sub < a.vchunks[s]) {
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u)
                        if (src_row[u] >= 0)
                            val[s][u] = load_chunk(a.vsrc[s], (size_t)src_row[u] * a.vchunks[s] + sub, a.vvec[s]);
                }
            }
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                if (sub < a.vchunks[s]) {
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u)
                        if (src_row[u] >= 0)
                            store_chunk(a.vdst[s], (size_t)row[u] * a.vchunks[s] + sub, a.vvec[s], val[s][u]);
                }
            }
            // rows wider than the lane group (images): remaining chunks, UNROLL rows in flight per lane
#pragma unroll
            for (int s = 0; s < NV; ++s) {
                if (a.vchunks[s] > lpr) {
                    for (int part = sub + lpr; part < a.vchunks[s]; part += lpr) {
                        float4 w[UNROLL];
#pragma unroll
                        for (int u = 0; u < UNROLL; ++u)
                            if (src_row[u] >= 0)
                                w[u] = load_chunk(a.vsrc[s], (size_t)src_row[u] * a.vchunks[s] + part, a.vvec[s]);
#pragma unroll
                        for (int u = 0; u < UNROLL; ++u)
                            if (src_row[u] >= 0)
                                store_chunk(a.vdst[s], (size_t)row[u] * a.vchunks[s] + part, a.vvec[s], w[u]);
                    }
                }
            }
        }
    }

    // ---- scalar streams: stores
    if (s_on) {
#pragma unroll
        for (int s = 0; s < AURPPO_MAX_STREAMS; ++s)
            if (s < a.n_scalar) a.sdst[s][srow] = sval[s];
    }
}

template <int UNROLL>
void launch_nv(int nv, int grid, hipStream_t st, const int32_t* idx, const GatherArgs& a) {
    switch (nv) {
        case 0: hipLaunchKernelGGL((k_gather<UNROLL, 0>), dim3(grid), dim3(256), 0, st, idx, a); break;
        case 1: hipLaunchKernelGGL((k_gather<UNROLL, 1>), dim3(grid), dim3(256), 0, st, idx, a); break;
        case 2: hipLaunchKernelGGL((k_gather<UNROLL, 2>), dim3(grid), dim3(256), 0, st, idx, a); break;
        default: hipLaunchKernelGGL((k_gather<UNROLL, 3>), dim3(grid), dim3(256), 0, st, idx, a);
    }
}

}  // namespace

extern "C" int aurppo_gather_f32(const int32_t* idx, int M, const float* const* src_h, float* const* dst_h,
                                 const int* row_elems_h, int n_streams, void* stream) {
    AURPPO_REQUIRE(idx && src_h && dst_h && row_elems_h, AURPPO_EINVAL, "aurppo_gather_f32: null pointer");
    AURPPO_REQUIRE(n_streams >= 1 && n_streams <= AURPPO_MAX_STREAMS, AURPPO_ESHAPE,
                   "aurppo_gather_f32: n_streams=%d outside [1,%d]", n_streams, AURPPO_MAX_STREAMS);
    AURPPO_REQUIRE(M >= 0, AURPPO_ESHAPE, "aurppo_gather_f32: M=%d negative", M);
    for (int s = 0; s < n_streams; ++s) {
        AURPPO_REQUIRE(src_h[s] && dst_h[s], AURPPO_EINVAL, "aurppo_gather_f32: null stream %d", s);
        AURPPO_REQUIRE(row_elems_h[s] >= 1, AURPPO_ESHAPE, "aurppo_gather_f32: row_elems[%d]=%d", s, row_elems_h[s]);
    }
    if (M == 0) return AURPPO_OK;
    // Tuning knobs (experiments only).  Measured on MI355X at M=131072, D=64 (profiles/r01): one row
    // slot per lane (UNROLL 1, 16-row tiles, 8192 workgroups at full occupancy) beats deeper unrolls
    // (rocprof: 24.2 us vs 26.4 / 28.3 us for UNROLL 2 / 4); a persistent variant that prefetched the
    // next tile's indices measured no better (24.8-27.9 us) and was dropped.  Rows wider than one lane
    // group (images) need the 4-deep unroll to keep loads in flight inside their chunk loop.
    const int kUnroll = aurppo_knobs().gather_unroll, kRowsOverride = aurppo_knobs().gather_rows;
    int widest = 1;
    for (int s = 0; s < n_streams; ++s) widest = row_elems_h[s] > widest ? row_elems_h[s] : widest;
    int unroll = widest > 256 ? 4 : 1;
    if (kUnroll == 1 || kUnroll == 2 || kUnroll == 4 || kUnroll == 8) unroll = kUnroll;
    hipStream_t st = (hipStream_t)stream;

    // Scalar streams ride along with the first launch; row streams go out kMaxRowStreams at a time.
    int first_row_stream = 0;
    bool scalars_done = false;
    while (!scalars_done || first_row_stream < n_streams) {
        GatherArgs a = {};
        a.M = M;
        int nv = 0, max_chunks = 1;
        int s = first_row_stream;
        for (; s < n_streams && nv < kMaxRowStreams; ++s) {
            const int re = row_elems_h[s];
            if (re == 1) continue;
            int vl = 0;
            if (re % 4 == 0 && aligned_to(src_h[s], 16) && aligned_to(dst_h[s], 16)) vl = 2;
            else if (re % 2 == 0 && aligned_to(src_h[s], 8) && aligned_to(dst_h[s], 8)) vl = 1;
            a.vsrc[nv] = src_h[s];
            a.vdst[nv] = dst_h[s];
            a.vvec[nv] = vl;
            a.vchunks[nv] = re >> vl;
            if (a.vchunks[nv] > max_chunks) max_chunks = a.vchunks[nv];
            ++nv;
        }
        // skip trailing scalar streams so the loop terminates
        while (s < n_streams && row_elems_h[s] == 1) ++s;
        first_row_stream = s;
        if (!scalars_done) {
            for (int k = 0; k < n_streams; ++k)
                if (row_elems_h[k] == 1) {
                    a.ssrc[a.n_scalar] = src_h[k];
                    a.sdst[a.n_scalar] = dst_h[k];
                    ++a.n_scalar;
                }
            scalars_done = true;
        }
        if (nv == 0 && a.n_scalar == 0) break;
        int l = 0;
        while ((1 << l) < max_chunks && l < 6) ++l;
        a.lpr_log2 = l;
        // one pass of `unroll` row slots per lane per workgroup (<= 256 rows so the scalar phase covers
        // the tile with one lane per row), halved while that leaves fewer than 512 workgroups
        int rows = nv ? (256 >> l) * unroll : 256;
        if (rows > 256) rows = 256;
        while (rows > (256 >> l) && rows > 16 && (M + rows - 1) / rows < 512) rows >>= 1;
        if (kRowsOverride > 0 && kRowsOverride <= 256) rows = kRowsOverride;
        a.rows_per_wg = rows;
        const int grid = (M + rows - 1) / rows;
        switch (unroll) {
            case 1: launch_nv<1>(nv, grid, st, idx, a); break;
            case 2: launch_nv<2>(nv, grid, st, idx, a); break;
            case 8: launch_nv<8>(nv, grid, st, idx, a); break;
            default: launch_nv<4>(nv, grid, st, idx, a);
        }
        AURPPO_LAUNCH_CHECK("k_gather");
    }
    return AURPPO_OK;
}
