// K11 -- 3x3 convolution (stride 1, zero padding 0..2) as an implicit GEMM on the bf16 matrix pipe: the hidden blocks of the
// robot policy's encoder (src/nets/base_cnns.py:32-45: nn.Conv2d(16,32,3,padding=1) ... nn.Conv2d(256,256,3)), forward AND the
// gradient with respect to the input (the same product with the filter transposed and flipped, padding 2 - p).
//
// Why: rocprofv3 counters over robot_ppo.update (profiles/r04/robot5_mfma_pmc.json) show the library's choice for these layers,
// miopenSp3AsmConv_v30_3_1_gfx9_fp32_f2x3 (Winograd F(2,3), 36-46 % of the update), issuing NO matrix instruction -- it is
// vector-ALU code, bounded by the 157 TFLOP/s fp32 FMA rate (x 2.25 for Winograd's fewer multiplies); fp32 products formed
// as six bf16 MFMAs (bf16x3.h, fp32-equivalent) have a ceiling of 2 500 / 6 = 417 TFLOP/s.
//
// GEMM view: M = output pixels of the whole batch, flattened (b, y, x); N = output channels; K = (tap, input channel), 16
// consecutive input channels of one tap per k-step (every hidden layer's width is a multiple of 16).  A wave owns 32
// consecutive pixels x up to four 32-channel blocks (4 x 16 accumulator registers):
//   * A operand straight from global memory, no LDS staging: lane (pixel m, half h) loads its 8 channels of the tap's input pixel
//     (NCHW: 8 dword loads, coalesced across the 32 pixels of the wave; zero outside the image), splits them into three bf16
//     planes in registers (44 vector instructions per k-step, against 6 x NB matrix instructions) -- a pixel is re-read once
//     per tap and channel group, from L1 / L2; the k-step after the current one is in flight while it computes;
//   * B operand: the filter as bf16 planes in operand order (k_conv_prep, L2-resident; the four waves of a workgroup work on
//     neighbouring pixel blocks of the same channel group and share its lines in L1);
//   * epilogue: the 32 x 32 accumulator blocks go through a padded LDS tile so that the stores are 128-byte rows of an output
//     plane instead of 16-byte pieces of 64 planes.
// Output is the convolution without bias (the block's bias + ReLU + max-pool tail is K9, csrc/pool.hip).
#include <stdio.h>

#pragma clang fp contract(fast)
#include "bf16x3.h"
#include "common.h"

using namespace bf3;

namespace {

typedef float f32x16c __attribute__((ext_vector_type(16)));
constexpr int kConvThreads = 256;
constexpr int kNBW = 4;              // 32-channel blocks per wave
constexpr int kTileLd = 33;          // floats per row of the epilogue's LDS tile

struct ConvArgs {
    const float* x;                  // (B, Cin, H, W)
    const unsigned short* wop;       // [n-block][k-step = tap * CG + cg][plane][lane][8]
    float* z;                        // (B, Cout, Ho, Wo)
    int B, Cin, H, W, Cout, Ho, Wo, pad;
    long long M;                     // B * Ho * Wo
    int CG;                          // Cin / 16
    int n_mb4;                       // workgroups along M (4 waves x kMB pixel blocks each)
};

__device__ __forceinline__ int acc_row_c(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }

// filter -> operand order.  transpose_flip = 0 (forward): B[k = (tap, ci)][n = co] = W[co][ci][tap];
// 1 (input gradient): the product's "input" channels are the forward pass's OUTPUT channels: B[k = (tap, co)][n = ci] =
// W[co][ci][8 - tap].  cin_gemm / cout_gemm are the product's own channel counts (cin_gemm a multiple of 16).
__global__ __launch_bounds__(256) void k_conv_prep(const float* __restrict__ w, int Co_w, int Ci_w, int cin_gemm, int cout_gemm,
                                                   int transpose_flip, unsigned short* __restrict__ wop, int taps) {
    const int CG = cin_gemm >> 4, KS = taps * CG, NBLK = (cout_gemm + 31) >> 5;
    const long long total = (long long)NBLK * KS * 64 * 8;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
        const long long r = e >> 9;
        const int ks = (int)(r % KS), nblk = (int)(r / KS);
        const int tap = ks / CG, cg = ks - tap * CG;
        const int n = nblk * 32 + (lane & 31), k = cg * 16 + 8 * (lane >> 5) + j;
        float v = 0.0f;
        if (n < cout_gemm) {
            if (!transpose_flip) v = w[((size_t)n * Ci_w + k) * taps + tap];               // W[co = n][ci = k][tap]
            else v = w[((size_t)k * Ci_w + n) * taps + (taps - 1 - tap)];                  // W[co = k][ci = n][flipped tap]
        }
        unsigned p0, p1, p2;
        split3(v, 0.0f, p0, p1, p2);
        const size_t at = (((size_t)(nblk * KS + ks) * 3) * 64 + lane) * 8 + j;
        wop[at] = (unsigned short)p0;
        wop[at + 512] = (unsigned short)p1;
        wop[at + 1024] = (unsigned short)p2;
    }
    (void)Co_w;
}

constexpr int kMB = 2;               // 32-pixel blocks per wave
constexpr int kKC = 2;               // k-steps per staged chunk of the filter

// LDS-DMA: 16 bytes per lane from global memory straight into LDS at dst + lane * 16 (wave-uniform dst), no registers in between
__device__ __forceinline__ void dma16(const void* src, void* dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_uniform, 16, 0, 0);
}

// Workgroup = 4 waves x (2 x 32 pixels) = 256 consecutive output pixels x NB 32-channel blocks.  The filter's fragments of two
// k-steps at a time (kKC x NB x 3 KB) are brought into LDS ONCE per workgroup by LDS-DMA, double-buffered, one barrier per chunk;
// every wave reads its B fragments from there (with each wave fetching its own from L2, the CU's L2 path -- about 16 B per
// cycle -- carried 12 KB per wave and k-step: 61-120 TFLOP/s, no better than the library).
// BUF (the input tensor is under 4 GB): the A values come through buffer loads -- per lane ONE 32-bit byte offset per tap and
// pixel block (out-of-image taps get an offset past the buffer: the hardware returns 0), the channel's offset in a scalar
// register.  The first version formed a 64-bit address and a zero select per value: 288 vector instructions per k-step against
// 48 matrix instructions, and the two barely overlap (SQ_VALU_MFMA_COEXEC_CYCLES 7 % of the matrix time, profiles/r04).
template <int NB, bool BUF>      // NB: 32-channel blocks of the channel group (1, 2 or 4)
__global__ __launch_bounds__(kConvThreads, 2) void k_conv3x3(const ConvArgs a) {
    constexpr int kChunkBytes = kKC * NB * 3 * 1024;
    constexpr int kLdsBytes = 2 * kChunkBytes > (kConvThreads / kWave) * 32 * kTileLd * 4 ? 2 * kChunkBytes
                                                                                        : (kConvThreads / kWave) * 32 * kTileLd * 4;
    __shared__ __attribute__((aligned(16))) char s_b[kLdsBytes];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb8 = (int)(blockIdx.x % (unsigned)a.n_mb4), ng = (int)(blockIdx.x / (unsigned)a.n_mb4);
    const int h = lane >> 5;
    const int HWo = a.Ho * a.Wo, HW = a.H * a.W;
    const int CG = a.CG, KS = 9 * CG;
    bool valid[kMB];
    int yy[kMB], xx[kMB];
    size_t img[kMB], out_px[kMB];
#pragma unroll
    for (int mb = 0; mb < kMB; ++mb) {
        const long long m = (((long long)mb8 * 4 + w) * kMB + mb) * 32 + (lane & 31);
        valid[mb] = m < a.M;
        int b = 0, y = 0, x = 0;
        if (valid[mb]) {
            b = (int)(m / HWo);
            const int r = (int)(m - (long long)b * HWo);
            y = r / a.Wo;
            x = r - y * a.Wo;
        }
        yy[mb] = y; xx[mb] = x;
        img[mb] = (size_t)b * a.Cin * HW;
        out_px[mb] = (size_t)b * a.Cout * HWo + (size_t)(y * a.Wo + x);
    }
    // the filter of this channel group: [nb][k-step][plane][lane][16 B]; a chunk = k-steps kc*kKC .. of every nb
    const char* const wgrp = reinterpret_cast<const char*>(a.wop) + (size_t)(ng * NB) * KS * 3 * 1024;
    // fragment f of a chunk (f = (kk * NB + nb) * 3 + plane): wave w brings fragments w, w + 4, ...
    auto stage = [&](int kc, int buf) {
        char* const dst = s_b + buf * kChunkBytes;
#pragma unroll
        for (int f = 0; f < kKC * NB * 3; ++f) {
            if ((f & 3) != w) continue;              // (wave-uniform)
            const int pl = f % 3, nb = (f / 3) % NB, kk = f / (3 * NB);
            const int ks = kc * kKC + kk;
            if (ks < KS) dma16(wgrp + ((size_t)(nb * KS + ks) * 3 + pl) * 1024 + lane * 16, dst + f * 1024);
        }
    };

    f32x16c acc[kMB][NB];
#pragma unroll
    for (int mb = 0; mb < kMB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mb][nb][e] = 0.0f;

    // the A values of one k-step: this lane's 8 channels of the tap's input pixel, for both pixel blocks
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, BUF ? (unsigned)((size_t)a.B * a.Cin * HW * 4) : 0u, 0x00020000);
    unsigned pix_off[kMB];           // BUF: byte offset of (b, channel 8 h, y - pad, x - pad) -- may wrap below zero, used with a valid tap only
#pragma unroll
    for (int mb = 0; mb < kMB; ++mb)
        pix_off[mb] = (unsigned)((img[mb] + (size_t)(8 * h) * HW) * 4) + (unsigned)(((yy[mb] - a.pad) * a.W + (xx[mb] - a.pad)) * 4);
    auto load_a = [&](int ks, float (&v)[kMB][8]) {
        const int tap = ks / CG, cg = ks - tap * CG;
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb) {
            const int iy = yy[mb] + ky - a.pad, ix = xx[mb] + kx - a.pad;
            const bool inb = valid[mb] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            if (BUF) {
                const unsigned vo = inb ? pix_off[mb] + (unsigned)((ky * a.W + kx) * 4) : 0xfffffff0u;     // past the buffer: reads 0
                const int so0 = cg * 16 * HW * 4;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[mb][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, vo, so0 + j * HW * 4, 0));
            } else {
                const size_t at = img[mb] + (size_t)(cg * 16 + 8 * h) * HW + (size_t)(inb ? iy * a.W + ix : 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = a.x[inb ? at + (size_t)j * HW : 0];
                    v[mb][j] = inb ? t : 0.0f;
                }
            }
        }
    };
    const int n_chunks = (KS + kKC - 1) / kKC;
    float abuf[2][kMB][8];           // k-step ks computes from abuf[ks & 1] while abuf[(ks + 1) & 1] is in flight (kKC = 2: the parity is kk's)
    static_assert(kKC == 2, "the A registers ping-pong on the k-step's position inside its chunk");
    stage(0, 0);
    load_a(0, abuf[0]);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // (the DMA pieces are older than the 16 loads of load_a)
    __syncthreads();
    for (int kc = 0; kc < n_chunks; ++kc) {
        if (kc + 1 < n_chunks) stage(kc + 1, (kc + 1) & 1);
        const char* const bsrc = s_b + (kc & 1) * kChunkBytes + lane * 16;
#pragma unroll
        for (int kk = 0; kk < kKC; ++kk) {
            const int ks = kc * kKC + kk;
            if (ks < KS) {
                float (&cur)[kMB][8] = abuf[kk];
                if (ks + 1 < KS) load_a(ks + 1, abuf[kk ^ 1]);
                Frag3 A[kMB];
#pragma unroll
                for (int mb = 0; mb < kMB; ++mb) {
                    unsigned p[4][3];
#pragma unroll
                    for (int q = 0; q < 4; ++q) split3(cur[mb][2 * q], cur[mb][2 * q + 1], p[q][0], p[q][1], p[q][2]);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const u32x4 v = {p[0][pl], p[1][pl], p[2][pl], p[3][pl]};
                        A[mb].p[pl] = __builtin_bit_cast(bf16x8, v);
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    Frag3 Bf;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) Bf.p[pl] = *reinterpret_cast<const bf16x8*>(bsrc + ((kk * NB + nb) * 3 + pl) * 1024);
#pragma unroll
                    for (int mb = 0; mb < kMB; ++mb) acc[mb][nb] = mma32x3(A[mb], Bf, acc[mb][nb]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the next chunk's DMA pieces (and the A values) have landed
        __syncthreads();
    }

    // ---- epilogue: per 32 x 32 block, accumulator -> LDS tile [n][pixel] -> 128-byte rows of the output planes
    float* const tile = reinterpret_cast<float*>(s_b) + w * 32 * kTileLd;
#pragma unroll
    for (int mb = 0; mb < kMB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n0 = (ng * NB + nb) * 32;
            if (n0 < a.Cout) {       // wave-uniform
#pragma unroll
                for (int e = 0; e < 16; ++e) tile[(lane & 31) * kTileLd + acc_row_c(e, lane)] = acc[mb][nb][e];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own stores (no other wave touches this tile)
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int n = h + 2 * i;
                    if (valid[mb] && n0 + n < a.Cout) a.z[out_px[mb] + (size_t)(n0 + n) * HWo] = tile[n * kTileLd + (lane & 31)];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
}

// ---- the same product for nn.Linear (src/nets/nets.py:21-27,33-39,45-51 with hidden_dim > 128, which the fused K7 / K7w steps do not
// cover): y (M, N) = x (M, K) . B (K, N), x row-major -- a 1 x 1 "convolution" whose pixels are the minibatch's rows.  Same
// filter staging and matrix loop as k_conv3x3; the A values of a k-step are 32 contiguous bytes per lane (two 16-byte loads), and
// the accumulator's lane-per-column layout already matches the row-major output (32 lanes = 128 contiguous bytes): no LDS tile.
struct LinArgs {
    const float* x;                  // (M, K)
    const unsigned short* wop;       // [n-block][k-step][plane][lane][8]
    float* y;                        // (M, N)
    const float* bias;               // (N,) or nullptr
    long long M;
    int K, N, n_mb;
    int act;                         // 0: none, 1: tanh (1 - 2 / (e^{2x} + 1), as the fused MLP steps form it)
};

template <int NB>
__global__ __launch_bounds__(kConvThreads, 2) void k_linear(const LinArgs a) {
    constexpr int kChunkBytes = kKC * NB * 3 * 1024;
    __shared__ __attribute__((aligned(16))) char s_b[2 * kChunkBytes];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mbq = (int)(blockIdx.x % (unsigned)a.n_mb), ng = (int)(blockIdx.x / (unsigned)a.n_mb);
    const int h = lane >> 5;
    const int KS = a.K >> 4;
    long long m0[kMB];
    const float* row[kMB];
    bool valid[kMB];
#pragma unroll
    for (int mb = 0; mb < kMB; ++mb) {
        m0[mb] = (((long long)mbq * 4 + w) * kMB + mb) * 32;
        const long long m = m0[mb] + (lane & 31);
        valid[mb] = m < a.M;
        row[mb] = a.x + (size_t)(valid[mb] ? m : 0) * a.K + 8 * h;
    }
    const char* const wgrp = reinterpret_cast<const char*>(a.wop) + (size_t)(ng * NB) * KS * 3 * 1024;
    auto stage = [&](int kc, int buf) {
        char* const dst = s_b + buf * kChunkBytes;
#pragma unroll
        for (int f = 0; f < kKC * NB * 3; ++f) {
            if ((f & 3) != w) continue;
            const int pl = f % 3, nb = (f / 3) % NB, kk = f / (3 * NB);
            const int ks = kc * kKC + kk;
            if (ks < KS) dma16(wgrp + ((size_t)(nb * KS + ks) * 3 + pl) * 1024 + lane * 16, dst + f * 1024);
        }
    };
    f32x16c acc[kMB][NB];
#pragma unroll
    for (int mb = 0; mb < kMB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mb][nb][e] = 0.0f;
    auto load_a = [&](int ks, float4 (&v)[kMB][2]) {
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb) {
            const float4* p = reinterpret_cast<const float4*>(row[mb] + ks * 16);
            v[mb][0] = p[0];
            v[mb][1] = p[1];
        }
    };
    const int n_chunks = (KS + kKC - 1) / kKC;
    float4 abuf[2][kMB][2];
    stage(0, 0);
    load_a(0, abuf[0]);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // (the DMA pieces are older than the 4 loads of load_a)
    __syncthreads();
    for (int kc = 0; kc < n_chunks; ++kc) {
        if (kc + 1 < n_chunks) stage(kc + 1, (kc + 1) & 1);
        const char* const bsrc = s_b + (kc & 1) * kChunkBytes + lane * 16;
#pragma unroll
        for (int kk = 0; kk < kKC; ++kk) {
            const int ks = kc * kKC + kk;
            if (ks < KS) {
                float4 (&cur)[kMB][2] = abuf[kk];
                if (ks + 1 < KS) load_a(ks + 1, abuf[kk ^ 1]);
                Frag3 A[kMB];
#pragma unroll
                for (int mb = 0; mb < kMB; ++mb) {
                    const float c[8] = {cur[mb][0].x, cur[mb][0].y, cur[mb][0].z, cur[mb][0].w, cur[mb][1].x, cur[mb][1].y, cur[mb][1].z, cur[mb][1].w};
                    unsigned p[4][3];
#pragma unroll
                    for (int q = 0; q < 4; ++q) split3(valid[mb] ? c[2 * q] : 0.0f, valid[mb] ? c[2 * q + 1] : 0.0f, p[q][0], p[q][1], p[q][2]);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const u32x4 v = {p[0][pl], p[1][pl], p[2][pl], p[3][pl]};
                        A[mb].p[pl] = __builtin_bit_cast(bf16x8, v);
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    Frag3 Bf;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) Bf.p[pl] = *reinterpret_cast<const bf16x8*>(bsrc + ((kk * NB + nb) * 3 + pl) * 1024);
#pragma unroll
                    for (int mb = 0; mb < kMB; ++mb) acc[mb][nb] = mma32x3(A[mb], Bf, acc[mb][nb]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int mb = 0; mb < kMB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = (ng * NB + nb) * 32 + (lane & 31);
            if (col < a.N) {
                const float bv = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long long r = m0[mb] + acc_row_c(e, lane);
                    float v = acc[mb][nb][e] + bv;
                    if (a.act == 1) v = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * v) + 1.0f);
                    if (r < a.M) a.y[(size_t)r * a.N + col] = v;
                }
            }
        }
}

// ---- nn.Linear's weight gradient on the same arithmetic: dW (N, K) = dY^T (N, M) . X (M, K), both operands row-major with the
// minibatch's rows as the product's inner dimension.  Both operands change every k-step, so both are split; to split every
// element once per WORKGROUP (not once per wave that uses it) the workgroup stages 32 rows of its 128 dY columns and its 128 X
// columns as bf16-plane X images in LDS (bf16x3.h: 8-byte stores along a row, fragments by transposed reads across rows -- the
// images k_mlp_step3 lands its observation rows in), and its four waves each form a 64 x 64 quarter of the 128 x 128 tile.
// The inner dimension is cut into S slices: partial[s] (N, K) per slice, summed by the caller in fixed order (deterministic).
struct WgradArgs {
    const float* dy;                 // (M, N)
    const float* x;                  // (M, K)
    float* part;                     // (S, N, K)
    long long M;
    int N, K, rows_per_slice, n_tiles_n, n_tiles_k;
};

__global__ __launch_bounds__(kConvThreads, 2) void k_linear_wgrad(const WgradArgs a) {
    __shared__ __attribute__((aligned(16))) char s_img[2 * 2 * 3 * kXPlane];      // [dY | X][64-column half][3 planes][32 rows x 128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = (int)(blockIdx.x % (unsigned)(a.n_tiles_n * a.n_tiles_k)), s = (int)(blockIdx.x / (unsigned)(a.n_tiles_n * a.n_tiles_k));
    const int n0 = (tile / a.n_tiles_k) * 128, k0 = (tile % a.n_tiles_k) * 128;
    const long long m_lo = (long long)s * a.rows_per_slice;
    long long m_hi = m_lo + a.rows_per_slice;
    if (m_hi > a.M) m_hi = a.M;
    char* const imgA = s_img;                               // dY columns n0 .. n0 + 127
    char* const imgB = s_img + 2 * 3 * kXPlane;            // X columns k0 .. k0 + 127
    const int wn = w >> 1, wk = w & 1;                      // this wave's 64 x 64 quarter
    // staging: a chunk = 32 rows x 128 columns of each operand = 1024 float4 per operand, four per thread
    const int r_of = tid >> 5, c4 = (tid & 31) * 4;         // + 8 rows per further slot
    f32x16c acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    float pa[4][4], pb[4][4];
    unsigned pok = 0u;               // bit u: dY piece u is real; bit 4 + u: X piece u (the zeroing waits until the chunk is staged)
    auto fetch = [&](long long m0) {
        pok = 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long r = m0 + r_of + 8 * u;
            const bool okr = r < m_hi;
            const bool oka = okr && n0 + c4 < a.N, okb = okr && k0 + c4 < a.K;      // (N and K are multiples of 4)
            const float4 va = *reinterpret_cast<const float4*>(a.dy + (oka ? (size_t)r * a.N + n0 + c4 : (size_t)0));
            const float4 vb = *reinterpret_cast<const float4*>(a.x + (okb ? (size_t)r * a.K + k0 + c4 : (size_t)0));
            pa[u][0] = va.x; pa[u][1] = va.y; pa[u][2] = va.z; pa[u][3] = va.w;
            pb[u][0] = vb.x; pb[u][1] = vb.y; pb[u][2] = vb.z; pb[u][3] = vb.w;
            pok |= (oka ? 1u : 0u) << u;
            pok |= (okb ? 1u : 0u) << (4 + u);
        }
    };
    fetch(m_lo);
    for (long long m0 = m_lo; m0 < m_hi; m0 += 32) {
        __syncthreads();                                    // the previous chunk's fragments have been read
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r_of + 8 * u;
            const bool oka = (pok >> u) & 1u, okb = (pok >> (4 + u)) & 1u;
            store_x4(imgA + (c4 >> 6) * 3 * kXPlane, r, c4 & 63, oka ? pa[u][0] : 0.0f, oka ? pa[u][1] : 0.0f, oka ? pa[u][2] : 0.0f,
                     oka ? pa[u][3] : 0.0f);
            store_x4(imgB + (c4 >> 6) * 3 * kXPlane, r, c4 & 63, okb ? pb[u][0] : 0.0f, okb ? pb[u][1] : 0.0f, okb ? pb[u][2] : 0.0f,
                     okb ? pb[u][3] : 0.0f);
        }
        __syncthreads();
        if (m0 + 32 < m_hi) fetch(m0 + 32);                 // the next chunk's rows, behind this chunk's products
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const Frag3 A0 = x_cols(imgA + wn * 3 * kXPlane, ks, 0, lane), A1 = x_cols(imgA + wn * 3 * kXPlane, ks, 32, lane);
            const Frag3 B0 = x_cols(imgB + wk * 3 * kXPlane, ks, 0, lane), B1 = x_cols(imgB + wk * 3 * kXPlane, ks, 32, lane);
            acc[0][0] = mma32x3(A0, B0, acc[0][0]);
            acc[0][1] = mma32x3(A0, B1, acc[0][1]);
            acc[1][0] = mma32x3(A1, B0, acc[1][0]);
            acc[1][1] = mma32x3(A1, B1, acc[1][1]);
        }
    }
    // C[m = dY column][n = X column]: the lane holds the X column, its registers the dY columns -- rows of `part` are contiguous over the lanes
    float* const out = a.part + (size_t)s * a.N * a.K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = k0 + wk * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = n0 + wn * 64 + i * 32 + acc_row_c(e, lane);
                if (row < a.N && col < a.K) out[(size_t)row * a.K + col] = acc[i][j][e];
            }
        }
}


// ---- K12: the 3x3 convolution's WEIGHT gradient on the same arithmetic.
//     dW[co][ci][ky][kx] = sum over (b, y, x) of dY[b][co][y][x] * X[b][ci][y + ky - p][x + kx - p]
// as a product C[m = co][n = ci * 9 + tap] = A[m][k = pixel] . B[k = pixel][n] whose inner dimension is the batch's output pixels
// (flattened) and whose output IS the filter's memory layout.  Both operands are new every k-step, so both are split: once per
// WORKGROUP, through LDS.  In NCHW the inner dimension is the contiguous one of both operands (a channel's pixels), which is the
// F image of bf16x3.h ([row = channel][32 samples], 8-byte stores of 4 consecutive samples, fragments by ds_read_b128 along a
// row): one image holds the workgroup's dY channels (64 WM rows) and its (ci, tap) columns (64 (4 / WM) rows) for a chunk of
// 32 pixels; each of the four waves forms a 64 x 64 part (2 x 2 accumulator blocks) of the tile.
//   * a thread stages 4 consecutive pixels of 8 (10) rows; the pixels' offsets / tap masks of a chunk come from a 32-entry table
//     that 32 lanes fill ahead of its use (two integer divisions per pixel, once per workgroup instead of once per row);
//   * loads are buffer loads relative to the slice's first image (32-bit offsets whatever the tensor's size); a tap outside the
//     image, a row past the tensor and a pixel past the slice get an offset past the buffer: the hardware returns 0.  QUAD (the
//     map's width a multiple of 4, padding <= 1: a quad of output pixels never leaves its row): one 16-byte load per row and
//     quad -- the X quad of a tap one column to the left starts at its second pixel and is shifted in registers;
//   * the values of chunk c + 2 are requested while chunk c is multiplied (two register sets);
//   * the pixels are cut into S slices (grid = tiles x S); k_fold_slices sums the partial filters in slice order (deterministic).
struct ConvWgradArgs {
    const float* dy;                 // (B, Co, Ho, Wo)
    const float* x;                  // (B, Ci, H, W)
    float* part;                     // (S, Co, Ci * 9)
    int B, Ci, H, W, Co, Ho, Wo, pad;
    long long M;                     // B * Ho * Wo
    int NK;                          // Ci * 9
    int px_per_slice;                // a multiple of 32
    int n_tiles_m, n_tiles_n;
};

constexpr unsigned kOob = 0x80000000u;      // buffer offsets from here on read 0 (num_records <= 2^31)

template <int PL>
__device__ __forceinline__ Frag3 wg_rows(const char* img, int f0, int ks, int lane) {       // A[m = f0 + ..][k] / B[k][n = f0 + ..]
    const int o = (foff(lane & 31, 8 * (lane >> 5)) ^ (ks << 5)) + f0 * kFRow;      // (f0 a multiple of 16: lane part + constants, bf16x3.h)
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = lds_b128(img + p * PL + o);
    return f;
}

template <int WM, bool QUAD>      // WM waves along the output channels: tile = 64 WM channels x 64 (4 / WM) filter columns
__global__ __launch_bounds__(kConvThreads, 2) void k_conv3x3_wgrad(const ConvWgradArgs a) {
    constexpr int kRowsA = 64 * WM, kRowsB = 64 * (4 / WM), kSlotsA = 2 * WM, kSlots = (kRowsA + kRowsB) / 32;
    constexpr int PL = (kRowsA + kRowsB) * kFRow;         // one bf16 plane of the image
    __shared__ __attribute__((aligned(16))) char s_img[3 * PL];
    // per chunk (three in flight): row 0 = the dY offsets of its 32 pixels, rows 1..9 = the X offsets per tap (the tap's shift
    // included); a pixel past the slice / a tap outside the image holds kOob
    __shared__ __attribute__((aligned(16))) unsigned s_tab[3][10][32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = a.n_tiles_m * a.n_tiles_n;
    const int tile = (int)(blockIdx.x % (unsigned)tiles), s = (int)(blockIdx.x / (unsigned)tiles);
    const int co0 = (tile / a.n_tiles_n) * kRowsA, n0 = (tile % a.n_tiles_n) * kRowsB;
    const int HoWo = a.Ho * a.Wo, HW = a.H * a.W;
    const long long m_lo = (long long)s * a.px_per_slice;
    long long m_hi = m_lo + a.px_per_slice;
    if (m_hi > a.M) m_hi = a.M;
    const int n_px = (int)(m_hi - m_lo);
    const int b_lo = (int)(m_lo / HoWo), r_lo = (int)(m_lo - (long long)b_lo * HoWo);
    // buffers that start at the slice's first image
    const size_t left_a = (size_t)(a.B - b_lo) * a.Co * HoWo * 4, left_b = (size_t)(a.B - b_lo) * a.Ci * HW * 4;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.dy + (size_t)b_lo * a.Co * HoWo), 0, (unsigned)(left_a < kOob ? left_a : kOob), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (size_t)b_lo * a.Ci * HW), 0, (unsigned)(left_b < kOob ? left_b : kOob), 0x00020000);
    // this thread's image rows (crow + 32 u) and its pixel quad
    const int q4 = (tid & 7) * 4, crow = tid >> 3;
    // rows past the tensor (channels >= Co, filter columns >= Ci * 9) are staged from wherever offset 0 + ... lands or from beyond
    // the buffer: their products are never stored
    unsigned row_off[kSlots];
    int row_tap[kSlots];
#pragma unroll
    for (int u = 0; u < kSlots; ++u) {
        const int row = crow + 32 * u;
        if (u < kSlotsA) {
            const int co = co0 + row;
            row_off[u] = co < a.Co ? (unsigned)co * (unsigned)HoWo * 4u : kOob;
            row_tap[u] = 0;
        } else {
            const int n = n0 + row - kRowsA;
            const int ci = n / 9;
            row_off[u] = n < a.NK ? (unsigned)(ci * HW) * 4u : kOob;
            row_tap[u] = 1 + (n - 9 * ci);
        }
    }
    // The table of chunk c: lane i < 32 of EVERY wave decomposes pixel i (two divisions by float reciprocal + fix-up: the slice's
    // pixel indices stay below 2^23, conv_wgrad_plan) and wave w fills rows w, w + 4, w + 8 -- left to 32 lanes of one wave, the
    // nine taps and two integer divisions were ~190 vector instructions per chunk on the wave every barrier waits for.
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)a.Wo;
    auto fdiv = [](int n, int d, float inv, int& rem) {
        int q = (int)((float)n * inv);
        int r = n - q * d;
        if (r < 0) { q -= 1; r += d; }
        else if (r >= d) { q += 1; r -= d; }
        rem = r;
        return q;
    };
    auto table = [&](int c) {
        if (lane < 32) {
            const int i = c * 32 + lane;
            unsigned (*const tab)[32] = s_tab[c % 3];
            int r, x;
            const int b = fdiv(r_lo + i, HoWo, inv_howo, r);        // (relative to the first pixel of image b_lo)
            const int y = fdiv(r, a.Wo, inv_wo, x);
            const int base = b * a.Ci * HW + (y - a.pad) * a.W + (x - a.pad);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int t = w + 4 * k;                             // (wave-uniform)
                if (t < 10) {
                    unsigned e;
                    if (t == 0) {
                        e = (unsigned)(b * a.Co * HoWo + r) * 4u;
                    } else {
                        const int ky = (t - 1) / 3, kx = (t - 1) - 3 * ky;
                        const int iy = y + ky - a.pad, ix = x + kx - a.pad;
                        const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                        e = in ? (unsigned)(base + ky * a.W + kx) * 4u : kOob;
                    }
                    tab[t][lane] = i < n_px ? e : kOob;
                }
            }
        }
    };
    // (the loads' results are not touched here: the shift of a row-start quad and the zero of a row-end pixel wait in `fl` -- two bits
    // per row -- until the chunk is split, two chunks later.  Applied at this point, the selects sat right behind the loads and every
    // request was a wait for its own loads.)
    auto fetch = [&](int c, float (&v)[kSlots][4], unsigned& fl) {
        const unsigned (*const tab)[32] = s_tab[c % 3];
        fl = 0u;
#pragma unroll
        for (int u = 0; u < kSlots; ++u) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(&tab[row_tap[u]][q4]);        // this row's offsets of the quad's pixels
            if (QUAD) {
                // the quad sits in one row of its image: pixel 1 is inside the image for every tap whose row is (W >= 4, pad <= 1);
                // pixel 0 / pixel 3 may fall off the row's ends
                u32x4 L;
                if (u < kSlotsA) {
                    L = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, t.x + row_off[u], 0, 0));
                    v[u][0] = __uint_as_float(L.x); v[u][1] = __uint_as_float(L.y);
                    v[u][2] = __uint_as_float(L.z); v[u][3] = __uint_as_float(L.w);
                } else {
                    const bool shl = t.x >= kOob;                 // pixel 0 is off the row: the load starts at pixel 1
                    const unsigned off = t.y < kOob ? (shl ? t.y : t.y - 4u) : kOob;
                    L = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, off + row_off[u], 0, 0));
                    v[u][0] = __uint_as_float(L.x); v[u][1] = __uint_as_float(L.y);
                    v[u][2] = __uint_as_float(L.z); v[u][3] = __uint_as_float(L.w);
                    fl |= (shl ? 1u : 0u) << (2 * u);
                    fl |= (t.w < kOob ? 0u : 2u) << (2 * u);
                }
            } else {
                const __amdgpu_buffer_rsrc_t r = u < kSlotsA ? ra : rb;
                v[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, t.x + row_off[u], 0, 0));
                v[u][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, t.y + row_off[u], 0, 0));
                v[u][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, t.z + row_off[u], 0, 0));
                v[u][3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, t.w + row_off[u], 0, 0));
            }
        }
    };
    const int wn = WM == 2 ? (w >> 1) : 0, wk = WM == 2 ? (w & 1) : w;
    const int rowA = wn * 64, rowB = kRowsA + wk * 64;
    f32x16c acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    const int n_chunks = (n_px + 31) >> 5;
    // one chunk: its values (requested two chunks ago) -> planes -> products; meanwhile chunk c + 2 is requested into the same registers
    auto step = [&](int c, float (&v)[kSlots][4], unsigned& fl) {
        __syncthreads();                                    // the previous chunk's fragments have been read; table c + 2 is written
#pragma unroll
        for (int u = 0; u < kSlots; ++u) {
            float x0 = v[u][0], x1 = v[u][1], x2 = v[u][2], x3 = v[u][3];
            if (QUAD && u >= kSlotsA) {
                const bool shl = (fl >> (2 * u)) & 1u, z3 = (fl >> (2 * u + 1)) & 1u;
                x3 = shl ? x2 : (z3 ? 0.0f : x3);
                x2 = shl ? x1 : x2;
                x1 = shl ? x0 : x1;
                x0 = shl ? 0.0f : x0;
            }
            unsigned a0, a1, a2, b0, b1, b2;
            split3(x0, x1, a0, a1, a2);
            split3(x2, x3, b0, b1, b2);
            const int o = foff(crow, q4) + 32 * u * kFRow;      // = foff(crow + 32 u, q4)
            *reinterpret_cast<u32x2*>(s_img + 0 * PL + o) = u32x2{a0, b0};
            *reinterpret_cast<u32x2*>(s_img + 1 * PL + o) = u32x2{a1, b1};
            *reinterpret_cast<u32x2*>(s_img + 2 * PL + o) = u32x2{a2, b2};
        }
        __syncthreads();
        if (c + 2 < n_chunks) fetch(c + 2, v, fl);
        if (c + 3 < n_chunks) table(c + 3);                 // (slot c % 3: last read by fetch(c), before this chunk's first barrier)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const Frag3 A0 = wg_rows<PL>(s_img, rowA, ks, lane), A1 = wg_rows<PL>(s_img, rowA + 32, ks, lane);
            const Frag3 B0 = wg_rows<PL>(s_img, rowB, ks, lane), B1 = wg_rows<PL>(s_img, rowB + 32, ks, lane);
            acc[0][0] = mma32x3(A0, B0, acc[0][0]);
            acc[0][1] = mma32x3(A0, B1, acc[0][1]);
            acc[1][0] = mma32x3(A1, B0, acc[1][0]);
            acc[1][1] = mma32x3(A1, B1, acc[1][1]);
        }
    };
    float va[kSlots][4], vb[kSlots][4];
    unsigned fa = 0u, fb = 0u;
    table(0);
    if (n_chunks > 1) table(1);
    if (n_chunks > 2) table(2);
    __syncthreads();
    fetch(0, va, fa);
    if (n_chunks > 1) fetch(1, vb, fb);
    for (int c = 0; c < n_chunks; c += 2) {
        step(c, va, fa);
        if (c + 1 < n_chunks) step(c + 1, vb, fb);
    }
    // C[m = co][n]: the lane holds the filter column, its registers the channels -- rows of `part` are contiguous over the lanes
    float* const out = a.part + (size_t)s * a.Co * a.NK;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wk * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = co0 + rowA + i * 32 + acc_row_c(e, lane);
                if (row < a.Co && col < a.NK) out[(size_t)row * a.NK + col] = acc[i][j][e];
            }
        }
}

// out[i] = part[0][i] + part[1][i] + ... in slice order; with `cols` > 0 also colsum[j] = the same sum over a (S, cols) array
__global__ __launch_bounds__(256) void k_fold_slices(const float* __restrict__ part, int S, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float t = 0.0f;
    int sl = 0;
    for (; sl + 8 <= S; sl += 8) {
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = part[(size_t)(sl + k) * n + i];
#pragma unroll
        for (int k = 0; k < 8; ++k) t += x[k];
    }
    for (; sl < S; ++sl) t += part[(size_t)sl * n + i];
    out[i] = t;
}

}  // namespace

extern "C" size_t aurppo_conv3x3_wop_bytes(int cin_gemm, int cout_gemm) {
    const size_t nblk = (size_t)((cout_gemm + 31) / 32 + kNBW);      // (+ one group of slack: a wave reads whole groups)
    return nblk * 9 * (size_t)(cin_gemm / 16) * 3 * 1024 + 64;
}

// mode 0: z = conv2d(x, w, padding = pad)                      x (B, Ci, H, W), w (Co, Ci, 3, 3), z (B, Co, H + 2 pad - 2, ...)
// mode 1: z = d conv2d / d input applied to x:  x (B, Co, Ho, Wo) is the output gradient, z (B, Ci, Ho + 2 - 2 pad, ...) the
//         input gradient, w the SAME (Co, Ci, 3, 3) filter, pad the FORWARD padding.
extern "C" int aurppo_conv3x3_f32(const float* x, const float* w, float* z, int B, int Ci_w, int Co_w, int H, int W, int pad,
                                  int mode, void* wop_ws, void* stream) {
    AURPPO_REQUIRE(x && w && z && wop_ws, AURPPO_EINVAL, "aurppo_conv3x3_f32: null pointer");
    AURPPO_REQUIRE(mode == 0 || mode == 1, AURPPO_EINVAL, "aurppo_conv3x3_f32: mode %d", mode);
    AURPPO_REQUIRE(pad >= 0 && pad <= 2, AURPPO_ESHAPE, "aurppo_conv3x3_f32: pad=%d (0..2)", pad);
    const int cin = mode == 0 ? Ci_w : Co_w, cout = mode == 0 ? Co_w : Ci_w;
    const int p = mode == 0 ? pad : 2 - pad;
    AURPPO_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0 && cout > 0 && cin % 16 == 0, AURPPO_ESHAPE,
                   "aurppo_conv3x3_f32: B=%d H=%d W=%d, %d input channels (a multiple of 16), %d output channels", B, H, W, cin, cout);
    const int Ho = H + 2 * p - 2, Wo = W + 2 * p - 2;
    AURPPO_REQUIRE(Ho > 0 && Wo > 0, AURPPO_ESHAPE, "aurppo_conv3x3_f32: empty output (%d x %d)", Ho, Wo);
    AURPPO_REQUIRE((size_t)B * cin * H * W < ((size_t)1 << 40) && aligned_to(wop_ws, 16), AURPPO_ESHAPE,
                   "aurppo_conv3x3_f32: operand too large / workspace not 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    unsigned short* wop = reinterpret_cast<unsigned short*>(wop_ws);
    hipLaunchKernelGGL(k_conv_prep, dim3(128), dim3(256), 0, s, w, Co_w, Ci_w, cin, cout, mode, wop, 9);
    AURPPO_LAUNCH_CHECK("k_conv_prep");
    ConvArgs a;
    a.x = x; a.wop = wop; a.z = z;
    a.B = B; a.Cin = cin; a.H = H; a.W = W; a.Cout = cout; a.Ho = Ho; a.Wo = Wo; a.pad = p;
    a.M = (long long)B * Ho * Wo;
    a.CG = cin / 16;
    const long long n_mb4 = (a.M + 32 * 4 * kMB - 1) / (32 * 4 * kMB);     // workgroups along M: 4 waves x kMB pixel blocks
    AURPPO_REQUIRE(n_mb4 < (1ll << 30), AURPPO_ESHAPE, "aurppo_conv3x3_f32: too many pixel blocks");
    a.n_mb4 = (int)n_mb4;
    const int nblk = (cout + 31) / 32;
    const int NB = nblk >= 4 ? 4 : (nblk >= 2 ? 2 : 1);
    const int n_ng = (nblk + NB - 1) / NB;
    const long long grid = n_mb4 * n_ng;
    AURPPO_REQUIRE(grid < (1ll << 31), AURPPO_ESHAPE, "aurppo_conv3x3_f32: grid too large");
    // buffer loads address 32 bits: inputs under 4 GB (minus the slack a negative tap offset may wrap through); larger ones keep 64-bit addresses
    const bool buf = (size_t)B * cin * H * W * 4 < ((size_t)1 << 32) - ((size_t)1 << 20);
    const dim3 g((unsigned)grid), blk(kConvThreads);
    if (buf) {
        if (NB == 4) hipLaunchKernelGGL((k_conv3x3<4, true>), g, blk, 0, s, a);
        else if (NB == 2) hipLaunchKernelGGL((k_conv3x3<2, true>), g, blk, 0, s, a);
        else hipLaunchKernelGGL((k_conv3x3<1, true>), g, blk, 0, s, a);
    } else {
        if (NB == 4) hipLaunchKernelGGL((k_conv3x3<4, false>), g, blk, 0, s, a);
        else if (NB == 2) hipLaunchKernelGGL((k_conv3x3<2, false>), g, blk, 0, s, a);
        else hipLaunchKernelGGL((k_conv3x3<1, false>), g, blk, 0, s, a);
    }
    AURPPO_LAUNCH_CHECK("k_conv3x3");
    return AURPPO_OK;
}

// mode 0: y (M, N_w) = x (M, K_w) . w (N_w, K_w)^T      -- nn.Linear without its bias; K_w a multiple of 16
// mode 1: y (M, K_w) = x (M, N_w) . w (N_w, K_w)         -- the gradient with respect to the input (x is dY); N_w a multiple of 16
// wop_ws: aurppo_conv3x3_wop_bytes(product's K, product's N) / 9 bytes suffice; the same function's size is accepted.
static int linear_impl(const float* x, const float* w, const float* bias, int act, float* y, long long M, int K_w, int N_w, int mode,
                       void* wop_ws, void* stream);

extern "C" int aurppo_linear_f32(const float* x, const float* w, float* y, long long M, int K_w, int N_w, int mode, void* wop_ws,
                                 void* stream) {
    return linear_impl(x, w, nullptr, 0, y, M, K_w, N_w, mode, wop_ws, stream);
}

// y (M, N_w) = act(x (M, K_w) . w (N_w, K_w)^T + bias): nn.Linear with its bias and, act = 1, the nn.Tanh behind it
// (src/nets/nets.py:21-27: every hidden layer of the reference's MLPs) in the product's epilogue.
extern "C" int aurppo_linear_bias_act_f32(const float* x, const float* w, const float* bias, float* y, long long M, int K_w, int N_w,
                                          int act, void* wop_ws, void* stream) {
    AURPPO_REQUIRE(act == 0 || act == 1, AURPPO_EINVAL, "aurppo_linear_bias_act_f32: act %d", act);
    return linear_impl(x, w, bias, act, y, M, K_w, N_w, 0, wop_ws, stream);
}

static int linear_impl(const float* x, const float* w, const float* bias, int act, float* y, long long M, int K_w, int N_w, int mode,
                       void* wop_ws, void* stream) {
    AURPPO_REQUIRE(x && w && y && wop_ws, AURPPO_EINVAL, "aurppo_linear_f32: null pointer");
    AURPPO_REQUIRE(mode == 0 || mode == 1, AURPPO_EINVAL, "aurppo_linear_f32: mode %d", mode);
    const int K = mode == 0 ? K_w : N_w, N = mode == 0 ? N_w : K_w;
    AURPPO_REQUIRE(M > 0 && K > 0 && N > 0 && K % 16 == 0, AURPPO_ESHAPE,
                   "aurppo_linear_f32: M=%lld, inner dimension %d (a multiple of 16), %d columns", M, K, N);
    AURPPO_REQUIRE(aligned_to(x, 16) && aligned_to(wop_ws, 16), AURPPO_EINVAL, "aurppo_linear_f32: x / workspace not 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    unsigned short* wop = reinterpret_cast<unsigned short*>(wop_ws);
    // filter in operand order: forward B[k][n] = w[n][k]; input gradient B[k = n_w][n = k_w] = w[k][n] (k_conv_prep, one tap)
    hipLaunchKernelGGL(k_conv_prep, dim3(64), dim3(256), 0, s, w, N_w, K_w, K, N, mode, wop, 1);
    AURPPO_LAUNCH_CHECK("k_conv_prep");
    LinArgs a;
    a.x = x; a.wop = wop; a.y = y; a.M = M; a.K = K; a.N = N;
    a.bias = bias; a.act = act;
    const long long n_mb = (M + 32 * 4 * kMB - 1) / (32 * 4 * kMB);
    const int nblk = (N + 31) / 32;
    const int NB = nblk >= 4 ? 4 : (nblk >= 2 ? 2 : 1);
    const int n_ng = (nblk + NB - 1) / NB;
    AURPPO_REQUIRE(n_mb * n_ng < (1ll << 31), AURPPO_ESHAPE, "aurppo_linear_f32: grid too large");
    a.n_mb = (int)n_mb;
    const dim3 g((unsigned)(n_mb * n_ng)), blk(kConvThreads);
    if (NB == 4) hipLaunchKernelGGL(k_linear<4>, g, blk, 0, s, a);
    else if (NB == 2) hipLaunchKernelGGL(k_linear<2>, g, blk, 0, s, a);
    else hipLaunchKernelGGL(k_linear<1>, g, blk, 0, s, a);
    AURPPO_LAUNCH_CHECK("k_linear");
    return AURPPO_OK;
}

// ---- weight gradients: the inner dimension (the minibatch's rows / the batch's output pixels) is cut into S slices, the slices'
// partial results land in the caller's workspace and k_fold_slices sums them in slice order.
// The slice count for `tiles` output tiles: the weight-gradient kernels hold two workgroups per CU (LDS), so the grid is dealt in
// rounds of 2 x 256 workgroups and a round with two workgroups in it costs what a full one does (1026 workgroups measured 611 us
// where 504 take 410): the S <= most that wastes the least of its last round, the smallest such S (fewer partial filters to fold).
static int slices_for(long long tiles, long long most) {
    constexpr long long kSlots = 2 * 256;
    if (most < 1) most = 1;
    if (most > 4096) most = 4096;
    long long best = 1;
    double best_util = 0.0;
    for (long long S = 1; S <= most && tiles * S <= 4 * kSlots; ++S) {
        const long long wgs = tiles * S, rounds = (wgs + kSlots - 1) / kSlots;
        const double util = (double)wgs / (double)(rounds * kSlots);
        if (util > best_util + 0.02) {
            best_util = util;
            best = S;
        }
    }
    return (int)best;
}
static int linear_wgrad_slices(long long M, int N, int K) {
    const long long tiles = (long long)((N + 127) / 128) * ((K + 127) / 128);
    return slices_for(tiles, (M + 255) / 256);                // at least 8 chunks of 32 rows per slice
}
extern "C" size_t aurppo_linear_wgrad_ws_bytes(long long M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (size_t)linear_wgrad_slices(M, N, K) * (size_t)N * (size_t)K * sizeof(float) + 64;
}
// dw (N, K) = dy (M, N)^T . x (M, K): nn.Linear's weight gradient (row-major fp32; N and K multiples of 4, 16-byte aligned operands)
extern "C" int aurppo_linear_wgrad_f32(const float* dy, const float* x, float* dw, long long M, int N, int K, void* ws, void* stream) {
    AURPPO_REQUIRE(dy && x && dw && ws, AURPPO_EINVAL, "aurppo_linear_wgrad_f32: null pointer");
    AURPPO_REQUIRE(M > 0 && N > 0 && K > 0 && N % 4 == 0 && K % 4 == 0, AURPPO_ESHAPE,
                   "aurppo_linear_wgrad_f32: M=%lld N=%d K=%d (N, K multiples of 4)", M, N, K);
    AURPPO_REQUIRE(aligned_to(dy, 16) && aligned_to(x, 16) && aligned_to(ws, 16), AURPPO_EINVAL,
                   "aurppo_linear_wgrad_f32: operands / workspace not 16-byte aligned");
    const int S0 = linear_wgrad_slices(M, N, K);
    WgradArgs a;
    a.dy = dy; a.x = x; a.part = reinterpret_cast<float*>(ws); a.M = M; a.N = N; a.K = K;
    const long long rps = ((M + S0 - 1) / S0 + 31) / 32 * 32;           // whole 32-row chunks per slice
    const long long S = (M + rps - 1) / rps;
    AURPPO_REQUIRE(S >= 1 && S <= S0 && rps < (1ll << 30), AURPPO_ESHAPE, "aurppo_linear_wgrad_f32: slice size");
    a.rows_per_slice = (int)rps;
    a.n_tiles_n = (N + 127) / 128;
    a.n_tiles_k = (K + 127) / 128;
    const long long grid = (long long)a.n_tiles_n * a.n_tiles_k * S;
    AURPPO_REQUIRE(grid < (1ll << 31), AURPPO_ESHAPE, "aurppo_linear_wgrad_f32: grid too large");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_linear_wgrad, dim3((unsigned)grid), dim3(kConvThreads), 0, st, a);
    AURPPO_LAUNCH_CHECK("k_linear_wgrad");
    const long long n = (long long)N * K;
    hipLaunchKernelGGL(k_fold_slices, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.part, (int)S, n, dw);
    AURPPO_LAUNCH_CHECK("k_fold_slices");
    return AURPPO_OK;
}

// K12: dw (Co, Ci, 3, 3) = the gradient of conv2d(x (B, Ci, H, W), w, padding = pad) with respect to w, given the output gradient
// dy (B, Co, H + 2 pad - 2, W + 2 pad - 2).  NCHW fp32.
namespace {
struct ConvWgradPlan {
    int WM, n_tiles_m, n_tiles_n, S, px_per_slice;
};
bool conv_wgrad_plan(int B, int Ci, int Co, int H, int W, int pad, ConvWgradPlan* p) {
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    if (B <= 0 || Ci <= 0 || Co <= 0 || Ho <= 0 || Wo <= 0 || pad < 0 || pad > 2) return false;
    const long long M = (long long)B * Ho * Wo, HoWo = (long long)Ho * Wo, HW = (long long)H * W;
    const int NK = Ci * 9;
    p->WM = Co > 64 ? 2 : 1;
    p->n_tiles_m = (Co + 64 * p->WM - 1) / (64 * p->WM);
    p->n_tiles_n = (NK + 64 * (4 / p->WM) - 1) / (64 * (4 / p->WM));
    const long long tiles = (long long)p->n_tiles_m * p->n_tiles_n;
    long long S = slices_for(tiles, (M + 511) / 512);       // at least 16 chunks of 32 pixels per slice
    long long pps = ((M + S - 1) / S + 31) / 32 * 32;
    // 32-bit offsets relative to the slice's first image: the images a slice touches must span less than 2^31 bytes of either tensor
    const long long per_img = 4 * (HoWo * Co > HW * Ci ? HoWo * Co : HW * Ci);
    if (per_img * 2 >= (1ll << 31)) return false;
    // ... and its pixel indices, counted from the first pixel of its first image, must stay exact in fp32 (the kernel's divisions)
    while (pps > 32 && ((pps / HoWo + 2) * per_img >= (1ll << 31) || pps + HoWo >= (1ll << 23))) pps = (pps / 2 + 31) / 32 * 32;
    if ((pps / HoWo + 2) * per_img >= (1ll << 31) || pps + HoWo >= (1ll << 23)) return false;
    S = (M + pps - 1) / pps;
    if (S * tiles >= (1ll << 31) || S > (1 << 20)) return false;
    p->S = (int)S;
    p->px_per_slice = (int)pps;
    return true;
}
}  // namespace

extern "C" size_t aurppo_conv3x3_wgrad_ws_bytes(int B, int Ci, int Co, int H, int W, int pad) {
    ConvWgradPlan p;
    if (!conv_wgrad_plan(B, Ci, Co, H, W, pad, &p)) return 0;
    return (size_t)p.S * (size_t)Co * (size_t)Ci * 9 * sizeof(float) + 64;
}

extern "C" int aurppo_conv3x3_wgrad_f32(const float* dy, const float* x, float* dw, int B, int Ci, int Co, int H, int W, int pad,
                                        void* ws, void* stream) {
    AURPPO_REQUIRE(dy && x && dw && ws, AURPPO_EINVAL, "aurppo_conv3x3_wgrad_f32: null pointer");
    ConvWgradPlan p;
    AURPPO_REQUIRE(conv_wgrad_plan(B, Ci, Co, H, W, pad, &p), AURPPO_ESHAPE,
                   "aurppo_conv3x3_wgrad_f32: B=%d Ci=%d Co=%d H=%d W=%d pad=%d (pad 0..2, non-empty output, one image of either tensor "
                   "under 1 GB)", B, Ci, Co, H, W, pad);
    AURPPO_REQUIRE(aligned_to(ws, 16), AURPPO_EINVAL, "aurppo_conv3x3_wgrad_f32: workspace not 16-byte aligned");
    ConvWgradArgs a;
    a.dy = dy; a.x = x; a.part = reinterpret_cast<float*>(ws);
    a.B = B; a.Ci = Ci; a.H = H; a.W = W; a.Co = Co; a.Ho = H + 2 * pad - 2; a.Wo = W + 2 * pad - 2; a.pad = pad;
    a.M = (long long)B * a.Ho * a.Wo;
    a.NK = Ci * 9;
    a.px_per_slice = p.px_per_slice;
    a.n_tiles_m = p.n_tiles_m; a.n_tiles_n = p.n_tiles_n;
    hipStream_t st = (hipStream_t)stream;
    const dim3 g((unsigned)((long long)p.n_tiles_m * p.n_tiles_n * p.S)), blk(kConvThreads);
    // a quad of output pixels stays inside one row of its image, whose 16-byte pieces are aligned
    const bool quad = a.Wo % 4 == 0 && W >= 4 && pad <= 1 && aligned_to(dy, 16);
    if (p.WM == 2) {
        if (quad) hipLaunchKernelGGL((k_conv3x3_wgrad<2, true>), g, blk, 0, st, a);
        else hipLaunchKernelGGL((k_conv3x3_wgrad<2, false>), g, blk, 0, st, a);
    } else {
        if (quad) hipLaunchKernelGGL((k_conv3x3_wgrad<1, true>), g, blk, 0, st, a);
        else hipLaunchKernelGGL((k_conv3x3_wgrad<1, false>), g, blk, 0, st, a);
    }
    AURPPO_LAUNCH_CHECK("k_conv3x3_wgrad");
    const long long n = (long long)Co * a.NK;
    hipLaunchKernelGGL(k_fold_slices, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.part, p.S, n, dw);
    AURPPO_LAUNCH_CHECK("k_fold_slices");
    return AURPPO_OK;
}
