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
