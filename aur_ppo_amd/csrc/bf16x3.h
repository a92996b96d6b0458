// fp32 matrix products on the bf16 matrix pipe: every fp32 operand is held as THREE bf16 planes
//     a = a0 + a1 + a2,   a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)      (exact: 8 + 8 + 8 >= 24 bits)
// and a product a*b is formed as the six bf16 products whose weight is >= 2^-24 of it
//     a0*b0 + (a0*b1 + a1*b0) + (a0*b2 + a2*b0 + a1*b1)
// (dropped: a1*b2, a2*b1, a2*b2 <= 2^-25 |a*b|), each exact in the MFMA's fp32 accumulator.  Six
// v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles, K = 2): 192 instead of
// 512 matrix-pipe cycles per 32 x 32 x 16 block, at fp32 accuracy (tools/bf16x3_check.hip measures both).
//
// This header holds what k_mlp_step3 (mlp3.hip) and that check program share: the split, the two LDS image layouts and
// the per-lane fragment addresses of every operand pattern the step needs.
//
// LDS images (bf16, one per plane):
//   X image  [row s: 32][col d: 64], 128-B rows  -- observations as they arrive from HBM (a lane holds 4 consecutive d);
//   F image  [row f: 64][col s: 32],  64-B rows  -- activations / their gradients, FEATURE-major, because a 32x32 MFMA
//            accumulator has its column (feature) on the lane and 4 consecutive rows (samples) in 4 registers: one
//            8-byte store per 4 values.
// Both are read two ways: along the row with ds_read_b128 (8 consecutive columns = one operand fragment) and across
// rows with ds_read_b64_tr_b16 (4 rows x 16 columns, delivered transposed; two of them = one fragment).  16-byte chunks
// are XOR-swizzled inside a row so that both kinds of read -- and the stores -- are bank-conflict free:
//   b128 serves 16 lanes per LDS cycle, lanes {0-3,12-15,20-27} / {4-11,16-19,28-31} (+32): their rows must spread
//   over all sixteen 16-B slots of the 256-B bank space; a transposed read serves 32 lanes (4 rows x 64 B) per cycle:
//   its four rows must fall in four different 64-B quarters.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bf3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct Frag3 {   // one operand fragment (8 k-values per lane) in its three planes
    bf16x8 p[3];
};

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};     // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// (a, b) -> three packed pairs {plane k of a in the low half, of b in the high half}
__device__ __forceinline__ void split3(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pack_bf16(a, b);
#ifdef K7_EXP_CHEAP_SPLIT    // timing experiment (WRONG results; tools/k7_experiments.sh): 3 of the split's 11 vector instructions
    p1 = p0;
    p2 = p0;
    return;
#endif
    const float ra = a - bf_lo(p0), rb = b - bf_hi(p0);
    p1 = pack_bf16(ra, rb);
    p2 = pack_bf16(ra - bf_lo(p1), rb - bf_hi(p1));
}
// the value three packed planes stand for (exact)
__device__ __forceinline__ float join_lo(unsigned p0, unsigned p1, unsigned p2) { return (bf_lo(p0) + bf_lo(p1)) + bf_lo(p2); }
__device__ __forceinline__ float join_hi(unsigned p0, unsigned p1, unsigned p2) { return (bf_hi(p0) + bf_hi(p1)) + bf_hi(p2); }

constexpr int kXRow = 128, kXPlane = 32 * kXRow;     // X image: bytes per row / per plane
constexpr int kFRow = 64, kFPlane = 64 * kFRow;      // F image

// ---- byte offsets inside one plane
__device__ __forceinline__ int xswz(int s) { return (((s >> 1) & 1) << 2) | (((s >> 2) ^ (s >> 3)) & 3); }
__device__ __forceinline__ int xoff(int s, int d) { return s * kXRow + ((((d >> 3) ^ xswz(s)) & 7) << 4) + ((d & 7) << 1); }
__device__ __forceinline__ int foff(int f, int s) { return f * kFRow + ((((s >> 3) ^ (f >> 2)) & 3) << 4) + ((s & 7) << 1); }

__device__ __forceinline__ bf16x8 lds_b128(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ s16x4 lds_tr(const char* p) {      // ds_read_b64_tr_b16; EXEC must be all ones (every lane addresses)
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<char*>(p)));
}
__device__ __forceinline__ bf16x8 join_tr(s16x4 a, s16x4 b) {
    union { bf16x8 v; s16x4 h[2]; } u;
    u.h[0] = a;
    u.h[1] = b;
    return u.v;
}

// lane coordinates of a transposed read: group g = lane >> 4 takes the 4 x 16 block whose rows are this lane's k-values
struct TrLane {
    int kq;    // row inside the fragment's 8 k-values that this lane ADDRESSES: 8 * (g >> 1) + q      (+ 4t for read t)
    int m0;    // first of the 4 columns this lane addresses: 16 * (g & 1) + 4 * p
};
__device__ __forceinline__ TrLane tr_lane32(int lane) {      // 32x32x16 operands: lane l holds k = 8 (l >> 5) + j at m = l & 31
    const int g = lane >> 4, i = lane & 15;
    return {8 * (g >> 1) + (i >> 2), 16 * (g & 1) + 4 * (i & 3)};
}
__device__ __forceinline__ TrLane tr_lane16(int lane) {      // 16x16x32 operands: lane l holds k = 8 (l >> 4) + j at m = l & 15
    const int g = lane >> 4, i = lane & 15;
    return {8 * g + (i >> 2), 4 * (i & 3)};
}

// ---- fragments.  `img` = plane 0 of the image (bytes), planes follow at `plane` bytes.
// A[m = s][k = d] (32x32x16) from an X image, k-step ks: row read
// (Every fragment address below is written as  <k-step 0's address of this lane>  +/^  <a constant of the k-step>: the swizzles are
// XORs of a few address bits, so a k-step moves the address by a constant -- xoff(s, d ^ c) = xoff(s, d) ^ (c << 1) for c a multiple
// of 8, foff(f + 16 k, s) = foff(f, s) + 1024 k, foff(f, s ^ 16) = foff(f, s) ^ 32 -- and written this way the lane-dependent part
// is formed once per chain and the k-step's part is an immediate or one XOR, where the plain form cost ~8 vector instructions per
// k-step to re-derive (~500 of k_mlpw3_step's 3 040 per tile).)
__device__ __forceinline__ Frag3 x_rows(const char* img, int ks, int lane) {
    const int o = xoff(lane & 31, 8 * (lane >> 5)) ^ (ks << 5);              // = xoff(lane & 31, 16 ks + 8 (lane >> 5)), ks < 4
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = lds_b128(img + p * kXPlane + o);
    return f;
}
// B[k = s][n = d0 + 0..31] (32x32x16) from an X image, k-step ks (16 samples): transposed read
__device__ __forceinline__ Frag3 x_cols(const char* img, int ks, int d0, int lane) {
    const TrLane t = tr_lane32(lane);
    const int d = d0 + t.m0;
    // xoff(16 ks + q, d) = 16 ks * kXRow + (xoff(q, d) ^ ((ks & 1) << 5)): xswz(16 ks + q) = xswz(q) ^ (2 (ks & 1)) for q < 16
    const int o0 = (xoff(t.kq, d) ^ ((ks & 1) << 5)) + 16 * ks * kXRow, o1 = (xoff(t.kq + 4, d) ^ ((ks & 1) << 5)) + 16 * ks * kXRow;
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = join_tr(lds_tr(img + p * kXPlane + o0), lds_tr(img + p * kXPlane + o1));
    return f;
}
// A[m = f0 + 0..31][k = s] or B[k = s][n = f0 + 0..31] (32x32x16) from an F image, k-step ks (16 samples): row read
__device__ __forceinline__ Frag3 f_rows(const char* img, int f0, int ks, int lane) {       // f0 a multiple of 16, ks < 2
    const int o = (foff(lane & 31, 8 * (lane >> 5)) ^ (ks << 5)) + f0 * kFRow;      // = foff(f0 + (lane & 31), 16 ks + 8 (lane >> 5))
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = lds_b128(img + p * kFPlane + o);
    return f;
}
// A[m = s][k = f] (32x32x16) from an F image, k-step ks (16 features): transposed read
__device__ __forceinline__ Frag3 f_cols(const char* img, int ks, int lane) {
    const TrLane t = tr_lane32(lane);
    const int o0 = foff(t.kq, t.m0) + 16 * ks * kFRow, o1 = foff(t.kq + 4, t.m0) + 16 * ks * kFRow;      // = foff(16 ks + kq (+ 4), m0): kq + 4 < 16
    Frag3 r;
#pragma unroll
    for (int p = 0; p < 3; ++p) r.p[p] = join_tr(lds_tr(img + p * kFPlane + o0), lds_tr(img + p * kFPlane + o1));
    return r;
}
// 16x16x32: A[m = s0 + 0..15][k = f] from an F image, k-step ks (32 features): transposed read
__device__ __forceinline__ Frag3 f_cols16(const char* img, int s0, int ks, int lane) {
    const TrLane t = tr_lane16(lane);
    const int o0 = foff(t.kq, s0 + t.m0) + 32 * ks * kFRow, o1 = foff(t.kq + 4, s0 + t.m0) + 32 * ks * kFRow;      // kq + 4 < 32
    Frag3 r;
#pragma unroll
    for (int p = 0; p < 3; ++p) r.p[p] = join_tr(lds_tr(img + p * kFPlane + o0), lds_tr(img + p * kFPlane + o1));
    return r;
}
// 16x16x32: B[k = s][n = f0 + 0..15] from an F image (all 32 samples = one k-step): row read
__device__ __forceinline__ Frag3 f_rows16(const char* img, int f0, int lane) {
    const int o = foff(f0 + (lane & 15), 8 * (lane >> 4));
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = lds_b128(img + p * kFPlane + o);
    return f;
}

// Small unswizzled images: `rowb` bytes per row, `planeb` bytes per plane.
// row read for 32x32x16 (k = 8 (l >> 5) + j) / 16x16x32 (k = 8 (l >> 4) + j): 8 consecutive columns from column c0 of row r
__device__ __forceinline__ Frag3 plain_rows(const char* img, int rowb, int planeb, int r, int c0) {
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = lds_b128(img + p * planeb + r * rowb + 2 * c0);
    return f;
}
// transposed read, 32x32x16: k = rows k0 + 0..15 of the image, m / n = columns c0 + 0..31
__device__ __forceinline__ Frag3 plain_cols(const char* img, int rowb, int planeb, int k0, int c0, int lane) {
    const TrLane t = tr_lane32(lane);
    const int o0 = (k0 + t.kq) * rowb + 2 * (c0 + t.m0), o1 = o0 + 4 * rowb;
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = join_tr(lds_tr(img + p * planeb + o0), lds_tr(img + p * planeb + o1));
    return f;
}

// ---- the six products
__device__ __forceinline__ f32x16v mma32x3(const Frag3& a, const Frag3& b, f32x16v c) {
#ifndef K7_EXP_HALF_MFMA     // timing experiment (WRONG results; tools/k7_experiments.sh): three of the six products
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], b.p[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[1], c, 0, 0, 0);
#endif
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[0], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ f32x4v mma16x3(const Frag3& a, const Frag3& b, f32x4v c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[2], b.p[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[0], c, 0, 0, 0);
    return c;
}

// ---- stores
// a 32x32 accumulator block (column f = f0 + (lane & 31) on the lane, rows s = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)) into an
// F image: per plane four 8-byte stores of 4 consecutive samples
__device__ __forceinline__ void store_acc_f(char* img, int f0, const float (&v)[16], int lane) {
    const int f = f0 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        unsigned a0, a1, a2, b0, b1, b2;
        split3(v[4 * gq + 0], v[4 * gq + 1], a0, a1, a2);
        split3(v[4 * gq + 2], v[4 * gq + 3], b0, b1, b2);
        const int o = foff(f, 4 * h) ^ (gq << 4);       // = foff(f, 8 gq + 4 h)
        *reinterpret_cast<u32x2*>(img + 0 * kFPlane + o) = u32x2{a0, b0};
        *reinterpret_cast<u32x2*>(img + 1 * kFPlane + o) = u32x2{a1, b1};
        *reinterpret_cast<u32x2*>(img + 2 * kFPlane + o) = u32x2{a2, b2};
    }
}
// the same 16 values back (exact): what store_acc_f wrote, as fp32
__device__ __forceinline__ void load_acc_f(const char* img, int f0, float (&v)[16], int lane) {
    const int f = f0 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int o = foff(f, 4 * h) ^ (gq << 4);       // = foff(f, 8 gq + 4 h)
        const u32x2 q0 = *reinterpret_cast<const u32x2*>(img + 0 * kFPlane + o);
        const u32x2 q1 = *reinterpret_cast<const u32x2*>(img + 1 * kFPlane + o);
        const u32x2 q2 = *reinterpret_cast<const u32x2*>(img + 2 * kFPlane + o);
        v[4 * gq + 0] = join_lo(q0.x, q1.x, q2.x);
        v[4 * gq + 1] = join_hi(q0.x, q1.x, q2.x);
        v[4 * gq + 2] = join_lo(q0.y, q1.y, q2.y);
        v[4 * gq + 3] = join_hi(q0.y, q1.y, q2.y);
    }
}
// four consecutive columns d0 .. d0+3 (d0 % 4 == 0) of row s of an X image
__device__ __forceinline__ void store_x4(char* img, int s, int d0, float x0, float x1, float x2, float x3) {
    unsigned a0, a1, a2, b0, b1, b2;
    split3(x0, x1, a0, a1, a2);
    split3(x2, x3, b0, b1, b2);
    const int o = xoff(s, d0);
    *reinterpret_cast<u32x2*>(img + 0 * kXPlane + o) = u32x2{a0, b0};
    *reinterpret_cast<u32x2*>(img + 1 * kXPlane + o) = u32x2{a1, b1};
    *reinterpret_cast<u32x2*>(img + 2 * kXPlane + o) = u32x2{a2, b2};
}
// one element (2-byte stores): X image / small unswizzled image
__device__ __forceinline__ void store_x1(char* img, int s, int d, float x) {
    unsigned p0, p1, p2;
    split3(x, 0.0f, p0, p1, p2);
    const int o = xoff(s, d);
    *reinterpret_cast<unsigned short*>(img + 0 * kXPlane + o) = (unsigned short)p0;
    *reinterpret_cast<unsigned short*>(img + 1 * kXPlane + o) = (unsigned short)p1;
    *reinterpret_cast<unsigned short*>(img + 2 * kXPlane + o) = (unsigned short)p2;
}
__device__ __forceinline__ void store_plain1(char* img, int rowb, int planeb, int r, int c, float x) {
    unsigned p0, p1, p2;
    split3(x, 0.0f, p0, p1, p2);
    const int o = r * rowb + 2 * c;
    *reinterpret_cast<unsigned short*>(img + 0 * planeb + o) = (unsigned short)p0;
    *reinterpret_cast<unsigned short*>(img + 1 * planeb + o) = (unsigned short)p1;
    *reinterpret_cast<unsigned short*>(img + 2 * planeb + o) = (unsigned short)p2;
}

// ---- weights as B operands streamed from global memory in operand order (k_mlp3_prep / k_adam_chain lay them out):
// block (role wi = net * 2 + cb, matrix mt, k-step ks, plane p) = 64 lanes x 8 bf16; lane l, element j holds
//   mt 0 (W1, forward):  W1[cb*32 + (l & 31)][16 ks + 8 (l >> 5) + j]      (zero beyond D)
//   mt 1 (W2, forward):  W2[cb*32 + (l & 31)][16 ks + 8 (l >> 5) + j]
//   mt 2 (W2, backward): W2[16 ks + 8 (l >> 5) + j][cb*32 + (l & 31)]
constexpr int kWopMats = 3, kWopKs = 4;
constexpr int kWopBlock = 64 * 8;                                           // bf16 elements
constexpr int kWopElems = 4 * kWopMats * kWopKs * 3 * kWopBlock;            // 4 roles
__device__ __host__ __forceinline__ int wop3_index(int wi, int mt, int ks, int p, int lane, int j) {
    return ((((wi * kWopMats + mt) * kWopKs + ks) * 3 + p) * 64 + lane) * 8 + j;
}
// where element (row, col) of W1 / W2 of net `net` lands: fills idx[] with the bf16 indices of plane 0 (planes 1, 2 follow at
// + kWopBlock each) and returns how many (1 for W1, 2 for W2: forward and backward copy)
__device__ __host__ __forceinline__ int wop3_places(int net, int is_w2, int row, int col, int (&idx)[2]) {
    // forward copy: out = row, in = col
    {
        const int cb = row >> 5, c = row & 31, ks = col >> 4, h = (col >> 3) & 1, j = col & 7;
        idx[0] = wop3_index(net * 2 + cb, is_w2 ? 1 : 0, ks, 0, c + 32 * h, j);
    }
    if (!is_w2) return 1;
    {   // backward copy: k = out = row, n = in = col
        const int cb = col >> 5, c = col & 31, ks = row >> 4, h = (row >> 3) & 1, j = row & 7;
        idx[1] = wop3_index(net * 2 + cb, 2, ks, 0, c + 32 * h, j);
    }
    return 2;
}

}  // namespace bf3
