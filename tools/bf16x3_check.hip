// Stand-alone check of aur_ppo_amd/csrc/bf16x3.h on the GPU: every operand pattern k_mlp_step3 uses (LDS image layouts,
// row / transposed fragment reads, the six-product bf16 MFMA) against an fp64 host reference, plus the same product on
// v_mfma_f32_32x32x2_f32 for comparison.   hipcc --offload-arch=gfx950 -O3 tools/bf16x3_check.hip -o /tmp/bf16x3_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

#include "../aur_ppo_amd/csrc/bf16x3.h"
using namespace bf3;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ int acc_row(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }

// in: X (32 x 64), W (64 x 64) [out][in], W3 (16 x 64), dO (32 x 16), all fp32 row-major.
// out: T1 = X W^T (32x64); T2 = H W^T with H = T1 stored to an F image and read transposed (32x64);
//      T3 = dW[o][i] = sum_s T1[s][o] * H2[s][i] with H2 = T2 image (64x64); T4 = dW1[o][d] = sum_s T1[s][o] X[s][d] (64x64)
//      T5 = head: H2 W3^T (32 x 16); T6 = dH = dO W3 (32 x 64); T7 = dW3[a][i] = sum_s dO[s][a] H2[s][i] (16 x 64)
//      T8 = T1 on f32 MFMA
__global__ __launch_bounds__(64) void k_check(const float* X, const float* W, const float* W3, const float* dO, const __bf16* wop,
                                              float* T1, float* T2, float* T3, float* T4, float* T5, float* T6, float* T7, float* T8) {
    __shared__ __attribute__((aligned(16))) char ximg[3 * kXPlane];
    __shared__ __attribute__((aligned(16))) char h1img[3 * kFPlane];
    __shared__ __attribute__((aligned(16))) char h2img[3 * kFPlane];
    __shared__ __attribute__((aligned(16))) char w3img[3 * 16 * 128];
    __shared__ __attribute__((aligned(16))) char doimg[3 * 16 * 64];
    const int lane = threadIdx.x;
    // X image: lane = (row s = lane >> 1 ... ) simple loop
    for (int e = lane; e < 32 * 16; e += 64) {
        const int s = e >> 4, d0 = (e & 15) * 4;
        store_x4(ximg, s, d0, X[s * 64 + d0], X[s * 64 + d0 + 1], X[s * 64 + d0 + 2], X[s * 64 + d0 + 3]);
    }
    for (int e = lane; e < 16 * 64; e += 64) store_plain1(w3img, 128, 16 * 128, e >> 6, e & 63, W3[e]);
    for (int e = lane; e < 32 * 16; e += 64) store_plain1(doimg, 64, 16 * 64, e & 15, e >> 4, dO[e]);   // [a][s]
    __syncthreads();
    // ---- T1: both column halves
    float t1v[2][16];
    for (int cb = 0; cb < 2; ++cb) {
        f32x16v acc = {0};
        for (int ks = 0; ks < 4; ++ks) {
            const Frag3 a = x_rows(ximg, ks, lane);
            Frag3 b;
            for (int p = 0; p < 3; ++p) b.p[p] = *reinterpret_cast<const bf16x8*>(wop + wop3_index(cb, 1, ks, p, lane, 0));
            acc = mma32x3(a, b, acc);
        }
        for (int e = 0; e < 16; ++e) {
            t1v[cb][e] = acc[e];
            T1[acc_row(e, lane) * 64 + cb * 32 + (lane & 31)] = acc[e];
        }
        store_acc_f(h1img, cb * 32, t1v[cb], lane);
    }
    __syncthreads();
    {   // round trip of the image
        float back[16];
        load_acc_f(h1img, 32, back, lane);
        for (int e = 0; e < 16; ++e)
            if (back[e] != t1v[1][e]) T1[0] = NAN;
    }
    // ---- T2 = H1 W^T, A transposed-read from the F image
    for (int cb = 0; cb < 2; ++cb) {
        f32x16v acc = {0};
        for (int ks = 0; ks < 4; ++ks) {
            const Frag3 a = f_cols(h1img, ks, lane);
            Frag3 b;
            for (int p = 0; p < 3; ++p) b.p[p] = *reinterpret_cast<const bf16x8*>(wop + wop3_index(cb, 1, ks, p, lane, 0));
            acc = mma32x3(a, b, acc);
        }
        float v[16];
        for (int e = 0; e < 16; ++e) {
            v[e] = acc[e];
            T2[acc_row(e, lane) * 64 + cb * 32 + (lane & 31)] = acc[e];
        }
        store_acc_f(h2img, cb * 32, v, lane);
    }
    __syncthreads();
    // ---- T3: dW[o][i] = sum_s H1[s][o] H2[s][i]: A rows of h1img, B rows of h2img
    for (int ob = 0; ob < 2; ++ob)
        for (int cb = 0; cb < 2; ++cb) {
            f32x16v acc = {0};
            for (int ks = 0; ks < 2; ++ks) acc = mma32x3(f_rows(h1img, ob * 32, ks, lane), f_rows(h2img, cb * 32, ks, lane), acc);
            for (int e = 0; e < 16; ++e) T3[(ob * 32 + acc_row(e, lane)) * 64 + cb * 32 + (lane & 31)] = acc[e];
        }
    // ---- T4: dW1[o][d] = sum_s H1[s][o] X[s][d]: A rows of h1img, B transposed from the X image
    for (int ob = 0; ob < 2; ++ob)
        for (int cb = 0; cb < 2; ++cb) {
            f32x16v acc = {0};
            for (int ks = 0; ks < 2; ++ks) acc = mma32x3(f_rows(h1img, ob * 32, ks, lane), x_cols(ximg, ks, cb * 32, lane), acc);
            for (int e = 0; e < 16; ++e) T4[(ob * 32 + acc_row(e, lane)) * 64 + cb * 32 + (lane & 31)] = acc[e];
        }
    // ---- T5: head out[s][a] = sum_i H2[s][i] W3[a][i], 16 rows per half
    for (int half = 0; half < 2; ++half) {
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < 2; ++ks)
            acc = mma16x3(f_cols16(h2img, 16 * half, ks, lane), plain_rows(w3img, 128, 16 * 128, lane & 15, 32 * ks + 8 * (lane >> 4)), acc);
        for (int e = 0; e < 4; ++e) T5[(16 * half + 4 * (lane >> 4) + e) * 16 + (lane & 15)] = acc[e];
    }
    // ---- T6: dH[s][i] = sum_a dO[s][a] W3[a][i]: A transposed from the [a][s] image, B transposed from the [a][i] image
    for (int cb = 0; cb < 2; ++cb) {
        f32x16v acc = {0};
        acc = mma32x3(plain_cols(doimg, 64, 16 * 64, 0, 0, lane), plain_cols(w3img, 128, 16 * 128, 0, cb * 32, lane), acc);
        for (int e = 0; e < 16; ++e) T6[acc_row(e, lane) * 64 + cb * 32 + (lane & 31)] = acc[e];
    }
    // ---- T7: dW3[a][i] = sum_s dO[s][a] H2[s][i] (16x16x32, one k-step of 32 samples), four 16-column blocks
    for (int blk = 0; blk < 4; ++blk) {
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma16x3(plain_rows(doimg, 64, 16 * 64, lane & 15, 8 * (lane >> 4)), f_rows16(h2img, 16 * blk, lane), acc);
        for (int e = 0; e < 4; ++e) T7[(4 * (lane >> 4) + e) * 64 + 16 * blk + (lane & 15)] = acc[e];
    }
    // ---- T8: T1 on the f32 MFMA
    for (int cb = 0; cb < 2; ++cb) {
        f32x16v acc = {0};
        for (int k = 0; k < 64; k += 2) {
            const float a = X[(lane & 31) * 64 + k + (lane >> 5)], b = W[(cb * 32 + (lane & 31)) * 64 + k + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        for (int e = 0; e < 16; ++e) T8[acc_row(e, lane) * 64 + cb * 32 + (lane & 31)] = acc[e];
    }
}

static unsigned short host_bf16(float x) {   // round to nearest even
    unsigned u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static double report(const char* name, const std::vector<float>& got, const std::vector<double>& ref, const std::vector<double>& mag) {
    double worst = 0, scale = 0;
    for (size_t i = 0; i < ref.size(); ++i) scale = fmax(scale, fabs(ref[i]));
    double worst_rel_mag = 0;
    for (size_t i = 0; i < ref.size(); ++i) {
        const double e = fabs((double)got[i] - ref[i]);
        worst = fmax(worst, e);
        worst_rel_mag = fmax(worst_rel_mag, e / mag[i]);
    }
    printf("%-28s max|err| %.3e  (%.2e of max |value| %.3g; %.2e of sum|a*b|)\n", name, worst, worst / scale, scale, worst_rel_mag);
    return worst_rel_mag;
}

int main() {
    srand(7);
    auto rnd = [] { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0); };
    std::vector<float> X(32 * 64), W(64 * 64), W3(16 * 64), dO(32 * 16);
    for (auto& v : X) v = rnd() * 3.0f;
    for (auto& v : W) v = rnd() * 0.4f;
    for (auto& v : W3) v = rnd() * 0.2f;
    for (auto& v : dO) v = rnd() * 1e-5f;       // gradients are small numbers
    // operand-order weight planes (mt 1 = forward W2-style copy of W for both roles 0, 1)
    std::vector<unsigned short> wop(kWopElems, 0);
    for (int o = 0; o < 64; ++o)
        for (int i = 0; i < 64; ++i) {
            int idx[2];
            wop3_places(0, 1, o, i, idx);
            float a = W[o * 64 + i];
            unsigned short p0 = host_bf16(a);
            float r = a - bf16_f(p0);
            unsigned short p1 = host_bf16(r);
            unsigned short p2 = host_bf16(r - bf16_f(p1));
            wop[idx[0]] = p0; wop[idx[0] + kWopBlock] = p1; wop[idx[0] + 2 * kWopBlock] = p2;
        }
    float *dX, *dW, *dW3, *ddO, *dT[8];
    __bf16* dwop;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dW3, W3.size() * 4)); CK(hipMalloc(&ddO, dO.size() * 4));
    CK(hipMalloc(&dwop, wop.size() * 2));
    const size_t tn[8] = {32 * 64, 32 * 64, 64 * 64, 64 * 64, 32 * 16, 32 * 64, 16 * 64, 32 * 64};
    for (int t = 0; t < 8; ++t) { CK(hipMalloc(&dT[t], tn[t] * 4)); CK(hipMemset(dT[t], 0, tn[t] * 4)); }
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW3, W3.data(), W3.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ddO, dO.data(), dO.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwop, wop.data(), wop.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dX, dW, dW3, ddO, dwop, dT[0], dT[1], dT[2], dT[3], dT[4], dT[5], dT[6], dT[7]);
    CK(hipDeviceSynchronize());
    std::vector<float> T[8];
    for (int t = 0; t < 8; ++t) { T[t].resize(tn[t]); CK(hipMemcpy(T[t].data(), dT[t], tn[t] * 4, hipMemcpyDeviceToHost)); }
    // references.  Later stages take the GPU's own fp32 outputs of the earlier ones as inputs (what the images hold).
    std::vector<double> R1(32 * 64), M1(32 * 64), R2(32 * 64), M2(32 * 64), R3(64 * 64), M3(64 * 64), R4(64 * 64), M4(64 * 64),
        R5(32 * 16), M5(32 * 16), R6(32 * 64), M6(32 * 64), R7(16 * 64), M7(16 * 64);
    for (int s = 0; s < 32; ++s)
        for (int o = 0; o < 64; ++o) {
            double a = 0, m = 0, a2 = 0, m2 = 0;
            for (int k = 0; k < 64; ++k) {
                a += (double)X[s * 64 + k] * W[o * 64 + k]; m += fabs((double)X[s * 64 + k] * W[o * 64 + k]);
                a2 += (double)T[0][s * 64 + k] * W[o * 64 + k]; m2 += fabs((double)T[0][s * 64 + k] * W[o * 64 + k]);
            }
            R1[s * 64 + o] = a; M1[s * 64 + o] = m; R2[s * 64 + o] = a2; M2[s * 64 + o] = m2;
        }
    for (int o = 0; o < 64; ++o)
        for (int i = 0; i < 64; ++i) {
            double a = 0, m = 0, a4 = 0, m4 = 0;
            for (int s = 0; s < 32; ++s) {
                a += (double)T[0][s * 64 + o] * T[1][s * 64 + i]; m += fabs((double)T[0][s * 64 + o] * T[1][s * 64 + i]);
                a4 += (double)T[0][s * 64 + o] * X[s * 64 + i]; m4 += fabs((double)T[0][s * 64 + o] * X[s * 64 + i]);
            }
            R3[o * 64 + i] = a; M3[o * 64 + i] = m; R4[o * 64 + i] = a4; M4[o * 64 + i] = m4;
        }
    for (int s = 0; s < 32; ++s)
        for (int a = 0; a < 16; ++a) {
            double v = 0, m = 0;
            for (int i = 0; i < 64; ++i) { v += (double)T[1][s * 64 + i] * W3[a * 64 + i]; m += fabs((double)T[1][s * 64 + i] * W3[a * 64 + i]); }
            R5[s * 16 + a] = v; M5[s * 16 + a] = m;
        }
    for (int s = 0; s < 32; ++s)
        for (int i = 0; i < 64; ++i) {
            double v = 0, m = 0;
            for (int a = 0; a < 16; ++a) { v += (double)dO[s * 16 + a] * W3[a * 64 + i]; m += fabs((double)dO[s * 16 + a] * W3[a * 64 + i]); }
            R6[s * 64 + i] = v; M6[s * 64 + i] = m;
        }
    for (int a = 0; a < 16; ++a)
        for (int i = 0; i < 64; ++i) {
            double v = 0, m = 0;
            for (int s = 0; s < 32; ++s) { v += (double)dO[s * 16 + a] * T[1][s * 64 + i]; m += fabs((double)dO[s * 16 + a] * T[1][s * 64 + i]); }
            R7[a * 64 + i] = v; M7[a * 64 + i] = m;
        }
    double w = 0;
    w = fmax(w, report("T1 X W^T (rows, regs)", T[0], R1, M1));
    w = fmax(w, report("T2 H W^T (F image, tr)", T[1], R2, M2));
    w = fmax(w, report("T3 dW = H1^T H2 (rows,rows)", T[2], R3, M3));
    w = fmax(w, report("T4 dW1 = H1^T X (rows, tr)", T[3], R4, M4));
    w = fmax(w, report("T5 head 16x16x32 (tr, rows)", T[4], R5, M5));
    w = fmax(w, report("T6 dH = dO W3 (tr, tr)", T[5], R6, M6));
    w = fmax(w, report("T7 dW3 16x16x32 (rows,rows)", T[6], R7, M7));
    const double f32err = report("T8 X W^T on f32 MFMA", T[7], R1, M1);
    printf("worst bf16x3 error / sum|a*b| = %.3e   (f32 MFMA: %.3e)\n", w, f32err);
    const bool ok = w < 4e-7 && !isnan(T[0][0]);
    printf(ok ? "BF16X3_CHECK_OK\n" : "BF16X3_CHECK_FAILED\n");
    return ok ? 0 : 1;
}
