/*
 * aurppo_oracle.c -- plain-C CPU restatement of the hot path.  TEST INFRASTRUCTURE ONLY
 * (see oracle/__init__.py): loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg, never by the product.  Same argument meaning as include/aurppo.h, host pointers.
 *
 * Parity status: pinned -- tests/test_oracle_golden.py checks every function against vectors
 * captured from the real reference (tests/golden/) and, for the shuffle, against numpy itself.
 *
 * Build:  make -C oracle      (gcc -O2 -ffp-contract=off: no fused multiply-add, like torch CPU)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- a4/a5: src/ppo.py:125-157, src/robot_ppo.py:224-244 ------------------------------- */
int oracle_gae_f32(const float* r, const float* v, const float* done, const float* next_value,
                   const float* next_done, float* adv, float* ret, int T, int N, double gamma, double lam, int mode) {
    const float g = (float)gamma, gl = (float)(gamma * lam);
    for (int n = 0; n < N; ++n) {
        float x = (mode == 1) ? next_value[n] : 0.0f;
        for (int t = T - 1; t >= 0; --t) {
            const size_t at = (size_t)t * N + n;
            const float nd = (t == T - 1) ? next_done[n] : done[at + N];
            const float nv = (t == T - 1) ? next_value[n] : v[at + N];
            const float nnt = 1.0f - nd;
            if (mode == 1) {
                x = r[at] + (g * nnt) * x; /* ppo.py:155 */
                ret[at] = x;
                adv[at] = x - v[at]; /* ppo.py:156 */
            } else {
                if (mode == 2 && t == T - 1) { /* robot_ppo.py:230: loop starts at T-2 */
                    x = 0.0f;
                } else {
                    const float delta = (r[at] + (g * nv) * nnt) - v[at]; /* ppo.py:139 */
                    x = delta + (gl * nnt) * x;                            /* ppo.py:140 */
                }
                adv[at] = x;
                ret[at] = x + v[at]; /* ppo.py:141 */
            }
        }
    }
    return 0;
}

/* ---- a6: numpy legacy RandomState (init_genrand, genrand_int32, random_interval, shuffle) --- */
typedef struct {
    uint32_t key[624];
    int pos;
} oracle_mt;

void oracle_mt_seed(oracle_mt* s, uint32_t seed) {
    s->key[0] = seed;
    for (int i = 1; i < 624; ++i) s->key[i] = 1812433253u * (s->key[i - 1] ^ (s->key[i - 1] >> 30)) + (uint32_t)i;
    s->pos = 624;
}

static uint32_t mt_next(oracle_mt* s) {
    if (s->pos == 624) {
        uint32_t* mt = s->key;
        for (int k = 0; k < 624; ++k) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->pos = 0;
    }
    uint32_t y = s->key[s->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

void oracle_mt_shuffle_i32(oracle_mt* s, int32_t* x, int n) {
    for (int i = n - 1; i >= 1; --i) {
        uint32_t mask = (uint32_t)i;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        uint32_t j;
        do {
            j = mt_next(s) & mask;
        } while (j > (uint32_t)i);
        const int32_t t = x[i];
        x[i] = x[j];
        x[j] = t;
    }
}

void oracle_mt_get_state(const oracle_mt* s, uint32_t* key, int32_t* pos) {
    memcpy(key, s->key, sizeof(s->key));
    *pos = s->pos;
}

size_t oracle_mt_sizeof(void) { return sizeof(oracle_mt); }

/* ---- a7: b_x[mb_inds]  (src/ppo.py:219-220,225,236,251-257) ------------------------------- */
int oracle_gather_f32(const int32_t* idx, int M, const float* src, float* dst, int row_elems) {
    for (int m = 0; m < M; ++m)
        memcpy(dst + (size_t)m * row_elems, src + (size_t)idx[m] * row_elems, 4u * (size_t)row_elems);
    return 0;
}

/* ---- a9/a10: src/ppo.py:225-264 (value branches: ppo.py:250-261, robot_ppo.py:390) ---------- */
int oracle_loss_fwd_bwd_f32(const float* newlogp, const float* oldlogp, const float* adv, const float* newv,
                            const float* oldv, const float* ret, const float* entropy, int M, double clip,
                            double ent_coef, double vf_coef, int norm_adv, int vloss_mode, float* out,
                            float* g_newlogp, float* g_newv, float* g_entropy) {
    const float c = (float)clip, lo = (float)(1.0 - clip), hi = (float)(1.0 + clip);
    const float invM = 1.0f / (float)M;
    double s = 0, q = 0;
    for (int i = 0; i < M; ++i) {
        s += adv[i];
        q += (double)adv[i] * adv[i];
    }
    const double m = s / M;
    double var = (q - s * m) / (double)(M - 1);
    if (var < 0) var = 0;
    const float mean = (float)m, std = (float)sqrt(var);
    double a_pg = 0, a_vl = 0, a_ent = 0, a_okl = 0, a_kl = 0, a_cf = 0;
    for (int i = 0; i < M; ++i) {
        const float lr = newlogp[i] - oldlogp[i];
        const float ratio = expf(lr);
        const float an = norm_adv ? (adv[i] - mean) / (std + 1e-8f) : adv[i];
        a_okl += -lr;
        a_kl += (ratio - 1.0f) - lr;
        a_cf += fabsf(ratio - 1.0f) > c;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float l1 = -an * ratio, l2 = -an * rc;
        a_pg += fmaxf(l1, l2);
        const float w1 = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f);
        const float inr = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
        g_newlogp[i] = ((w1 * (-an) + (1.0f - w1) * (-an) * inr) * invM) * ratio;
        float dvl;
        if (vloss_mode == 1) {
            const float du = newv[i] - ret[i], vu = du * du, dv = newv[i] - oldv[i];
            const float dc = (oldv[i] + fminf(fmaxf(dv, -c), c)) - ret[i], vc = dc * dc;
            a_vl += fmaxf(vu, vc);
            const float u1 = vu > vc ? 1.0f : (vu == vc ? 0.5f : 0.0f);
            const float in = (dv >= -c && dv <= c) ? 1.0f : 0.0f;
            dvl = (u1 * (2.0f * du) + (1.0f - u1) * (2.0f * dc) * in) * (0.5f * invM);
        } else {
            const float du = newv[i] - (vloss_mode == 0 ? ret[i] : oldv[i]);
            a_vl += du * du;
            dvl = (2.0f * du) * (0.5f * invM);
        }
        g_newv[i] = dvl * (float)vf_coef;
        a_ent += entropy[i];
        g_entropy[i] = -(float)ent_coef * invM;
    }
    const float pg = (float)(a_pg / M), vl = 0.5f * (float)(a_vl / M), ent = (float)(a_ent / M);
    out[0] = (pg - (float)ent_coef * ent) + vl * (float)vf_coef;
    out[1] = pg;
    out[2] = vl;
    out[3] = ent;
    out[4] = (float)(a_okl / M);
    out[5] = (float)(a_kl / M);
    out[6] = (float)(a_cf / M);
    out[7] = mean;
    out[8] = std;
    return 0;
}

/* ---- a11: nn.utils.clip_grad_norm_ (src/ppo.py:268) ----------------------------------------- */
int oracle_grad_norm_clip_f32(float* g, int64_t n, double max_norm, float* out_norm) {
    double q = 0;
    for (int64_t i = 0; i < n; ++i) q += (double)g[i] * g[i];
    const float norm = (float)sqrt(q);
    float coef = (float)max_norm / (norm + 1e-6f);
    if (coef > 1.0f) coef = 1.0f;
    for (int64_t i = 0; i < n; ++i) g[i] *= coef;
    *out_norm = norm;
    return 0;
}
