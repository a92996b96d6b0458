// Per-sample PPO loss terms and their gradients (src/ppo.py:225-264), shared by the stand-alone loss
// kernel (loss.hip) and the fused MLP step (mlp.hip).  torch autograd conventions: max() ties split
// 0.5/0.5, clamp passes gradient on the closed interval.
#pragma once
#include "common.h"

struct PpoHyper {
    int M;
    float clip, lo, hi, ent_coef, vf_coef;
    int norm_adv, vloss_mode;
};

struct PpoSample {
    float g_logp;  // d loss / d newlogp
    float g_v;     // d loss / d newv
    float pg, vl, okl, kl, cf;  // this sample's contribution to the (un-normalised) sums
};

__device__ __forceinline__ PpoSample ppo_sample(float newlogp, float oldlogp, float a_raw, float v, float vo, float R,
                                                float mean, float denom, float invM, const PpoHyper& p) {
    PpoSample o;
    const float lr = newlogp - oldlogp;
    const float ratio = expf(lr);
    const float an = p.norm_adv ? (a_raw - mean) / denom : a_raw;
    o.okl = -lr;
    o.kl = (ratio - 1.0f) - lr;
    o.cf = (fabsf(ratio - 1.0f) > p.clip) ? 1.0f : 0.0f;
    const float rc = fminf(fmaxf(ratio, p.lo), p.hi);
    const float l1 = -an * ratio;
    const float l2 = -an * rc;
    o.pg = fmaxf(l1, l2);
    const float w1 = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f);
    const float inr = (ratio >= p.lo && ratio <= p.hi) ? 1.0f : 0.0f;
    const float dpg = (w1 * (-an) + (1.0f - w1) * (-an) * inr) * invM;
    o.g_logp = dpg * ratio;
    float dvl;
    if (p.vloss_mode == AURPPO_VLOSS_CLIPPED) {
        const float du = v - R;
        const float vu = du * du;
        const float dv = v - vo;
        const float dcl = fminf(fmaxf(dv, -p.clip), p.clip);
        const float dc = (vo + dcl) - R;
        const float vc = dc * dc;
        o.vl = fmaxf(vu, vc);
        const float u1 = vu > vc ? 1.0f : (vu == vc ? 0.5f : 0.0f);
        const float inv = (dv >= -p.clip && dv <= p.clip) ? 1.0f : 0.0f;
        dvl = (u1 * (2.0f * du) + (1.0f - u1) * (2.0f * dc) * inv) * (0.5f * invM);
    } else {
        const float du = v - (p.vloss_mode == AURPPO_VLOSS_RETURNS ? R : vo);
        o.vl = du * du;
        dvl = (2.0f * du) * (0.5f * invM);
    }
    o.g_v = dvl * p.vf_coef;
    return o;
}

static inline PpoHyper make_hyper(int M, double clip, double ent_coef, double vf_coef, int norm_adv, int vloss_mode) {
    PpoHyper p;
    p.M = M;
    p.clip = (float)clip;
    p.lo = (float)(1.0 - clip);  // Python forms 1-eps / 1+eps in fp64; torch rounds them to fp32
    p.hi = (float)(1.0 + clip);
    p.ent_coef = (float)ent_coef;
    p.vf_coef = (float)vf_coef;
    p.norm_adv = norm_adv ? 1 : 0;
    p.vloss_mode = vloss_mode;
    return p;
}
