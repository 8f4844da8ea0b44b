// Per-sample distribution and loss arithmetic shared by the loss kernels (loss.hip) and the fused MLP step
// (mlp_fused.hip).  Reference arithmetic restated here:
//   distributions.py:18-29  + torch/distributions/categorical.py:78,150-162   (Categorical)
//   distributions.py:36-41,86-95 + torch/distributions/normal.py:88-103,115-116 (DiagGaussian)
//   algo/ppo.py:61-81, algo/a2c_acktr.py:48-51,56-65,71-72 (losses; backward = what autograd derives,
//   including torch's "ties split the gradient" rule of minimum / maximum)
#pragma once
#include "common.hpp"

namespace acktr {

constexpr float HALF_LOG_2PI = 0.91893853320467274178f;   // log(sqrt(2*pi))

struct DistEval {
    float logp, ent;
};

// ---- categorical -------------------------------------------------------------------------
__device__ __forceinline__ DistEval categorical_eval(const float* z, int A, long a, float* lse_out) {
    float m = z[0];
    for (int i = 1; i < A; ++i) m = fmaxf(m, z[i]);
    float s = 0.f;
    for (int i = 0; i < A; ++i) s += expf(z[i] - m);
    const float lse = m + logf(s);
    float ent = 0.f;
    for (int i = 0; i < A; ++i) {
        const float lp = fmaxf(z[i] - lse, -3.402823466e+38f);   // clamp(logits, min=finfo.min)
        ent -= lp * expf(lp);
    }
    DistEval r;
    r.logp = (a >= 0 && a < A) ? z[a] - lse : 0.f;
    r.ent = ent;
    *lse_out = lse;
    return r;
}
// dz_i = dlogp*(delta_ia - p_i) - dent*p_i*(logp_i + H)
__device__ __forceinline__ void categorical_bwd(const float* z, int A, long a, float lse, float ent, float dlogp,
                                                float dent, float* dz) {
    for (int i = 0; i < A; ++i) {
        const float lp = z[i] - lse, p = expf(lp);
        dz[i] = dlogp * ((i == a ? 1.f : 0.f) - p) - dent * p * (lp + ent);
    }
}

// ---- diagonal gaussian -------------------------------------------------------------------
__device__ __forceinline__ DistEval gaussian_eval(const float* mu, const float* logstd, const float* act, int A) {
    DistEval r;
    r.logp = 0.f;
    r.ent = 0.f;
    for (int j = 0; j < A; ++j) {
        const float sd = expf(logstd[j]);
        const float var = sd * sd, ls = logf(sd), d = act[j] - mu[j];
        r.logp += -(d * d) / (2.f * var) - ls - HALF_LOG_2PI;
        r.ent += 0.5f + HALF_LOG_2PI + ls;
    }
    return r;
}

struct LossTerms {
    float dlogp, dv, vl, al;
};

// PPO: ppo.py:61-77 and its autograd backward.  Returns the SUMMANDS (not yet divided by B).
__device__ __forceinline__ LossTerms ppo_terms(float logp, float v, float old_logp, float old_v, float ret, float adv,
                                               float clip, int clipped_vl) {
    LossTerms t;
    const float ratio = expf(logp - old_logp);
    const float lo = 1.0f - clip, hi = 1.0f + clip;
    const float rc = fminf(fmaxf(ratio, lo), hi);
    const float s1 = ratio * adv, s2 = rc * adv;
    t.al = -fminf(s1, s2);
    const float inr = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
    float g1, g2;   // d min(s1,s2) / d s1, / d s2   (ties split, torch minimum backward)
    if (s1 < s2) { g1 = 1.f; g2 = 0.f; }
    else if (s1 > s2) { g1 = 0.f; g2 = 1.f; }
    else { g1 = 0.5f; g2 = 0.5f; }
    t.dlogp = -(g1 * adv + g2 * adv * inr) * ratio;
    const float e = v - ret;
    if (clipped_vl) {
        const float dvv = v - old_v;
        const float vpc = old_v + fminf(fmaxf(dvv, -clip), clip);
        const float ec = vpc - ret;
        const float vl = e * e, vlc = ec * ec;
        const float inv = (dvv >= -clip && dvv <= clip) ? 1.f : 0.f;
        float h1, h2;   // torch maximum backward
        if (vl > vlc) { h1 = 1.f; h2 = 0.f; }
        else if (vl < vlc) { h1 = 0.f; h2 = 1.f; }
        else { h1 = 0.5f; h2 = 0.5f; }
        t.vl = 0.5f * fmaxf(vl, vlc);
        t.dv = 0.5f * (h1 * 2.f * e + h2 * 2.f * ec * inv);
    } else {
        t.vl = 0.5f * e * e;
        t.dv = e;
    }
    return t;
}

// Per-sample loss summands and their derivatives w.r.t. logp and v (before the 1/B and value_coef factors).
// ALGO is a template parameter on purpose: it is launch-uniform, and with a run-time `algo` hipcc (ROCm 7.2)
// mis-structurises the three-way branch -- the Fisher lanes never execute `t.al = -logp` (seen in the ISA).
template <int ALGO>
__device__ __forceinline__ LossTerms algo_terms(float logp, float v, float old_logp, float old_v, float ret,
                                                float advn, float clip, int clipped_vl) {
    if constexpr (ALGO == 0) return ppo_terms(logp, v, old_logp, old_v, ret, advn, clip, clipped_vl);
    LossTerms t;
    if constexpr (ALGO == 1) {            // a2c_acktr.py:48-51: adv = R - v; VL = mean(adv^2); AL = -mean(adv.detach()*logp)
        const float adv = ret - v;
        t.vl = adv * adv;
        t.al = -(adv * logp);
        t.dlogp = -adv;
        t.dv = -2.f * adv;
    } else {                    // a2c_acktr.py:56-65: pg_fisher = -mean(logp); vf_fisher = -mean((v-(v+noise).detach())^2)
        const float sample = v + advn;
        const float diff = v - sample;
        t.vl = -(diff * diff);
        t.al = -logp;
        t.dlogp = -1.0f;
        t.dv = -2.f * diff;
    }
    return t;
}

}  // namespace acktr
