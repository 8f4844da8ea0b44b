// Action distributions (log-prob / entropy, forward + backward) and the fused PPO / A2C / ACKTR-Fisher
// loss kernels.  One lane per sample; the rollout-side operands (actions, old log-probs, old values,
// returns, advantages) are read THROUGH the minibatch index, so storage.py:127-140's seven gathers
// never materialise.  Loss sums are reduced block-wise into fixed-order partials (deterministic).
//
// Reference arithmetic restated here:
//   distributions.py:18-29  + torch/distributions/categorical.py:78,150-162   (Categorical)
//   distributions.py:36-41,86-95 + torch/distributions/normal.py:88-103,115-116 (DiagGaussian)
//   algo/ppo.py:61-81, algo/a2c_acktr.py:48-51,56-65,71-72 (losses; backward = what autograd derives,
//   including torch's "ties split the gradient" rule of minimum / maximum)
#include "loss_terms.hpp"
#include "../../include/acktr_hip.h"

namespace acktr {

constexpr int LOSS_THREADS = 256;
constexpr int LOSS_SINGLE_MAX_B = 256;      // one workgroup covers the batch: it finishes the sums itself
// One kernel for all (algo, dist) pairs.  algo: 0 PPO, 1 A2C, 2 ACKTR Fisher.
// SINGLE: the batch fits one workgroup (B <= LOSS_SINGLE_MAX_B: the MLP / A2C configurations), which then finishes the
// sums itself and the fixed-order finalize launch disappears.  (One 1024-lane workgroup for B = 1024 was measured
// slower than four workgroups + finalize: 14.8 us against 9.2 + 4.6.)
template <int DIST, int ALGO, bool SINGLE>
__global__ __launch_bounds__(LOSS_THREADS) void policy_loss_kernel(acktr_loss_args a, float* partial) {
    __shared__ float sm[16];
    const int A = a.A;
    float vl = 0.f, al = 0.f, en = 0.f;
    const int NP = 3 + (DIST == 1 ? A : 0);
    float dls[DIST == 1 ? 16 : 1];
    if (DIST == 1) for (int j = 0; j < 16; ++j) dls[j] = 0.f;
    for (int b = blockIdx.x * (int)blockDim.x + threadIdx.x; b < a.B; b += a.B) {
        const long src = a.gather_idx ? (long)a.gather_idx[b] : (long)b;
        const float v = a.values[(long)b * a.ld_values];
        const float* din = a.dist_in + (long)b * a.ld_dist;
        float lse = 0.f;
        long act_i = 0;
        const float* act_f = nullptr;
        DistEval d;
        if (DIST == 0) {
            act_i = ((const int64_t*)a.actions)[src];
            d = categorical_eval(din, A, act_i, &lse);
        } else {
            act_f = (const float*)a.actions + src * A;
            d = gaussian_eval(din, a.logstd, act_f, A);
        }
        if (a.logp_out) a.logp_out[b] = d.logp;
        const float o_logp = a.old_logp ? a.old_logp[src] : 0.f;
        const float o_val = a.old_values ? a.old_values[src] : 0.f;
        const float ret = a.returns ? a.returns[src] : 0.f;
        const float advn = a.adv ? a.adv[src] : 0.f;          // advantage (PPO) or Fisher noise (ACKTR)
        const LossTerms t = algo_terms<ALGO>(d.logp, v, o_logp, o_val, ret, advn, a.clip, a.use_clipped_value_loss);
        vl += t.vl; al += t.al;
        const float dlogp = t.dlogp * a.inv_batch;
        const float dv = ((ALGO == 2) ? 1.0f : a.value_coef) * t.dv * a.inv_batch;
        en += d.ent;
        // entropy enters the loss as  -entropy_coef * mean(entropy)   (not in the Fisher loss)
        const float dent = (ALGO == 2) ? 0.f : -a.entropy_coef * a.inv_batch;
        if (a.dvalues) a.dvalues[(long)b * a.ld_dvalues] = dv;
        float* dd = a.ddist + (long)b * a.ld_ddist;
        if (DIST == 0) {
            categorical_bwd(din, A, act_i, lse, d.ent, dlogp, dent, dd);
        } else {
            for (int j = 0; j < A; ++j) {
                const float sd = expf(a.logstd[j]);
                const float var = sd * sd, dj = act_f[j] - din[j];
                dd[j] = dlogp * dj / var;
                const float g = dlogp * (dj * dj / var - 1.f) + dent;
                dls[j] += g;
                if (a.dlogstd_rows) a.dlogstd_rows[(long)b * A + j] = g;
            }
        }
    }
    float* out = partial + (long)blockIdx.x * NP;
    float r;
    if (SINGLE) {
        r = block_sum(vl, sm); if (threadIdx.x == 0 && a.loss_acc) a.loss_acc[0] += r * a.inv_batch;
        r = block_sum(al, sm); if (threadIdx.x == 0 && a.loss_acc) a.loss_acc[1] += r * a.inv_batch;
        r = block_sum(en, sm); if (threadIdx.x == 0 && a.loss_acc) a.loss_acc[2] += r * a.inv_batch;
        if (DIST == 1)
            for (int j = 0; j < A; ++j) {
                r = block_sum(dls[j], sm);
                if (threadIdx.x == 0) a.dlogstd[j] = r;
            }
        return;
    }
    r = block_sum(vl, sm); if (threadIdx.x == 0) out[0] = r;
    r = block_sum(al, sm); if (threadIdx.x == 0) out[1] = r;
    r = block_sum(en, sm); if (threadIdx.x == 0) out[2] = r;
    if (DIST == 1)
        for (int j = 0; j < A; ++j) {
            r = block_sum(dls[j], sm);
            if (threadIdx.x == 0) out[3 + j] = r;
        }
}

// fixed-order final reduction: loss_acc[0..2] += mean terms; dlogstd[j] = sum
__global__ __launch_bounds__(64) void policy_loss_finalize_kernel(const float* partial, int nblk, int NP, float inv_batch,
                                                                  float* loss_acc, float* dlogstd) {
    const int j = threadIdx.x;
    if (j >= NP) return;
    float s = 0.f;
    for (int i = 0; i < nblk; ++i) s += partial[(long)i * NP + j];
    if (j < 3) { if (loss_acc) loss_acc[j] += s * inv_batch; }
    else if (dlogstd) dlogstd[j - 3] = s;
}

// ---- stand-alone distribution kernels (autograd path of Policy.evaluate_actions) ----------
template <int DIST>
__global__ __launch_bounds__(LOSS_THREADS) void dist_fwd_kernel(const float* dist_in, long ld, const float* logstd,
                                                                const void* actions, int B, int A, float* logp,
                                                                float* entropy) {
    const int b = blockIdx.x * LOSS_THREADS + threadIdx.x;
    if (b >= B) return;
    DistEval d;
    if (DIST == 0) {
        float lse;
        d = categorical_eval(dist_in + (long)b * ld, A, ((const int64_t*)actions)[b], &lse);
    } else {
        d = gaussian_eval(dist_in + (long)b * ld, logstd, (const float*)actions + (long)b * A, A);
    }
    logp[b] = d.logp;
    entropy[b] = d.ent;
}

// entropy.mean() of model.py:77 -- one workgroup, fixed summation order (run-to-run deterministic)
__global__ __launch_bounds__(LOSS_THREADS) void mean_kernel(const float* x, int n, float* out) {
    __shared__ float sm[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += LOSS_THREADS) s += x[i];
    const float r = block_sum(s, sm);
    if (threadIdx.x == 0) out[0] = r / (float)n;
}

// Rollout-side head (Policy.act, model.py:54-66): mode or a sample of the action distribution and the log-prob of
// the chosen action, one sample per lane.
//   categorical: mode = first argmax of the logits (torch.argmax); sample (deterministic == 0) = inverse CDF of
//                softmax(logits) at the caller's uniform u[b] in [0, 1); sample (deterministic == 2) = the exponential
//                race torch.multinomial(probs, 1) runs, on the caller's Exp(1) draws q[b, i];
//   gaussian   : mode = mean; sample = mean + std * eps with the caller's standard normal eps[b, j] (torch.normal's
//                own formula, FixedNormal.sample); the log-prob follows torch Normal.log_prob term by term.
template <int DIST>
__global__ __launch_bounds__(LOSS_THREADS) void dist_act_kernel(const float* dist_in, long ld, const float* logstd,
                                                                const float* noise, int deterministic, int B, int A,
                                                                void* action_out, float* logp) {
    const int b = blockIdx.x * LOSS_THREADS + threadIdx.x;
    if (b >= B) return;
    const float* z = dist_in + (long)b * ld;
    if (DIST == 0) {
        float m = z[0];
        int am = 0;
        for (int i = 1; i < A; ++i) if (z[i] > m) { m = z[i]; am = i; }
        float ssum = 0.f;
        for (int i = 0; i < A; ++i) ssum += expf(z[i] - m);
        const float lse = m + logf(ssum);
        int a = am;
        if (deterministic == 2) {
            // the reference's own draw on its CPU path: Categorical.sample -> torch.multinomial(probs, 1), whose
            // one-sample fast path is argmax_i p_i / q_i with q ~ Exp(1) drawn for every (row, category); the caller
            // hands in q from the same generator call (torch.empty(B, A).exponential_(1)), the common normaliser of
            // p cancels in the argmax, ties resolve to the first index like torch.argmax
            float best = -1.f;
            a = 0;
            for (int i = 0; i < A; ++i) {
                const float r = expf(z[i] - m) / noise[(long)b * A + i];
                if (r > best) { best = r; a = i; }
            }
        } else if (!deterministic) {
            const float target = noise[b] * ssum;          // u * sum_i exp(z_i - m): no division per category
            float acc = 0.f;
            a = A - 1;
            for (int i = 0; i < A; ++i) {
                acc += expf(z[i] - m);
                if (target < acc) { a = i; break; }
            }
        }
        ((int64_t*)action_out)[b] = a;
        logp[b] = z[a] - lse;
    } else {
        float lp = 0.f;
        float* out = (float*)action_out + (long)b * A;
        for (int j = 0; j < A; ++j) {
            const float sd = expf(logstd[j]);
            const float act = deterministic == 1 ? z[j] : z[j] + sd * noise[(long)b * A + j];
            const float var = sd * sd, d = act - z[j];
            lp += -(d * d) / (2.f * var) - logf(sd) - HALF_LOG_2PI;
            out[j] = act;
        }
        logp[b] = lp;
    }
}

template <int DIST>
__global__ __launch_bounds__(LOSS_THREADS) void dist_bwd_kernel(const float* dist_in, long ld, const float* logstd,
                                                                const void* actions, const float* dlogp,
                                                                const float* dent, int B, int A, float* ddist, long ldd,
                                                                float* partial, float* dls_rows) {
    __shared__ float sm[16];
    const int b = blockIdx.x * LOSS_THREADS + threadIdx.x;
    float dls[DIST == 1 ? 16 : 1];
    if (DIST == 1) for (int j = 0; j < 16; ++j) dls[j] = 0.f;
    if (b < B) {
        const float* din = dist_in + (long)b * ld;
        float* dd = ddist + (long)b * ldd;
        const float gl = dlogp ? dlogp[b] : 0.f, ge = dent ? dent[b] : 0.f;
        if (DIST == 0) {
            float lse;
            const long act = ((const int64_t*)actions)[b];
            const DistEval d = categorical_eval(din, A, act, &lse);
            categorical_bwd(din, A, act, lse, d.ent, gl, ge, dd);
        } else {
            const float* act = (const float*)actions + (long)b * A;
            for (int j = 0; j < A; ++j) {
                const float sd = expf(logstd[j]);
                const float var = sd * sd, dj = act[j] - din[j];
                dd[j] = gl * dj / var;
                dls[j] = gl * (dj * dj / var - 1.f) + ge;
                if (dls_rows) dls_rows[(long)b * A + j] = dls[j];
            }
        }
    }
    if (DIST == 1)
        for (int j = 0; j < A; ++j) {
            const float r = block_sum(dls[j], sm);
            if (threadIdx.x == 0) partial[(long)blockIdx.x * (3 + A) + 3 + j] = r;
        }
}

}  // namespace acktr

using namespace acktr;

extern "C" long acktr_policy_loss_workspace_bytes(int B, int A) {
    return (long)cdiv(B, LOSS_THREADS) * (3 + A) * (long)sizeof(float) + 256;
}

extern "C" int acktr_policy_loss_f32(const acktr_loss_args* args, void* stream) {
    ACKTR_ARG(args, "policy_loss: null args");
    const acktr_loss_args& a = *args;
    ACKTR_ARG(a.algo >= 0 && a.algo <= 2 && (a.dist == 0 || a.dist == 1), "policy_loss: bad algo/dist");
    ACKTR_ARG(a.B > 0 && a.A > 0, "policy_loss: bad sizes B=%d A=%d", a.B, a.A);
    ACKTR_ARG(a.dist == 0 || a.A <= 16, "policy_loss: gaussian action dim %d > 16 unsupported", a.A);
    ACKTR_ARG(a.values && a.dist_in && a.actions && a.ddist, "policy_loss: null tensor");
    ACKTR_ARG(a.dist == 0 || (a.logstd && a.dlogstd), "policy_loss: gaussian needs logstd/dlogstd");
    ACKTR_ARG(a.algo != 0 || (a.old_logp && a.old_values && a.returns && a.adv), "policy_loss: PPO needs rollout arrays");
    ACKTR_ARG(a.algo != 1 || a.returns, "policy_loss: A2C needs returns");
    ACKTR_ARG(a.algo != 2 || a.adv, "policy_loss: Fisher loss needs the noise tensor in `adv`");
    ACKTR_ARG(a.workspace && a.workspace_bytes >= acktr_policy_loss_workspace_bytes(a.B, a.A),
              "policy_loss: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = cdiv(a.B, LOSS_THREADS);
    float* partial = (float*)a.workspace;
    const bool single = a.B <= LOSS_SINGLE_MAX_B;
#define LAUNCH_LOSS(D, G)                                                                                           \
    do {                                                                                                            \
        if (single) hipLaunchKernelGGL((policy_loss_kernel<D, G, true>), dim3(1), dim3(LOSS_THREADS), 0, s, a, partial); \
        else hipLaunchKernelGGL((policy_loss_kernel<D, G, false>), dim3(nblk), dim3(LOSS_THREADS), 0, s, a, partial); \
    } while (0)
    if (a.dist == 0) {
        if (a.algo == 0) LAUNCH_LOSS(0, 0); else if (a.algo == 1) LAUNCH_LOSS(0, 1); else LAUNCH_LOSS(0, 2);
    } else {
        if (a.algo == 0) LAUNCH_LOSS(1, 0); else if (a.algo == 1) LAUNCH_LOSS(1, 1); else LAUNCH_LOSS(1, 2);
    }
#undef LAUNCH_LOSS
    ACKTR_LAUNCH_CHECK();
    if (single) return 0;
    const int NP = 3 + (a.dist == 1 ? a.A : 0);
    hipLaunchKernelGGL(policy_loss_finalize_kernel, dim3(1), dim3(64), 0, s, partial, nblk, NP, a.inv_batch, a.loss_acc,
                       a.dist == 1 ? a.dlogstd : (float*)nullptr);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_dist_fwd_f32(int dist, const float* dist_in, long ld, const float* logstd, const void* actions,
                                  int B, int A, float* logp, float* entropy, float* entropy_mean, void* stream) {
    ACKTR_ARG(dist_in && actions && logp && entropy, "dist_fwd: null pointer");
    ACKTR_ARG(B > 0 && A > 0 && ld >= A, "dist_fwd: bad sizes");
    ACKTR_ARG(dist == 0 || (dist == 1 && logstd && A <= 16), "dist_fwd: bad distribution arguments");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = cdiv(B, LOSS_THREADS);
    if (dist == 0)
        hipLaunchKernelGGL((dist_fwd_kernel<0>), dim3(nblk), dim3(LOSS_THREADS), 0, s, dist_in, ld, logstd, actions, B, A,
                           logp, entropy);
    else
        hipLaunchKernelGGL((dist_fwd_kernel<1>), dim3(nblk), dim3(LOSS_THREADS), 0, s, dist_in, ld, logstd, actions, B, A,
                           logp, entropy);
    ACKTR_LAUNCH_CHECK();
    if (entropy_mean) {
        hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(LOSS_THREADS), 0, s, entropy, B, entropy_mean);
        ACKTR_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int acktr_dist_act_f32(int dist, const float* dist_in, long ld, const float* logstd, const float* noise,
                                  int deterministic, int B, int A, void* action_out, float* logp, void* stream) {
    ACKTR_ARG(dist_in && action_out && logp, "dist_act: null pointer");
    ACKTR_ARG(B > 0 && A > 0 && ld >= A, "dist_act: bad sizes");
    ACKTR_ARG(dist == 0 || (dist == 1 && logstd && A <= 16), "dist_act: bad distribution arguments");
    ACKTR_ARG(deterministic >= 0 && deterministic <= 2, "dist_act: deterministic must be 0, 1 or 2");
    ACKTR_ARG(deterministic == 1 || noise, "dist_act: sampling needs the noise array (uniform [B] / Exp(1) [B,A] / standard normal [B,A])");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = cdiv(B, LOSS_THREADS);
    if (dist == 0)
        hipLaunchKernelGGL((dist_act_kernel<0>), dim3(nblk), dim3(LOSS_THREADS), 0, s, dist_in, ld, logstd, noise,
                           deterministic, B, A, action_out, logp);
    else
        hipLaunchKernelGGL((dist_act_kernel<1>), dim3(nblk), dim3(LOSS_THREADS), 0, s, dist_in, ld, logstd, noise,
                           deterministic, B, A, action_out, logp);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_dist_bwd_f32(int dist, const float* dist_in, long ld, const float* logstd, const void* actions,
                                  const float* dlogp, const float* dentropy, int B, int A, float* ddist, long ldd,
                                  float* dlogstd, float* dlogstd_rows, void* workspace, long workspace_bytes, void* stream) {
    ACKTR_ARG(dist_in && actions && ddist, "dist_bwd: null pointer");
    ACKTR_ARG(B > 0 && A > 0 && ld >= A && ldd >= A, "dist_bwd: bad sizes");
    ACKTR_ARG(dist == 0 || (dist == 1 && logstd && dlogstd && A <= 16), "dist_bwd: bad distribution arguments");
    ACKTR_ARG(dist == 0 || (workspace && workspace_bytes >= acktr_policy_loss_workspace_bytes(B, A)),
              "dist_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = cdiv(B, LOSS_THREADS);
    if (dist == 0) {
        hipLaunchKernelGGL((dist_bwd_kernel<0>), dim3(nblk), dim3(LOSS_THREADS), 0, s, dist_in, ld, logstd, actions, dlogp,
                           dentropy, B, A, ddist, ldd, (float*)workspace, (float*)nullptr);
        ACKTR_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL((dist_bwd_kernel<1>), dim3(nblk), dim3(LOSS_THREADS), 0, s, dist_in, ld, logstd, actions, dlogp,
                           dentropy, B, A, ddist, ldd, (float*)workspace, dlogstd_rows);
        ACKTR_LAUNCH_CHECK();
        hipLaunchKernelGGL(policy_loss_finalize_kernel, dim3(1), dim3(64), 0, s, (const float*)workspace, nblk, 3 + A, 0.f,
                           (float*)nullptr, dlogstd);
        ACKTR_LAUNCH_CHECK();
    }
    return 0;
}
