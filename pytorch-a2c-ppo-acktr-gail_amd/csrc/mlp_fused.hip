// One-launch minibatch step of the MLP policy (reference a2c_ppo_acktr/model.py:198-229 MLPBase + distributions.py:76-95
// DiagGaussian + the loss blocks of algo/ppo.py:57-81 / algo/a2c_acktr.py:38-51,70-72): forward of both tanh towers and
// the two heads, the PPO / A2C loss with its backward seeds, and every parameter gradient, for a minibatch of up to 64
// rows -- the reference's MuJoCo configuration (README.md:115: 2048 steps, 32 minibatches of 64).
//
// Why one workgroup: at 64 rows the whole step is ~4 MFLOP; run layer by layer it is ~30 dependent launches of
// ~4.5 us each and nothing else.  Here the parameters (10 k floats), the gathered observations and all activations
// live in the LDS of one CU (<= 150 KB) and the step is a chain of 64 x 64 x K products on the matrix cores:
//   512 threads = one group of 256 per tower; the four waves of a group own the four 32 x 32 tiles of a product and feed
//   v_mfma_f32_32x32x2_f32 (exact fp32) straight from LDS; matrices whose rows are indexed by lanes have odd pitches.
// The backward overwrites activations in place (dh2 over h2, dh1 over h1), so no further LDS is needed.
// Gradients are written straight into the caller's flat gradient buffer (overwrite, like the layer kernels).
#include "loss_terms.hpp"
#include "optim_terms.hpp"
#include "../../include/acktr_hip.h"

namespace acktr {

constexpr int MF_ROWS = 64;       // rows per step (padded with zero rows)
constexpr int MF_H = 64;          // hidden width (MLPBase default, model.py:199)
constexpr int MF_MAX_IN = 64;
constexpr int MF_MAX_A = 8;
constexpr int MF_PH = 65;         // pitch of [64 x 64] matrices in LDS (odd: rows indexed by lanes are conflict-free)
constexpr int MF_THREADS = 512;   // two groups of 256: one per tower (critic / actor) wherever the towers are independent

typedef float mf_f16 __attribute__((ext_vector_type(16)));

// One 32 x 32 tile per wave on the matrix cores: acc(m, n) = sum_{k < K} a(m0 + m, k) * b(n0 + n, k), K even.
// v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain; lane (li = l % 32, kh = l / 32) supplies a(m0 + li, k + kh) and
// b(n0 + li, k + kh) straight from LDS.  Result layout: register v of lane l is (row 4 kh + (v & 3) + 8 (v >> 2), column li).
// (The first version ran these products on the vector ALUs, 4 x 4 outputs per lane: 3.4 us per 64 x 64 x 64 stage,
//  five stages; one dependent MFMA chain of 32 steps is 0.9 us.)
template <class FA, class FB>
__device__ __forceinline__ void mf_mm(mf_f16& acc, const int K, int m0, int n0, int li, int kh, FA a, FB b) {
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll 8
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a(m0 + li, k + kh), b(n0 + li, k + kh), acc, 0, 0, 0);
}
// row / column of accumulator register v of this lane inside the tile
__device__ __forceinline__ int mf_row(int v, int kh) { return 4 * kh + (v & 3) + 8 * (v >> 2); }

struct MfLds {      // float offsets into the dynamic LDS block
    int w0c, w0a, w2c, w2a, wv, wm, b0c, b0a, b2c, b2a, bm, x, h1c, h1a, h2c, h2a, dv, dmean, red, total;
    int pin;        // pitch of the [64 x IN] matrices (odd)
};
__host__ __device__ inline MfLds mf_layout(int IN, int A) {
    MfLds L;
    L.pin = IN | 1;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) / 4 * 4; return r; };
    L.w0c = take(MF_H * L.pin); L.w0a = take(MF_H * L.pin);
    L.w2c = take(MF_H * MF_PH); L.w2a = take(MF_H * MF_PH);
    L.wv = take(MF_H); L.wm = take(MF_MAX_A * MF_H);
    L.b0c = take(MF_H); L.b0a = take(MF_H); L.b2c = take(MF_H); L.b2a = take(MF_H); L.bm = take(MF_MAX_A);
    L.x = take(MF_ROWS * L.pin);
    L.h1c = take(MF_ROWS * MF_PH); L.h1a = take(MF_ROWS * MF_PH);
    L.h2c = take(MF_ROWS * MF_PH); L.h2a = take(MF_ROWS * MF_PH);
    L.dv = take(MF_ROWS); L.dmean = take(MF_ROWS * MF_MAX_A);
    L.red = take(64);
    L.total = o;
    return L;
}

template <int ALGO, bool CAT>
__global__ __launch_bounds__(MF_THREADS) void mlp_fused_step_kernel(acktr_mlp_step_args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, tw = tid >> 8, t256 = tid & 255;
    // tower tw: four waves = the four 32 x 32 tiles of a 64 x 64 product
    const int li = tid & 31, kh = (tid >> 5) & 1, m0 = 32 * ((tid >> 7) & 1), n0 = 32 * ((tid >> 6) & 1);
    const int IN = a.IN, A = a.A, B = a.B;
    const MfLds L = mf_layout(IN, A);
    const int PIN = L.pin;

    // ---- stage 0: parameters and the gathered observation rows into LDS
    for (int i = tid; i < MF_H * IN; i += MF_THREADS) {
        const int u = i / IN, k = i - u * IN;
        sm[L.w0c + u * PIN + k] = a.wc0[i];
        sm[L.w0a + u * PIN + k] = a.wa0[i];
    }
    for (int i = tid; i < MF_H * MF_H; i += MF_THREADS) {
        const int u = i >> 6, k = i & 63;
        sm[L.w2c + u * MF_PH + k] = a.wc2[i];
        sm[L.w2a + u * MF_PH + k] = a.wa2[i];
    }
    for (int i = tid; i < A * MF_H; i += MF_THREADS) sm[L.wm + i] = a.wm[i];
    if (tid < MF_H) {
        sm[L.wv + tid] = a.wv[tid];
        sm[L.b0c + tid] = a.bc0[tid]; sm[L.b0a + tid] = a.ba0[tid];
        sm[L.b2c + tid] = a.bc2[tid]; sm[L.b2a + tid] = a.ba2[tid];
        if (tid < A) sm[L.bm + tid] = a.bm[tid];
    }
    for (int i = tid; i < MF_ROWS * IN; i += MF_THREADS) {
        const int r = i / IN, k = i - r * IN;
        float v = 0.f;
        if (r < B) {
            const long src = a.gather_idx ? (long)a.gather_idx[r] : (long)r;
            v = a.obs[src * IN + k];
        }
        sm[L.x + r * PIN + k] = v;
    }
    __syncthreads();

    mf_f16 acc;
    // ---- stage 1: h1 = tanh(x W0^T + b0), both towers        (model.py:208-216,226-227)
    {
        const float* W = sm + (tw ? L.w0a : L.w0c);
        const float* bias = sm + (tw ? L.b0a : L.b0c);
        float* H = sm + (tw ? L.h1a : L.h1c);
        const float* X = sm + L.x;
        mf_mm(acc, (IN + 1) & ~1, m0, n0, li, kh, [&](int r, int k) { return k < IN ? X[r * PIN + k] : 0.f; },
              [&](int u, int k) { return k < IN ? W[u * PIN + k] : 0.f; });
#pragma unroll
        for (int v = 0; v < 16; ++v) H[(m0 + mf_row(v, kh)) * MF_PH + n0 + li] = tanhf(acc[v] + bias[n0 + li]);
    }
    __syncthreads();
    // ---- stage 2: h2 = tanh(h1 W2^T + b2)
    {
        const float* W = sm + (tw ? L.w2a : L.w2c);
        const float* bias = sm + (tw ? L.b2a : L.b2c);
        const float* H1 = sm + (tw ? L.h1a : L.h1c);
        float* H2 = sm + (tw ? L.h2a : L.h2c);
        mf_mm(acc, MF_H, m0, n0, li, kh, [&](int r, int k) { return H1[r * MF_PH + k]; }, [&](int u, int k) { return W[u * MF_PH + k]; });
#pragma unroll
        for (int v = 0; v < 16; ++v) H2[(m0 + mf_row(v, kh)) * MF_PH + n0 + li] = tanhf(acc[v] + bias[n0 + li]);
    }
    __syncthreads();
    // ---- stage 3: heads, distribution, loss and its seeds -- one lane per row (wave 0)
    if (tid < MF_ROWS) {
        const int r = tid;
        const float* h2c = sm + L.h2c + r * MF_PH;
        const float* h2a = sm + L.h2a + r * MF_PH;
        float v = 0.f;
        for (int u = 0; u < MF_H; ++u) v = __builtin_fmaf(h2c[u], sm[L.wv + u], v);
        v += a.bv[0];
        float mean[MF_MAX_A];
#pragma unroll
        for (int j = 0; j < MF_MAX_A; ++j) {
            float m = 0.f;
            if (j < A) {
                for (int u = 0; u < MF_H; ++u) m = __builtin_fmaf(h2a[u], sm[L.wm + j * MF_H + u], m);
                m += sm[L.bm + j];
            }
            mean[j] = m;
        }
        float vl = 0.f, al = 0.f, en = 0.f, dv = 0.f;
        float dls[MF_MAX_A], dm[MF_MAX_A];
#pragma unroll
        for (int j = 0; j < MF_MAX_A; ++j) { dls[j] = 0.f; dm[j] = 0.f; }
        if (r < B) {
            const long src = a.gather_idx ? (long)a.gather_idx[r] : (long)r;
            const float o_logp = a.old_logp ? a.old_logp[src] : 0.f;
            const float o_val = a.old_values ? a.old_values[src] : 0.f;
            const float ret = a.returns ? a.returns[src] : 0.f;
            const float advn = a.adv ? a.adv[src] : 0.f;
            if constexpr (CAT) {
                // distributions.py:59-73 Categorical on the logits `mean[]` (the arithmetic of head_fused.hip's phase B)
                const long act64 = a.actions_i64[src];
                const int act_l = (act64 >= 0 && act64 < A) ? (int)act64 : -1;
                float mx = mean[0];
#pragma unroll
                for (int j = 1; j < MF_MAX_A; ++j) if (j < A) mx = fmaxf(mx, mean[j]);
                float se = 0.f;
#pragma unroll
                for (int j = 0; j < MF_MAX_A; ++j) if (j < A) se += expf(mean[j] - mx);
                const float lse = mx + logf(se);
                float ent = 0.f, za = 0.f;
#pragma unroll
                for (int j = 0; j < MF_MAX_A; ++j) if (j < A) {
                    const float lp = fmaxf(mean[j] - lse, -3.402823466e+38f);
                    ent -= lp * expf(lp);
                    if (j == act_l) za = mean[j];
                }
                const float logp = act_l >= 0 ? za - lse : 0.f;
                const LossTerms t = algo_terms<ALGO>(logp, v, o_logp, o_val, ret, advn, a.clip, a.use_clipped_value_loss);
                vl = t.vl; al = t.al; en = ent;
                const float dlogp = t.dlogp * a.inv_batch;
                dv = a.value_coef * t.dv * a.inv_batch;
                const float dent = -a.entropy_coef * a.inv_batch;
#pragma unroll
                for (int j = 0; j < MF_MAX_A; ++j)
                    if (j < A) {
                        const float lp = mean[j] - lse, pr = expf(lp);
                        dm[j] = dlogp * ((j == act_l ? 1.f : 0.f) - pr) - dent * pr * (lp + ent);
                    }
            } else {
                const float* act = a.actions + src * A;
                float actv[MF_MAX_A];
#pragma unroll
                for (int j = 0; j < MF_MAX_A; ++j) actv[j] = j < A ? act[j] : 0.f;
                const DistEval d = gaussian_eval(mean, a.logstd, actv, A);
                const LossTerms t = algo_terms<ALGO>(d.logp, v, o_logp, o_val, ret, advn, a.clip, a.use_clipped_value_loss);
                vl = t.vl; al = t.al; en = d.ent;
                const float dlogp = t.dlogp * a.inv_batch;
                dv = a.value_coef * t.dv * a.inv_batch;
                const float dent = -a.entropy_coef * a.inv_batch;
#pragma unroll
                for (int j = 0; j < MF_MAX_A; ++j) {
                    if (j < A) {
                        const float sd = expf(a.logstd[j]);
                        const float var = sd * sd, dj = actv[j] - mean[j];
                        dm[j] = dlogp * dj / var;
                        dls[j] = dlogp * (dj * dj / var - 1.f) + dent;
                    }
                }
            }
        }
        sm[L.dv + r] = dv;
#pragma unroll
        for (int j = 0; j < MF_MAX_A; ++j) sm[L.dmean + r * MF_MAX_A + j] = dm[j];
        // wave 0 holds all 64 rows: fixed-order wave sums
        const float svl = wave_sum(vl), sal = wave_sum(al), sen = wave_sum(en), sdv = wave_sum(dv);
        if (tid == 0) {
            if (a.loss_acc) {
                a.loss_acc[0] += svl * a.inv_batch;
                a.loss_acc[1] += sal * a.inv_batch;
                a.loss_acc[2] += sen * a.inv_batch;
            }
            a.gbv[0] = sdv;
        }
#pragma unroll
        for (int j = 0; j < MF_MAX_A; ++j) {
            if (j < A) {                                         // uniform
                const float s1 = wave_sum(dls[j]), s2 = wave_sum(dm[j]);
                if (tid == 0) { if (!CAT) a.glogstd[j] = s1; a.gbm[j] = s2; }
            }
        }
    }
    __syncthreads();
    // ---- stage 4a: head weight gradients (h2 still holds activations)
    if (tid < MF_H) {
        const int u = tid;
        float g = 0.f;
        for (int r = 0; r < MF_ROWS; ++r) g = __builtin_fmaf(sm[L.dv + r], sm[L.h2c + r * MF_PH + u], g);
        a.gwv[u] = g;
    }
    for (int i = tid; i < A * MF_H; i += MF_THREADS) {
        const int j = i >> 6, u = i & 63;
        float g = 0.f;
        for (int r = 0; r < MF_ROWS; ++r) g = __builtin_fmaf(sm[L.dmean + r * MF_MAX_A + j], sm[L.h2a + r * MF_PH + u], g);
        a.gwm[i] = g;
    }
    __syncthreads();
    // ---- stage 4b: dh2 = (d head-out W_head) * (1 - h2^2), in place over h2
    for (int i = tid; i < MF_ROWS * MF_H; i += MF_THREADS) {
        const int r = i >> 6, u = i & 63;
        const float hc = sm[L.h2c + r * MF_PH + u];
        sm[L.h2c + r * MF_PH + u] = (sm[L.dv + r] * sm[L.wv + u]) * (1.f - hc * hc);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < MF_MAX_A; ++j)
            if (j < A) s = __builtin_fmaf(sm[L.dmean + r * MF_MAX_A + j], sm[L.wm + j * MF_H + u], s);
        const float ha = sm[L.h2a + r * MF_PH + u];
        sm[L.h2a + r * MF_PH + u] = s * (1.f - ha * ha);
    }
    __syncthreads();
    // ---- stage 5: dW2[u][i] = sum_r dh2[r][u] h1[r][i]; db2[u] = sum_r dh2[r][u]
    {
        const float* D2 = sm + (tw ? L.h2a : L.h2c);
        const float* H1 = sm + (tw ? L.h1a : L.h1c);
        float* gw = tw ? a.gwa2 : a.gwc2;
        float* gb = tw ? a.gba2 : a.gbc2;
        mf_mm(acc, MF_ROWS, m0, n0, li, kh, [&](int u, int r) { return D2[r * MF_PH + u]; }, [&](int i, int r) { return H1[r * MF_PH + i]; });
#pragma unroll
        for (int v = 0; v < 16; ++v) gw[(m0 + mf_row(v, kh)) * MF_H + n0 + li] = acc[v];
        if (t256 < MF_H) {
            float g = 0.f;
            for (int r = 0; r < MF_ROWS; ++r) g += D2[r * MF_PH + t256];
            gb[t256] = g;
        }
    }
    __syncthreads();
    // ---- stage 6: dh1[r][i] = (sum_u dh2[r][u] W2[u][i]) * (1 - h1[r][i]^2), in place over h1
    {
        const float* D2 = sm + (tw ? L.h2a : L.h2c);
        const float* W = sm + (tw ? L.w2a : L.w2c);
        float* H1 = sm + (tw ? L.h1a : L.h1c);
        mf_mm(acc, MF_H, m0, n0, li, kh, [&](int r, int u) { return D2[r * MF_PH + u]; }, [&](int i, int u) { return W[u * MF_PH + i]; });
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            float* p = H1 + (m0 + mf_row(v, kh)) * MF_PH + n0 + li;
            const float h = *p;
            *p = acc[v] * (1.f - h * h);
        }
    }
    __syncthreads();
    // ---- stage 7: dW0[u][k] = sum_r dh1[r][u] x[r][k]; db0[u] = sum_r dh1[r][u]
    {
        const float* D1 = sm + (tw ? L.h1a : L.h1c);
        const float* X = sm + L.x;
        float* gw = tw ? a.gwa0 : a.gwc0;
        float* gb = tw ? a.gba0 : a.gbc0;
        mf_mm(acc, MF_ROWS, m0, n0, li, kh, [&](int u, int r) { return D1[r * MF_PH + u]; },
              [&](int k, int r) { return k < IN ? X[r * PIN + k] : 0.f; });
#pragma unroll
        for (int v = 0; v < 16; ++v)
            if (n0 + li < IN) gw[(m0 + mf_row(v, kh)) * IN + n0 + li] = acc[v];
        if (t256 < MF_H) {
            float g = 0.f;
            for (int r = 0; r < MF_ROWS; ++r) g += D1[r * MF_PH + t256];
            gb[t256] = g;
        }
    }
    // ---- stage 8 (optional): clip_grad_norm_ + Adam over the flat buffers, same arithmetic as clip_update_kernel<0>
    // (optim.hip).  The step is ONE workgroup, so the global norm needs no second launch: two ~5 us launches of a 57 us
    // step.  The gradient stores above are this workgroup's own: visible to all its lanes behind the barrier.
    if (a.opt_params) {
        __syncthreads();
        double* smd = reinterpret_cast<double*>(sm + L.red);     // 16 doubles of the 64-float reduction scratch (the kernel's
        float* smo = sm + L.red + 40;                            // LDS is all dynamic: 160 KB leave no room for static arrays)
        // every operand of the thread's (up to) six float4 groups is requested before the norm is reduced: one memory round
        // trip for the whole tail instead of one per loop iteration (the scalar loops took 8 us)
        typedef float v4 __attribute__((ext_vector_type(4)));
        constexpr int MAXG = 6;                                  // 6 * 512 * 4 = 12288 floats: MLPBase(64) with IN <= 64, A <= 8
        const long n4 = a.opt_n / 4;                             // (flat buffers are padded to 16-byte slots: opt_n % 4 == 0)
        v4 gv[MAXG], pv[MAXG], av[MAXG], bv[MAXG];
#pragma unroll
        for (int u = 0; u < MAXG; ++u) {
            const long i = tid + (long)u * MF_THREADS;
            const long ic = i < n4 ? i : 0;
            gv[u] = reinterpret_cast<const v4*>(a.opt_grads)[ic];
            pv[u] = reinterpret_cast<const v4*>(a.opt_params)[ic];
            av[u] = reinterpret_cast<const v4*>(a.opt_exp_avg)[ic];
            bv[u] = reinterpret_cast<const v4*>(a.opt_exp_avg_sq)[ic];
        }
        double ss = 0.0;
#pragma unroll
        for (int u = 0; u < MAXG; ++u)
            if (tid + (long)u * MF_THREADS < n4)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const double x = (double)gv[u][j]; ss += x * x; }
        ss = block_sum(ss, smd);
        if (tid == 0) {
            const int step = *a.opt_step_dev + 1;
            *a.opt_step_dev = step;
            const float lr = a.opt_hyper_dev[0], beta1 = a.opt_hyper_dev[1], beta2 = a.opt_hyper_dev[2];
            float coef = 1.0f;
            if (a.opt_max_norm > 0.f) {
                const float c = a.opt_max_norm / ((float)sqrt(ss) + 1e-6f);
                coef = c > 1.0f ? 1.0f : c;
            }
            smo[0] = coef;
            smo[1] = (float)((double)lr / (1.0 - pow((double)beta1, (double)step)));
            smo[2] = (float)sqrt(1.0 - pow((double)beta2, (double)step));
        }
        __syncthreads();
        const float coef = smo[0], step_size = smo[1], bc2_sqrt = smo[2];
        const float beta1 = a.opt_hyper_dev[1], beta2 = a.opt_hyper_dev[2], eps = a.opt_hyper_dev[3];
        const bool clip = a.opt_max_norm > 0.f;
#pragma unroll
        for (int u = 0; u < MAXG; ++u) {
            const long i = tid + (long)u * MF_THREADS;
            if (i < n4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float pi = pv[u][j], gi = gv[u][j], a1 = av[u][j], a2 = bv[u][j];
                    if (clip) gi = gi * coef;
                    adam_elem(pi, gi, a1, a2, beta1, beta2, eps, bc2_sqrt, step_size);
                    pv[u][j] = pi; gv[u][j] = gi; av[u][j] = a1; bv[u][j] = a2;
                }
                reinterpret_cast<v4*>(a.opt_params)[i] = pv[u];
                if (clip) reinterpret_cast<v4*>(a.opt_grads)[i] = gv[u];
                reinterpret_cast<v4*>(a.opt_exp_avg)[i] = av[u];
                reinterpret_cast<v4*>(a.opt_exp_avg_sq)[i] = bv[u];
            }
        }
    }
}

}  // namespace acktr

using namespace acktr;

extern "C" int acktr_mlp_fused_step_supported(int B, int IN, int H, int A) {
    return B >= 1 && B <= MF_ROWS && H == MF_H && IN >= 1 && IN <= MF_MAX_IN && A >= 1 && A <= MF_MAX_A;
}

extern "C" int acktr_mlp_fused_step_f32(const acktr_mlp_step_args* args, void* stream) {
    ACKTR_ARG(args, "mlp_fused_step: null args");
    const acktr_mlp_step_args& a = *args;
    ACKTR_ARG(acktr_mlp_fused_step_supported(a.B, a.IN, a.H, a.A),
              "mlp_fused_step: unsupported shape B=%d IN=%d H=%d A=%d (B <= %d, H == %d, IN <= %d, A <= %d)", a.B, a.IN,
              a.H, a.A, MF_ROWS, MF_H, MF_MAX_IN, MF_MAX_A);
    ACKTR_ARG(a.algo == 0 || a.algo == 1, "mlp_fused_step: algo must be 0 (PPO) or 1 (A2C)");
    const bool cat = a.categorical != 0;
    ACKTR_ARG(a.wc0 && a.bc0 && a.wc2 && a.bc2 && a.wa0 && a.ba0 && a.wa2 && a.ba2 && a.wv && a.bv && a.wm && a.bm &&
                  (cat || a.logstd), "mlp_fused_step: null parameter pointer");
    ACKTR_ARG(a.gwc0 && a.gbc0 && a.gwc2 && a.gbc2 && a.gwa0 && a.gba0 && a.gwa2 && a.gba2 && a.gwv && a.gbv && a.gwm &&
                  a.gbm && (cat || a.glogstd), "mlp_fused_step: null gradient pointer");
    ACKTR_ARG(a.obs && (cat ? a.actions_i64 != nullptr : a.actions != nullptr), "mlp_fused_step: null observations / actions");
    ACKTR_ARG(a.algo != 0 || (a.old_logp && a.old_values && a.returns && a.adv), "mlp_fused_step: PPO needs rollout arrays");
    ACKTR_ARG(a.algo != 1 || a.returns, "mlp_fused_step: A2C needs returns");
    ACKTR_ARG(!a.opt_params || (a.opt_grads && a.opt_exp_avg && a.opt_exp_avg_sq && a.opt_hyper_dev && a.opt_step_dev && a.opt_n > 0),
              "mlp_fused_step: the optimiser tail needs all of its buffers");
    ACKTR_ARG(!a.opt_params || (a.opt_n % 4 == 0 && a.opt_n <= 6L * MF_THREADS * 4 &&
                                ((((uintptr_t)a.opt_params | (uintptr_t)a.opt_grads | (uintptr_t)a.opt_exp_avg | (uintptr_t)a.opt_exp_avg_sq) & 15) == 0)),
              "mlp_fused_step: the optimiser tail needs 16-byte aligned flat buffers of n %% 4 == 0, n <= %d floats", 6 * MF_THREADS * 4);
    const MfLds L = mf_layout(a.IN, a.A);
    const int smem = L.total * (int)sizeof(float);
    ACKTR_ARG(smem <= 160 * 1024, "mlp_fused_step: LDS budget exceeded (%d bytes)", smem);
    static bool attr_set[4] = {false, false, false, false};
    auto launch = [&](auto kern, int which) -> int {
        if (!attr_set[which]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               160 * 1024);
            if (e != hipSuccess) { set_error("mlp_fused_step: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
            attr_set[which] = true;
        }
        hipLaunchKernelGGL(kern, dim3(1), dim3(MF_THREADS), smem, (hipStream_t)stream, a);
        return 0;
    };
    int rc;
    if (a.algo == 0) rc = cat ? launch(mlp_fused_step_kernel<0, true>, 2) : launch(mlp_fused_step_kernel<0, false>, 0);
    else rc = cat ? launch(mlp_fused_step_kernel<1, true>, 3) : launch(mlp_fused_step_kernel<1, false>, 1);
    if (rc) return rc;
    ACKTR_LAUNCH_CHECK();
    return 0;
}
