// GAIL discriminator pieces that are not GEMMs (reference a2c_ppo_acktr/algo/gail.py):
//   * minibatch assembly: [policy (s,a); expert (s,a); alpha-mixup] rows in one matrix          gail.py:36-41,67-76
//   * BCE-with-logits losses and their backward seeds                                            gail.py:78-85
//   * gradient penalty  lambda * mean((||d D/d x|| - 1)^2)  and its seed d pen / d g             gail.py:46-54
//   * the element-wise links of the DOUBLE backward through the two tanh layers                  (autograd of gail.py:46-54,93)
//   * predict_reward: log(sigmoid d) - log(1 - sigmoid d), the running discounted return,
//     per-step batch moments for the RunningMeanStd merge, and the final scaling                 gail.py:97-110
// The linear layers themselves (forward, input gradients, weight gradients) run on the MFMA GEMM / skinny kernels
// of linear.hip; the host mirror (a2c_ppo_acktr/algo/gail.py) chains them.
#include "common.hpp"
#include "gemm_core.hpp"
#include "../../include/acktr_hip.h"

#pragma clang fp contract(off)

namespace acktr {

// X[0:B] = [ps | pa], X[B:2B] = [es | ea], X[2B:3B] = alpha * expert + (1 - alpha) * policy (copies == 2: again at X[3B:4B])
// (idx, optional: policy row b is row idx[b] of ps / pa -- the minibatch gather of storage.py:127-140 fused in)
__global__ __launch_bounds__(256) void gail_mix_concat_kernel(const float* ps, const float* pa, const int64_t* idx,
                                                              const float* es, const float* ea, const float* alpha, int B,
                                                              int Ds, int Da, int copies, float* X) {
    const int D = Ds + Da;
    const long n = (long)B * D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int b = (int)(i / D), c = (int)(i % D);
        const long pb = idx ? idx[b] : (long)b;
        const float p = c < Ds ? ps[pb * Ds + c] : pa[pb * Da + (c - Ds)];
        const float e = c < Ds ? es[(long)b * Ds + c] : ea[(long)b * Da + (c - Ds)];
        const float a = alpha[b];
        X[i] = p;
        X[n + i] = e;
        const float mx = ((a) * (e)) + ((((1.0f) - (a))) * (p));           // alpha*expert + (1-alpha)*policy
        X[2 * n + i] = mx;
        if (copies == 2) X[3 * n + i] = mx;
    }
}

// torch's binary_cross_entropy_with_logits: (1-y)*x + max(-x,0) + log(exp(-max(-x,0)) + exp(-x-max(-x,0)))
__device__ __forceinline__ float bce_logits(float x, float y) {
    const float mx = fmaxf(-x, 0.f);
    return (1.f - y) * x + mx + logf(expf(-mx) + expf(-x - mx));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// one workgroup: d[3B] -> seed[3B] (policy sigma/B, expert (sigma-1)/B, mixup 1), seed_bce[3B] (same, mixup 0),
// loss_out[0] = mean bce(expert, 1) + mean bce(policy, 0).  stacked: seed_bce has 4B rows, 1 on rows [2B, 3B) and 0 on
// [3B, 4B) -- the layer-3 column of the stacked weight-gradient product (acktr_gail_param_grads_f32).
__global__ __launch_bounds__(256) void gail_bce_seed_kernel(const float* d, int B, int stacked, float* seed, float* seed_bce,
                                                            float* loss_out) {
    __shared__ double sm[16];
    double lp = 0.0, le = 0.0;
    const float invB = 1.0f / (float)B;
    for (int i = threadIdx.x; i < 3 * B; i += 256) {
        const float x = d[i];
        float s, sb;
        if (i < B) { lp += (double)bce_logits(x, 0.f); s = sb = sigmoidf(x) * invB; }
        else if (i < 2 * B) { le += (double)bce_logits(x, 1.f); s = sb = (sigmoidf(x) - 1.f) * invB; }
        else { s = 1.f; sb = stacked ? 1.f : 0.f; }
        seed[i] = s;
        seed_bce[i] = sb;
    }
    if (stacked)
        for (int i = 3 * B + threadIdx.x; i < 4 * B; i += 256) seed_bce[i] = 0.f;
    lp = block_sum(lp, sm);
    le = block_sum(le, sm);
    if (threadIdx.x == 0) loss_out[0] = (float)(le / B) + (float)(lp / B);
}

// one wave per row: r = d pen / d g, pen partial; second stage below sums the row terms in order
__global__ __launch_bounds__(256) void gail_pen_seed_kernel(const float* g, int B, int D, float lambda_, float* r,
                                                            float* row_term) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float ss = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = g[(long)row * D + c]; ss += v * v; }
    ss = __shfl(wave_sum(ss), 0, 64);                 // wave_sum leaves the total in lane 0 only
    const float nrm = sqrtf(ss);
    const float dev = nrm - 1.f;
    // d/dg [ lambda/B * (||g|| - 1)^2 ] = lambda/B * 2 (||g|| - 1) * g / ||g||   (0 at ||g|| = 0, like torch's norm backward)
    const float k = nrm > 0.f ? lambda_ * 2.f * dev / ((float)B * nrm) : 0.f;
    for (int c = lane; c < D; c += 64) r[(long)row * D + c] = k * g[(long)row * D + c];
    if (lane == 0) row_term[row] = dev * dev;
}
__global__ __launch_bounds__(256) void gail_pen_sum_kernel(const float* row_term, int B, float lambda_, float* loss_out) {
    __shared__ double sm[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < B; i += 256) s += (double)row_term[i];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) loss_out[1] = lambda_ * (float)(s / B);
}

// both stages in one workgroup (B <= PEN_ONE_WG_MAX_B): one lane per row (D is the observation + action width, tens of
// columns; every lane's loads are independent, where a wave-per-row walk would serialise B/4 memory round trips); the row
// terms are summed in fp64
constexpr int PEN_ONE_WG_MAX_B = 4096;
__global__ __launch_bounds__(256) void gail_pen_one_wg_kernel(const float* g, int B, int D, float lambda_, float* r,
                                                              float* loss_out) {
    __shared__ double sm[16];
    double acc = 0.0;
    for (int row = threadIdx.x; row < B; row += 256) {
        const float* gr = g + (long)row * D;
        float ss = 0.f;
        for (int c = 0; c < D; ++c) ss += gr[c] * gr[c];
        const float nrm = sqrtf(ss);
        const float dev = nrm - 1.f;
        const float k = nrm > 0.f ? lambda_ * 2.f * dev / ((float)B * nrm) : 0.f;
        for (int c = 0; c < D; ++c) r[(long)row * D + c] = k * gr[c];
        acc += (double)(dev * dev);
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) loss_out[1] = lambda_ * (float)(acc / B);
}

// Stacked weight gradients dW[m, n] = sum_k A[k, m] X[k, n] (k = row of the stacked operands) for the two hidden layers:
// one workgroup per 32x32 tile of dW1 [H, D] or dW2 [H, H], 8 waves each over a 1/8 slice of k.  Both operands are
// contiguous along m / n, so a wave's 32x32x2 MFMA operands are plain coalesced dword loads (lane l: k = 2s + l/32,
// m = m0 + l%32), no LDS staging; the eight partial tiles meet in LDS and are added in wave order.
struct GailWgradJob { const float* a; const float* x; float* dw; int N; int tiles_n; };
__global__ __launch_bounds__(512) void gail_wgrad_tile_kernel(GailWgradJob j0, GailWgradJob j1, int tiles0, int rows, int H) {
    __shared__ float red[8][16][64];
    const bool first = (int)blockIdx.x < tiles0;
    const GailWgradJob& J = first ? j0 : j1;
    const int t = first ? blockIdx.x : blockIdx.x - tiles0;
    const int m0 = (t / J.tiles_n) * 32, n0 = (t % J.tiles_n) * 32;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, kh = lane >> 5;
    const int per = ((rows + 15) / 16) * 2;                           // k per wave, even
    const int k0 = wave * per, k1 = min(rows, k0 + per);
    const bool mok = m0 + li < H, nok = n0 + li < J.N;
    const float* ap = J.a + m0 + li;
    const float* xp = J.x + n0 + li;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    for (int k = k0; k < k1; k += 16) {
        float av[8], xv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int kk = k + 2 * s + kh;
            const bool ok = kk < k1;
            av[s] = (ok && mok) ? ap[(long)kk * H] : 0.f;
            xv[s] = (ok && nok) ? xp[(long)kk * J.N] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], xv[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) red[wave][v][lane] = acc[v];
    __syncthreads();
    // 1024 tile elements, 512 threads: element (v, lane) -> row 4*(lane/32) + (v&3) + 8*(v>>2), column lane%32
    for (int e = threadIdx.x; e < 1024; e += 512) {
        const int v = e >> 6, l = e & 63;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sum += red[w][v][l];
        const int m = m0 + 4 * (l >> 5) + (v & 3) + 8 * (v >> 2), n = n0 + (l & 31);
        if (m < H && n < J.N) J.dw[(long)m * J.N + n] = sum;
    }
}

// One wave per output: db_l[o] = sum of a_l[r, o] over the rows r outside [skip_lo, skip_hi) for columns [0, H) of a1,
// [H, 2H) of a2 (both [rows, H]) and the single column a3 [rows]; dw3[i] = sum over ALL rows of a3[r] * h2[r, i].
__global__ __launch_bounds__(64) void gail_bias_sums_kernel(const float* a1, const float* a2, const float* a3, const float* h2,
                                                            int rows, int H, int skip_lo, int skip_hi, float* b1, float* b2,
                                                            float* b3, float* dw3) {
    const int c = blockIdx.x, l = c / H;                              // l = 3: the last block
    float s = 0.f;
    if (l == 2) {
        const float* h = h2 + (c - 2 * H);
        for (int r = threadIdx.x; r < rows; r += 64) s += a3[r] * h[(long)r * H];
    } else {
        const float* a = l == 0 ? a1 + c : (l == 1 ? a2 + (c - H) : a3);
        const int ld = l == 3 ? 1 : H;
        for (int r = threadIdx.x; r < rows; r += 64)
            if (r < skip_lo || r >= skip_hi) s += a[(long)r * ld];
    }
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        if (l == 0) b1[c] = s; else if (l == 1) b2[c - H] = s; else if (l == 2) dw3[c - 2 * H] = s; else b3[0] = s;
    }
}

// element-wise links of the (double) backward through y = tanh(a), with h = tanh(a) stored:
//   mode 0: out0 = (1 - h^2) * in0                                                       (tanh' gate)
//   mode 1: out0 = (1 - h^2) * in0 ; out1 = in1 * in0              (d v1 and d t1' from d u1 (in0), v1 (in1))
//   mode 2: out0 = (1 - h^2) * in0 ; out1 = (1 - h^2) * (-2 h * w[col] * in0)   (d w3 terms and d a2 from d u2 (in0))
//   mode 3: out0 = (1 - h^2) * (in0 + (-2 h * in1))                           (d a1 from d h1 (in0) and d t1' (in1))
__global__ __launch_bounds__(256) void gail_tanh_link_kernel(int mode, const float* h, const float* in0, const float* in1,
                                                             const float* w, long n, int cols, float* out0, float* out1) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float hv = h[i], x = in0[i];
        const float gate = 1.f - hv * hv;
        if (mode == 0) {
            out0[i] = gate * x;
        } else if (mode == 1) {
            out0[i] = gate * x;
            out1[i] = in1[i] * x;
        } else if (mode == 2) {
            out0[i] = gate * x;
            out1[i] = gate * (-2.f * hv * (w[i % cols] * x));
        } else {
            out0[i] = gate * (x + (-2.f * hv * in1[i]));
        }
    }
}

// reward = log(sigmoid d) - log(1 - sigmoid d)   (gail.py:101-102, evaluated like the reference, not as d itself)
__global__ __launch_bounds__(256) void gail_reward_kernel(const float* d, long n, float* reward) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float s = sigmoidf(d[i]);
        reward[i] = logf(s) - logf(1.f - s);
    }
}

// running return over S consecutive steps of N environments: ret = ret*mask*gamma + reward (gail.py:107), one lane per env;
// writes every step's returns row and keeps the last in `ret_io`
__global__ __launch_bounds__(256) void gail_returns_kernel(const float* reward, const float* masks, int S, int N, float gamma,
                                                           float* ret_io, float* ret_steps) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float r = ret_io[n];
    for (int s = 0; s < S; ++s) {
        r = ((((r) * (masks[(long)s * N + n]))) * (gamma)) + (reward[(long)s * N + n]);
        ret_steps[(long)s * N + n] = r;
    }
    ret_io[n] = r;
}

// per-step batch moments in fp64 (numpy mean / var over the N environments): mom[s] = {mean, var}
__global__ __launch_bounds__(256) void gail_step_moments_kernel(const float* ret_steps, int N, double* mom) {
    __shared__ double sm[16];
    const float* row = ret_steps + (long)blockIdx.x * N;
    double s = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) s += (double)row[i];
    s = block_sum(s, sm);
    __shared__ double mean_s;
    if (threadIdx.x == 0) mean_s = s / N;
    __syncthreads();
    const double mean = mean_s;
    double v = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) { const double dlt = (double)row[i] - mean; v += dlt * dlt; }
    v = block_sum(v, sm);
    if (threadIdx.x == 0) { mom[2 * blockIdx.x] = mean; mom[2 * blockIdx.x + 1] = v / N; }
}

// reward[s, n] /= scale[s]
__global__ __launch_bounds__(256) void gail_scale_rows_kernel(float* reward, const float* scale, int S, int N) {
    const long n = (long)S * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) reward[i] = reward[i] / scale[i / N];
}

static int ew_grid(long n) {
    long b = (n + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}
}  // namespace acktr

using namespace acktr;

extern "C" int acktr_gail_mix_concat_f32(const float* policy_state, const float* policy_action, const float* expert_state,
                                         const float* expert_action, const float* alpha, int B, int Ds, int Da, float* x_out,
                                         void* stream) {
    ACKTR_ARG(policy_state && policy_action && expert_state && expert_action && alpha && x_out, "gail_mix_concat: null pointer");
    ACKTR_ARG(B > 0 && Ds > 0 && Da > 0, "gail_mix_concat: bad sizes");
    hipLaunchKernelGGL(gail_mix_concat_kernel, dim3(ew_grid((long)B * (Ds + Da))), dim3(256), 0, (hipStream_t)stream,
                       policy_state, policy_action, (const int64_t*)nullptr, expert_state, expert_action, alpha, B, Ds, Da, 1, x_out);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_gail_mix_concat_idx_f32(const float* state_rows, const float* action_rows, const int64_t* idx,
                                             const float* expert_state, const float* expert_action, const float* alpha,
                                             int B, int Ds, int Da, int mix_copies, float* x_out, void* stream) {
    ACKTR_ARG(state_rows && action_rows && expert_state && expert_action && alpha && x_out, "gail_mix_concat_idx: null pointer");
    ACKTR_ARG(B > 0 && Ds > 0 && Da > 0, "gail_mix_concat_idx: bad sizes");
    ACKTR_ARG(mix_copies == 1 || mix_copies == 2, "gail_mix_concat_idx: mix_copies must be 1 or 2");
    hipLaunchKernelGGL(gail_mix_concat_kernel, dim3(ew_grid((long)B * (Ds + Da))), dim3(256), 0, (hipStream_t)stream,
                       state_rows, action_rows, idx, expert_state, expert_action, alpha, B, Ds, Da, mix_copies, x_out);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_gail_bce_seed_f32(const float* d, int B, int stacked, float* seed, float* seed_bce, float* loss_out,
                                       void* stream) {
    ACKTR_ARG(d && seed && seed_bce && loss_out && B > 0, "gail_bce_seed: bad arguments");
    hipLaunchKernelGGL(gail_bce_seed_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d, B, stacked ? 1 : 0, seed, seed_bce,
                       loss_out);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_gail_pen_seed_f32(const float* g, int B, int D, float lambda_, float* r, float* row_scratch,
                                       float* loss_out, void* stream) {
    ACKTR_ARG(g && r && loss_out && B > 0 && D > 0, "gail_pen_seed: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (B <= PEN_ONE_WG_MAX_B) {
        hipLaunchKernelGGL(gail_pen_one_wg_kernel, dim3(1), dim3(256), 0, s, g, B, D, lambda_, r, loss_out);
        ACKTR_LAUNCH_CHECK();
        return 0;
    }
    ACKTR_ARG(row_scratch, "gail_pen_seed: B > %d needs row_scratch", PEN_ONE_WG_MAX_B);
    hipLaunchKernelGGL(gail_pen_seed_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, g, B, D, lambda_, r, row_scratch);
    ACKTR_LAUNCH_CHECK();
    hipLaunchKernelGGL(gail_pen_sum_kernel, dim3(1), dim3(256), 0, s, row_scratch, B, lambda_, loss_out);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

// Stacked weight gradients of the three layers in two launches: the 32x32 tiles of dW1 and dW2, and one wave per element
// of dw3 / the three bias gradients.
extern "C" int acktr_gail_param_grads_f32(const float* a1, const float* x, const float* a2, const float* h1, const float* a3,
                                          const float* h2, int rows, int D, int H, int skip_lo, int skip_hi, float* dw1,
                                          float* db1, float* dw2, float* db2, float* dw3, float* db3, void* stream) {
    ACKTR_ARG(a1 && x && a2 && h1 && a3 && h2 && dw1 && db1 && dw2 && db2 && dw3 && db3, "gail_param_grads: null pointer");
    ACKTR_ARG(rows > 0 && D > 0 && H > 0 && skip_lo >= 0 && skip_lo <= skip_hi && skip_hi <= rows, "gail_param_grads: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    const int tm = cdiv(H, 32);
    GailWgradJob j0{a1, x, dw1, D, cdiv(D, 32)}, j1{a2, h1, dw2, H, tm};
    hipLaunchKernelGGL(gail_wgrad_tile_kernel, dim3(tm * j0.tiles_n + tm * j1.tiles_n), dim3(512), 0, s, j0, j1,
                       tm * j0.tiles_n, rows, H);
    ACKTR_LAUNCH_CHECK();
    hipLaunchKernelGGL(gail_bias_sums_kernel, dim3(3 * H + 1), dim3(64), 0, s, a1, a2, a3, h2, rows, H, skip_lo, skip_hi,
                       db1, db2, db3, dw3);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_gail_tanh_link_f32(int mode, const float* h, const float* in0, const float* in1, const float* w, long n,
                                        int cols, float* out0, float* out1, void* stream) {
    ACKTR_ARG(mode >= 0 && mode <= 3 && h && in0 && out0 && n > 0 && cols > 0, "gail_tanh_link: bad arguments");
    ACKTR_ARG((mode != 1 && mode != 3) || in1, "gail_tanh_link: mode %d needs in1", mode);
    ACKTR_ARG((mode != 1 && mode != 2) || out1, "gail_tanh_link: mode %d needs out1", mode);
    ACKTR_ARG(mode != 2 || w, "gail_tanh_link: mode 2 needs w");
    hipLaunchKernelGGL(gail_tanh_link_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, mode, h, in0, in1, w, n, cols,
                       out0, out1);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_gail_reward_f32(const float* d, long n, float* reward, void* stream) {
    ACKTR_ARG(d && reward && n > 0, "gail_reward: bad arguments");
    hipLaunchKernelGGL(gail_reward_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, d, n, reward);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_gail_returns_moments_f32(const float* reward, const float* masks, int S, int N, float gamma, float* ret_io,
                                              float* ret_steps, double* moments, void* stream) {
    ACKTR_ARG(reward && masks && ret_io && ret_steps && moments && S > 0 && N > 0, "gail_returns_moments: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gail_returns_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, reward, masks, S, N, gamma, ret_io, ret_steps);
    ACKTR_LAUNCH_CHECK();
    hipLaunchKernelGGL(gail_step_moments_kernel, dim3(S), dim3(256), 0, s, ret_steps, N, moments);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

// RunningMeanStd.update_from_moments for S consecutive batches (gail.py:108-110 called once per rollout step): the merge
// is a scalar recurrence in fp64, so ONE lane walks the S steps (same operation order as the numpy restatement in
// a2c_ppo_acktr/algo/gail.py) and writes scale[s] = sqrt(var_after_step_s + 1e-8) for gail.py:110.
__global__ void gail_rms_merge_kernel(const double* moments, int S, double batch_count, double* state, float* scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double mean = state[0], var = state[1], count = state[2];
    for (int s = 0; s < S; ++s) {
        const double bm = moments[2 * s], bv = moments[2 * s + 1];
        const double delta = bm - mean;
        const double tot = count + batch_count;
        const double new_mean = mean + delta * batch_count / tot;
        const double m_2 = var * count + bv * batch_count + delta * delta * count * batch_count / tot;
        mean = new_mean;
        var = m_2 / tot;
        count = tot;
        scale[s] = (float)sqrt(var + 1e-8);
    }
    state[0] = mean; state[1] = var; state[2] = count;
}

extern "C" int acktr_gail_rms_merge_f64(const double* moments, int S, double batch_count, double* state_inout,
                                        float* scale_out, void* stream) {
    ACKTR_ARG(moments && state_inout && scale_out && S > 0 && batch_count > 0, "gail_rms_merge: bad arguments");
    hipLaunchKernelGGL(gail_rms_merge_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, moments, S, batch_count,
                       state_inout, scale_out);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_gail_scale_rows_f32(float* reward, const float* scale, int S, int N, void* stream) {
    ACKTR_ARG(reward && scale && S > 0 && N > 0, "gail_scale_rows: bad arguments");
    hipLaunchKernelGGL(gail_scale_rows_kernel, dim3(ew_grid((long)S * N)), dim3(256), 0, (hipStream_t)stream, reward, scale, S, N);
    ACKTR_LAUNCH_CHECK();
    return 0;
}
