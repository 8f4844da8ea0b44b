// One launch for everything between the FC features and their gradient on the NatureCNN + Categorical path:
//   head = feat Wh^T + bh            (critic 512->1 and action head 512->A as ONE [1+A, 512] layer; model.py:185,195,
//                                     distributions.py:69-73)
//   categorical log-prob / entropy, PPO / A2C / Fisher loss and its derivative (loss.hip's arithmetic, shared through
//                                     loss_terms.hpp; ppo.py:61-81, a2c_acktr.py:48-65)
//   dfeat = (dhead Wh) * (feat > 0)   (backward of the head and of the FC layer's ReLU)
//   dWh = dhead^T feat, dbh = sum dhead   (per-workgroup partials, finished by head_finalize_kernel in a fixed order)
// Layer by layer this was nine launches (skinny forward, loss + finalize, skinny dgrad, skinny wgrad partial + two
// slab-reduction passes + bias reduction) of 4-9 us each on a 1024-row minibatch, every one of them latency-bound.
// Here a wave owns whole rows: lane l holds floats [4l, 4l+4) and [256+4l, 256+4l+4) of the row and of each weight
// row in registers, so the features are read from HBM once and serve the dot products, the ReLU gate and the
// weight gradient; the 1+A dot products are butterfly-reduced (every lane ends with the same head row, so the loss
// arithmetic is wave-uniform).
#include "loss_terms.hpp"
#include "head_finalize.hpp"
#include "../../include/acktr_hip.h"

namespace acktr {

typedef float hf4 __attribute__((ext_vector_type(4)));
constexpr int HS_OUT = 8;          // 1 + A <= 8
constexpr int HS_H = 512;          // hidden size (model.py:173 default)
constexpr int HS_RPW = 2;          // rows a wave has in flight (minibatch 1024: 2 -> 128 workgroups, 11.2 us; 4 -> 64, 13.2 us)
constexpr int HS_WAVES = 4;
constexpr int HS_ROWS = HS_RPW * HS_WAVES;
constexpr int HS_TAIL = 12;        // per-slab tail: 8 bias sums, value loss, action loss, entropy, pad
constexpr int HS_MAX_SLABS = 128;

__device__ __forceinline__ float lane_bcast(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// Sum over the 64 lanes, the same bits in every lane: four DPP steps inside each 16-lane row (quad_perm [1,0,3,2],
// [2,3,0,1], row_half_mirror, row_mirror -- every step adds two partials that both partners hold, and the addition
// commutes), then the four row sums in row order through readlane.  (__shfl_xor is a ds_bpermute round trip per
// step: 6 dependent LDS round trips per sum were most of this kernel's time.)
__device__ __forceinline__ float wave_allsum(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    return ((lane_bcast(v, 0) + lane_bcast(v, 16)) + lane_bcast(v, 32)) + lane_bcast(v, 48);
}
__device__ __forceinline__ float dot4(const hf4& a, const hf4& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

// slabs: [gridDim.x][OUT*512 + HS_TAIL]
template <int ALGO>
__global__ __launch_bounds__(256) void head_step_kernel(acktr_loss_args a, const float* __restrict__ feat,
                                                        const float* __restrict__ wh, const float* __restrict__ bh,
                                                        float* __restrict__ dfeat, float* __restrict__ slabs,
                                                        long slab_stride, int OUT, int nfs, long fs_stride,
                                                        const float* __restrict__ fbias) {
    // nfs > 0: `feat` is not the finished feature matrix but the FC layer's nfs split-K slabs (acktr_linear_fwd_slabs_f32):
    // the features are relu(sum of the slabs in order + fbias), formed right here in the registers that feed the dot
    // products -- the FC forward's slab-reduction launch (4.7 us of the 557 us step) and its 2 MB round trip disappear.
    __shared__ float red[HS_WAVES - 1][HS_OUT][HS_H];          // 48 KB: waves 1..3 hand their dW partials to wave 0
    __shared__ float tail[HS_WAVES][HS_TAIL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int A = a.A, B = a.B;
    hf4 w[HS_OUT][2], acc[HS_OUT][2];
    float bias[HS_OUT], bs[HS_OUT];
#pragma unroll
    for (int o = 0; o < HS_OUT; ++o) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            w[o][q] = hf4{0.f, 0.f, 0.f, 0.f};
            if (o < OUT) w[o][q] = *reinterpret_cast<const hf4*>(wh + (long)o * HS_H + q * 256 + 4 * lane);
            acc[o][q] = hf4{0.f, 0.f, 0.f, 0.f};
        }
        bias[o] = (o < OUT && bh) ? bh[o] : 0.f;
        bs[o] = 0.f;
    }
    float vl = 0.f, al = 0.f, en = 0.f;
    for (int base = blockIdx.x * HS_ROWS; base < B; base += gridDim.x * HS_ROWS) {
        const int r0 = base + wave * HS_RPW;
        // rollout-side operands of the wave's rows: lane l fetches those of row r0 + l%4 (index, then the five gathered
        // values -- two dependent round trips for the whole chunk instead of two per row), handed out by readlane
        const int lr = r0 + (lane & (HS_RPW - 1)) < B ? r0 + (lane & (HS_RPW - 1)) : B - 1;
        const long src_l = a.gather_idx ? (long)a.gather_idx[lr] : (long)lr;
        hf4 x[HS_RPW][2];
#pragma unroll
        for (int r = 0; r < HS_RPW; ++r) {                      // all rows of the wave requested before the first is used
            const int rr = r0 + r < B ? r0 + r : B - 1;
#pragma unroll
            for (int q = 0; q < 2; ++q) x[r][q] = *reinterpret_cast<const hf4*>(feat + (long)rr * HS_H + q * 256 + 4 * lane);
        }
        if (nfs > 0) {                                          // (wave-uniform)
            for (int sl = 1; sl < nfs; ++sl) {
                hf4 y[HS_RPW][2];
#pragma unroll
                for (int r = 0; r < HS_RPW; ++r) {
                    const int rr = r0 + r < B ? r0 + r : B - 1;
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        y[r][q] = *reinterpret_cast<const hf4*>(feat + (long)sl * fs_stride + (long)rr * HS_H + q * 256 + 4 * lane);
                }
#pragma unroll
                for (int r = 0; r < HS_RPW; ++r)
#pragma unroll
                    for (int q = 0; q < 2; ++q) x[r][q] += y[r][q];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const hf4 fb = *reinterpret_cast<const hf4*>(fbias + q * 256 + 4 * lane);
#pragma unroll
                for (int r = 0; r < HS_RPW; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float v = x[r][q][j] + fb[j]; x[r][q][j] = v > 0.f ? v : 0.f; }
            }
        }
        const long act64 = ((const int64_t*)a.actions)[src_l];
        const int act_l = (act64 >= 0 && act64 < A) ? (int)act64 : -1;
        const float ol_l = a.old_logp ? a.old_logp[src_l] : 0.f, ov_l = a.old_values ? a.old_values[src_l] : 0.f;
        const float ret_l = a.returns ? a.returns[src_l] : 0.f, adv_l = a.adv ? a.adv[src_l] : 0.f;
        // ---- phase A: the 1+A dot products of every row, butterfly-reduced; lane l keeps the head row of ITS row
        float hp[HS_OUT];
#pragma unroll
        for (int o = 0; o < HS_OUT; ++o) hp[o] = 0.f;
#pragma unroll
        for (int r = 0; r < HS_RPW; ++r) {
            float p[HS_OUT];
#pragma unroll
            for (int o = 0; o < HS_OUT; ++o) p[o] = dot4(x[r][0], w[o][0]) + dot4(x[r][1], w[o][1]);
#pragma unroll
            for (int o = 0; o < HS_OUT; ++o) {
                p[o] = wave_allsum(p[o]) + bias[o];          // rows o >= OUT have zero weights and bias: no branch
                hp[o] = (lane & (HS_RPW - 1)) == r ? p[o] : hp[o];
            }
        }
        // ---- phase B: categorical_eval / categorical_bwd of loss_terms.hpp and the loss terms, ONCE per chunk with
        // lane l working on row r0 + l%4 (the arithmetic is ~1000 instructions per row: run wave-uniformly per row it
        // was the bulk of the kernel's time)
        float dh_l[HS_OUT];
        {
            float m = hp[1];
#pragma unroll
            for (int i = 1; i < HS_OUT - 1; ++i) if (i < A) m = fmaxf(m, hp[1 + i]);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < HS_OUT - 1; ++i) if (i < A) s += expf(hp[1 + i] - m);
            const float lse = m + logf(s);
            float ent = 0.f, za = 0.f;
#pragma unroll
            for (int i = 0; i < HS_OUT - 1; ++i) if (i < A) {
                const float lp = fmaxf(hp[1 + i] - lse, -3.402823466e+38f);
                ent -= lp * expf(lp);
                if (i == act_l) za = hp[1 + i];
            }
            const float logp = act_l >= 0 ? za - lse : 0.f;
            const LossTerms t = algo_terms<ALGO>(logp, hp[0], ol_l, ov_l, ret_l, adv_l, a.clip, a.use_clipped_value_loss);
            const float dlogp = t.dlogp * a.inv_batch;
            const float dent = (ALGO == 2) ? 0.f : -a.entropy_coef * a.inv_batch;
            dh_l[0] = ((ALGO == 2) ? 1.0f : a.value_coef) * t.dv * a.inv_batch;
#pragma unroll
            for (int i = 0; i < HS_OUT - 1; ++i) {
                dh_l[1 + i] = 0.f;
                if (i < A) {
                    const float lp = hp[1 + i] - lse, pr = expf(lp);
                    dh_l[1 + i] = dlogp * ((i == act_l ? 1.f : 0.f) - pr) - dent * pr * (lp + ent);
                }
            }
            const int b = r0 + lane;
            if (lane < HS_RPW && b < B) {                        // the lanes that own a row report it
                vl += t.vl; al += t.al; en += ent;
                if (a.logp_out) a.logp_out[b] = logp;
                float* hrow = const_cast<float*>(a.values) + (long)b * a.ld_values;
                float* drow = a.dvalues + (long)b * a.ld_dvalues;
#pragma unroll
                for (int o = 0; o < HS_OUT; ++o) if (o < OUT) { hrow[o] = hp[o]; drow[o] = dh_l[o]; }
            }
        }
        // ---- phase C: dfeat (gated by the FC layer's ReLU), dW and db partials, row by row with d(head) broadcast
#pragma unroll
        for (int r = 0; r < HS_RPW; ++r) {
            const int b = r0 + r;
            if (b < B) {                                         // wave-uniform
                float dh[HS_OUT];
#pragma unroll
                for (int o = 0; o < HS_OUT; ++o) dh[o] = lane_bcast(dh_l[o], r);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    hf4 g = hf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int o = 0; o < HS_OUT; ++o) g += dh[o] * w[o][q];
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = x[r][q][j] > 0.f ? g[j] : 0.f;
                    *reinterpret_cast<hf4*>(dfeat + (long)b * HS_H + q * 256 + 4 * lane) = g;
#pragma unroll
                    for (int o = 0; o < HS_OUT; ++o) acc[o][q] += dh[o] * x[r][q];
                }
#pragma unroll
                for (int o = 0; o < HS_OUT; ++o) bs[o] += dh[o];
            }
        }
    }
    // the loss sums live in lanes 0..3 (one row each per chunk): add them in lane order
    vl = ((lane_bcast(vl, 0) + lane_bcast(vl, 1)) + lane_bcast(vl, 2)) + lane_bcast(vl, 3);
    al = ((lane_bcast(al, 0) + lane_bcast(al, 1)) + lane_bcast(al, 2)) + lane_bcast(al, 3);
    en = ((lane_bcast(en, 0) + lane_bcast(en, 1)) + lane_bcast(en, 2)) + lane_bcast(en, 3);
    // ---- workgroup partial: waves 1..3 -> LDS, wave 0 adds them in wave order and writes the slab
    if (wave > 0) {
#pragma unroll
        for (int o = 0; o < HS_OUT; ++o)
#pragma unroll
            for (int q = 0; q < 2; ++q) *reinterpret_cast<hf4*>(&red[wave - 1][o][q * 256 + 4 * lane]) = acc[o][q];
    }
    if (lane == 0) {
#pragma unroll
        for (int o = 0; o < HS_OUT; ++o) tail[wave][o] = bs[o];
        tail[wave][8] = vl; tail[wave][9] = al; tail[wave][10] = en; tail[wave][11] = 0.f;
    }
    __syncthreads();
    if (wave != 0) return;
    float* slab = slabs + (long)blockIdx.x * slab_stride;
#pragma unroll
    for (int o = 0; o < HS_OUT; ++o) {
        if (o < OUT) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                hf4 v = acc[o][q];
#pragma unroll
                for (int k = 0; k < HS_WAVES - 1; ++k) v += *reinterpret_cast<const hf4*>(&red[k][o][q * 256 + 4 * lane]);
                *reinterpret_cast<hf4*>(slab + (long)o * HS_H + q * 256 + 4 * lane) = v;
            }
        }
    }
    if (lane < HS_TAIL) slab[(long)OUT * HS_H + lane] = ((tail[0][lane] + tail[1][lane]) + tail[2][lane]) + tail[3][lane];
}

// Fixed-order sum of the workgroup slabs (head_finalize.hpp)
__global__ __launch_bounds__(256) void head_finalize_kernel(const acktr_head_fin f) { (void)head_finalize_block(f, blockIdx.x); }

static int head_slabs(int B) {
    int nb = cdiv(B, HS_ROWS);
    return nb > HS_MAX_SLABS ? HS_MAX_SLABS : nb;
}

}  // namespace acktr

using namespace acktr;

extern "C" int acktr_head_step_supported(int B, int H, int A, int dist) {
    return (B > 0 && H == HS_H && A >= 1 && 1 + A <= HS_OUT && dist == 0) ? 1 : 0;
}

extern "C" long acktr_head_step_workspace_bytes(int B, int H, int A) {
    return (long)head_slabs(B) * ((long)(1 + A) * H + HS_TAIL) * (long)sizeof(float);
}

static int head_step_launch(const acktr_loss_args* args, const float* feat, const float* wh, const float* bh, float* dfeat,
                            float* dwh, float* dbh, int H, void* stream, acktr_head_fin* defer, int nfs = 0,
                            long fs_stride = 0, const float* fbias = nullptr) {
    ACKTR_ARG(args, "head_step: null args");
    const acktr_loss_args& a = *args;
    ACKTR_ARG(acktr_head_step_supported(a.B, H, a.A, a.dist), "head_step: unsupported shape B=%d H=%d A=%d dist=%d", a.B, H,
              a.A, a.dist);
    ACKTR_ARG(a.algo >= 0 && a.algo <= 2, "head_step: bad algo");
    ACKTR_ARG(feat && wh && dfeat && dwh && a.values && a.dvalues && a.actions, "head_step: null tensor");
    ACKTR_ARG(a.ld_values >= 1 + a.A && a.ld_dvalues >= 1 + a.A, "head_step: head rows need 1+A floats");
    ACKTR_ARG(((((uintptr_t)feat) | ((uintptr_t)wh) | ((uintptr_t)dfeat) | ((uintptr_t)a.workspace)) & 15) == 0,
              "head_step: feat, wh, dfeat and the workspace must be 16-byte aligned");
    ACKTR_ARG(a.algo != 0 || (a.old_logp && a.old_values && a.returns && a.adv), "head_step: PPO needs rollout arrays");
    ACKTR_ARG(a.algo != 1 || a.returns, "head_step: A2C needs returns");
    ACKTR_ARG(a.algo != 2 || a.adv, "head_step: Fisher loss needs the noise tensor in `adv`");
    ACKTR_ARG(a.workspace && a.workspace_bytes >= acktr_head_step_workspace_bytes(a.B, H, a.A), "head_step: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int OUT = 1 + a.A, nb = head_slabs(a.B);
    const long stride = (long)OUT * H + HS_TAIL;
    float* slabs = (float*)a.workspace;
    ACKTR_ARG(nfs == 0 || (nfs >= 1 && nfs <= 16 && fbias && fs_stride >= (long)a.B * H && fs_stride % 4 == 0 &&
                           ((uintptr_t)fbias & 15) == 0), "head_step: bad feature-slab arguments");
    if (a.algo == 0) hipLaunchKernelGGL((head_step_kernel<0>), dim3(nb), dim3(256), 0, s, a, feat, wh, bh, dfeat, slabs, stride, OUT, nfs, fs_stride, fbias);
    else if (a.algo == 1) hipLaunchKernelGGL((head_step_kernel<1>), dim3(nb), dim3(256), 0, s, a, feat, wh, bh, dfeat, slabs, stride, OUT, nfs, fs_stride, fbias);
    else hipLaunchKernelGGL((head_step_kernel<2>), dim3(nb), dim3(256), 0, s, a, feat, wh, bh, dfeat, slabs, stride, OUT, nfs, fs_stride, fbias);
    ACKTR_LAUNCH_CHECK();
    acktr_head_fin f;
    f.slabs = slabs; f.nb = nb; f.stride = stride; f.n4 = (int)(stride / 4); f.nw = OUT * H; f.OUT = OUT;
    f.gw = dwh; f.gb = dbh; f.loss_acc = a.loss_acc; f.inv_batch = a.inv_batch;
    if (defer) { *defer = f; return 0; }
    hipLaunchKernelGGL(head_finalize_kernel, dim3(cdiv(f.n4, 32)), dim3(256), 0, s, f);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_head_step_f32(const acktr_loss_args* args, const float* feat, const float* wh, const float* bh,
                                   float* dfeat, float* dwh, float* dbh, int H, void* stream) {
    return head_step_launch(args, feat, wh, bh, dfeat, dwh, dbh, H, stream, nullptr);
}

extern "C" int acktr_head_step_deferred_f32(const acktr_loss_args* args, const float* feat, const float* wh, const float* bh,
                                            float* dfeat, float* dwh, float* dbh, int H, acktr_head_fin* pending,
                                            void* stream) {
    ACKTR_ARG(pending && dbh, "head_step_deferred: null pointer");
    return head_step_launch(args, feat, wh, bh, dfeat, dwh, dbh, H, stream, pending);
}

extern "C" int acktr_head_step_slabs_f32(const acktr_loss_args* args, const float* feat_slabs, int nslabs, long slab_stride,
                                         const float* fc_bias, const float* wh, const float* bh, float* dfeat, float* dwh,
                                         float* dbh, int H, acktr_head_fin* pending, void* stream) {
    ACKTR_ARG(nslabs >= 1, "head_step_slabs: nslabs must be >= 1");
    ACKTR_ARG(!pending || dbh, "head_step_slabs: a deferred finalize needs dbh");
    return head_step_launch(args, feat_slabs, wh, bh, dfeat, dwh, dbh, H, stream, pending, nslabs, slab_stride, fc_bias);
}
