/* acktr_hip.h -- C-ABI of the MI355X (gfx950) learner-update library.
 *
 * The reference (ikostrikov/pytorch-a2c-ppo-acktr-gail) has no FFI of its own:
 * its update path is pure Python over torch ops.  This header is therefore the
 * drop-in boundary this build defines for that path -- one entry point per
 * arithmetic block of the reference, each citing the reference lines it
 * replaces.  The host-side mirror of the reference classes
 * (pytorch-a2c-ppo-acktr-gail_amd/a2c_ppo_acktr) binds these with ctypes;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory unless named host_*;
 *   - all buffers are caller-owned; no entry point allocates or frees;
 *   - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous;
 *   - return 0 on success, -1 on an argument error, a positive hipError_t on a
 *     launch failure; acktr_last_error() returns the message (thread-local);
 *   - fp32 everywhere (int64 for discrete actions and gather indices).
 */
#ifndef ACKTR_HIP_H
#define ACKTR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACKTR_ABI_VERSION 1

int acktr_abi_version(void);
const char* acktr_target_arch(void);
const char* acktr_last_error(void);

/* ---- rollout storage --------------------------------------------------------------------- */

/* RolloutStorage.compute_returns -- a2c_ppo_acktr/storage.py:66-105.
 * rewards [T,N]; value_preds, returns, masks, bad_masks [T+1,N]; next_value [N].
 * gamma32 = (float)gamma, gamma_lambda32 = (float)(gamma*gae_lambda) (product taken in double).
 * Bit-exact with the reference's fp32 op order in all four modes; writes returns[0..T-1] and
 * value_preds[T] (GAE) or returns[T] (non-GAE), like the reference. */
int acktr_compute_returns_f32(const float* rewards, float* value_preds, float* returns, const float* masks,
                              const float* bad_masks, const float* next_value, int T, int N, float gamma32,
                              float gamma_lambda32, int use_gae, int use_proper_time_limits, void* stream);

/* Advantage normalisation -- a2c_ppo_acktr/algo/ppo.py:35-37.
 * adv[i] = ((returns[i]-value_preds[i]) - mean) / (std_unbiased + 1e-5), i < n (= T*N).
 * workspace: acktr_adv_normalize_workspace_bytes() bytes of device memory. */
long acktr_adv_normalize_workspace_bytes(void);
int acktr_adv_normalize_f32(const float* returns, const float* value_preds, float* adv, long n, void* workspace,
                            void* stream);

/* The same normalisation for callers that cannot allocate (code under hipGraph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  NO counterpart in the reference.  It takes no workspace: the block partials
 * live in a 16 KiB file-scope __device__ array of the library (one copy per GPU, held until the process ends),
 * whose per-device address is looked up once with hipGetSymbolAddress.  The call itself allocates and frees nothing.
 * Same kernels, same fixed reduction order -> bit-identical to acktr_adv_normalize_f32.  n must be
 * <= acktr_adv_normalize_noalloc_max_n() (2^28 elements), else -1 with a message.  All calls on one GPU share that
 * array: calls that overlap on one GPU (different streams) race; nothing in it has to start at zero.
 * The device that owns `stream` must be the current device (checked; -1 otherwise): the scratch address is per device.
 * acktr_adv_normalize_noalloc_init() resolves the current device's scratch address ahead of time -- call it once per
 * device OUTSIDE any stream capture (the first compute call on a device would otherwise do the lookup itself, which is
 * not something to do under capture); it launches nothing and allocates nothing. */
long acktr_adv_normalize_noalloc_max_n(void);
int acktr_adv_normalize_noalloc_init(void);
int acktr_adv_normalize_noalloc_f32(const float* returns, const float* value_preds, float* adv, long n, void* stream);

/* Minibatch gather -- a2c_ppo_acktr/storage.py:127-140 (x.view(-1, ...)[indices]).
 * dst[i,:] = src[idx[i],:], rows of row_bytes bytes (multiple of 4); idx on device, 0 <= idx < src_rows. */
int acktr_gather_rows(const void* src, const int64_t* idx, void* dst, long rows, long row_bytes, long src_rows,
                      void* stream);

/* ---- policy network operators (fp32 MFMA implicit GEMM) ---------------------------------- */

/* nn.Conv2d forward of CNNBase -- a2c_ppo_acktr/model.py:176-179,190.
 * y[B,OC,OH,OW] = relu?( in_scale * conv(x, w) + bias ), NCHW, no padding.  Supported geometries:
 * (H=W=84,K=8,S=4), (H=W=20,K=4,S=2), (H=W=9,K=3,S=1); any C, OC.  in_scale folds `inputs / 255.0`.
 * gather_idx (nullable, int64[B]): image b is read from x + gather_idx[b]*C*H*W -- the minibatch
 * gather of storage.py:127 fused into the loader.  x, w 16-byte aligned.  `workspace` (nullable, 16-byte
 * aligned, ..._workspace_bytes) holds the weights re-ordered for the LDS-stationary kernel; without it the
 * generic tile-GEMM kernel runs.  prepacked = 1: the workspace already holds this layer's re-ordered weights (written
 * by acktr_conv2d_pack_weights_f32 since `w` last changed) and the re-ordering launch is skipped. */
long acktr_conv2d_fwd_workspace_bytes(int C, int H, int W, int OC, int KH, int KW, int S);
int acktr_conv2d_fwd_f32(const float* x, const int64_t* gather_idx, const float* w, const float* bias, float* y,
                         int B, int C, int H, int W, int OC, int KH, int KW, int S, float in_scale, int relu,
                         void* workspace, long workspace_bytes, int prepacked, void* stream);

/* autograd backward of the same Conv2d w.r.t. weight and bias: dw[OC,C,KH,KW] = in_scale * sum dy (x) patches,
 * db[OC] = sum dy (db nullable).  Split-K slabs live in `workspace` (see ..._workspace_bytes); the
 * slab reduction is fixed-order, so results are run-to-run deterministic. */
long acktr_conv2d_wgrad_workspace_bytes(int B, int C, int H, int W, int OC, int KH, int KW, int S);
int acktr_conv2d_wgrad_f32(const float* x, const int64_t* gather_idx, const float* dy, float* dw, float* db, int B,
                           int C, int H, int W, int OC, int KH, int KW, int S, float in_scale, void* workspace,
                           long workspace_bytes, void* stream);
/* A weight gradient whose workgroup slabs are written but not yet summed: dst[0 .. 4*n4) = scale * sum over the nslabs
 * slabs (stride floats apart) in the fixed order of the stand-alone reduction.  dst == NULL: nothing pending. */
typedef struct acktr_slab_fin {
    float* dst; const float* slabs; int nslabs; long n4; long stride; float scale;
} acktr_slab_fin;
/* The same weight gradient with its final slab sum left PENDING when the layer runs the stationary-accumulator kernel
 * whose slabs one launch can sum and db directly follows dw (the policy's three conv layers at minibatches >= 256):
 * `*pending` then describes it for acktr_clip_adam_dev_head_f32 or acktr_conv2d_dgrad_carry_f32 and dw / db are NOT
 * complete until that call; otherwise the call
 * completes as acktr_conv2d_wgrad_f32 does and pending->dst is NULL.  No counterpart in the reference. */
int acktr_conv2d_wgrad_deferred_f32(const float* x, const int64_t* gather_idx, const float* dy, float* dw, float* db,
                                    int B, int C, int H, int W, int OC, int KH, int KW, int S, float in_scale,
                                    void* workspace, long workspace_bytes, acktr_slab_fin* pending, void* stream);

/* autograd backward w.r.t. the input: dx[B,C,H,W] = conv_transpose(dy, w) (* (x_act > 0) when x_act is
 * given: the ReLU backward of the layer that produced x, model.py:177-179).  `workspace` (16-byte aligned,
 * ..._workspace_bytes) receives the re-ordered weights and the zero-padded copy of dy; prepacked as for the forward. */
long acktr_conv2d_dgrad_workspace_bytes(int B, int C, int H, int W, int OC, int KH, int KW, int S);
int acktr_conv2d_dgrad_f32(const float* dy, const float* w, const float* x_act, float* dx, int B, int C, int H, int W,
                           int OC, int KH, int KW, int S, void* workspace, long workspace_bytes, int prepacked,
                           void* stream);
/* The same with the zero-padding passes between chained dgrads folded into the producers (launch scheduling and data
 * layout only; no counterpart in the reference):
 *   flags bit 0 = prepacked (as above); bit 1 = the zero-padded dY ALREADY sits in `workspace` at
 *   acktr_conv2d_dgrad_padded_dy_offset_bytes(...) ([B, OC, 11, 11] planes for the two stationary geometries: 7x7 dY
 *   padded by 2, 9x9 dY padded by 1), written there by the producer of dY; `dy` may then be NULL and the padding launch
 *   is skipped.  The caller keeps the borders of that region zero (allocate it zeroed; producers write interiors only).
 *   dx_padded (nullable; 3x3 stride-1 stationary path only): dx is ALSO written as [B, C, H+2*pad, W+2*pad] planes,
 *   interior only -- point it at the next layer's padded-dY region.
 * ..._padded_dy_offset_bytes returns -1 for a shape whose kernel reads dY as it is -- NatureCNN's third layer (64 <- 32
 * channels, 3x3 stride 1, model.py:179) runs in overlap-add form over the 49 real positions of an image
 * (csrc/conv3_dgrad_oa.hpp; ACKTR_D3_OA=0 in the environment selects the gather over the padded dY instead): pass `dy`
 * and leave flags bit 1 clear then; dx_padded is served all the same. */
long acktr_conv2d_dgrad_padded_dy_offset_bytes(int C, int H, int W, int OC, int KH, int KW, int S);
int acktr_conv2d_dgrad_ex_f32(const float* dy, const float* w, const float* x_act, float* dx, int B, int C, int H, int W,
                              int OC, int KH, int KW, int S, void* workspace, long workspace_bytes, int flags,
                              float* dx_padded, int pad, void* stream);
/* acktr_conv2d_dgrad_ex_f32 with a PENDING weight-gradient slab sum of the launch before it riding along (`carry`,
 * nullable; as left by acktr_conv2d_wgrad_deferred_f32): extra workgroups of the stationary-weight launch sum the slabs
 * into carry->dst while the last, partial round of the input gradient's work items leaves most CUs idle -- one
 * dependent launch fewer per layer and step.  carry->dst is complete when this call's stream work is.  Shapes off the
 * stationary path run the sum as a launch of its own first.  Launch scheduling only; no counterpart in the reference. */
int acktr_conv2d_dgrad_carry_f32(const float* dy, const float* w, const float* x_act, float* dx, int B, int C, int H,
                                 int W, int OC, int KH, int KW, int S, void* workspace, long workspace_bytes, int flags,
                                 float* dx_padded, int pad, const acktr_slab_fin* carry, void* stream);
/* The ReLU gate as BITS (data layout only; no counterpart in the reference, whose autograd keeps the whole activation
 * for ReLU's backward, model.py:177).  The forward of the 8x8 stride-4 layer on its stationary-weight path (workspace
 * given) also writes gate_mask: word [r / 32][oc] (uint32, OC words per 32 positions, r = b*OH*OW + oy*OW + ox) holds
 * bit (r % 32) = (y[b, oc, oy, ox] > 0); acktr_conv2d_gate_mask_bytes(B, OC, OH, OW) bytes, 16-byte aligned.  The input
 * gradient of the FOLLOWING 4x4 stride-2 layer (32 input channels, 20x20) reads it in place of x_act when flags bit 2 of
 * acktr_conv2d_dgrad_ex_f32 / _carry_f32 is set (x_act then points at the mask): 1.6 MB instead of 52 MB at 1024
 * frames.  Results are bit-identical to the float gate.  Other shapes reject the mask. */
long acktr_conv2d_gate_mask_bytes(int B, int OC, int OH, int OW);
int acktr_conv2d_fwd_gate_f32(const float* x, const int64_t* gather_idx, const float* w, const float* bias, float* y,
                              int B, int C, int H, int W, int OC, int KH, int KW, int S, float in_scale, int relu,
                              void* workspace, long workspace_bytes, int prepacked, uint32_t* gate_mask, void* stream);


/* Re-ordered weights of up to 8 conv layers in ONE launch (no counterpart call in the reference: it is the weight
 * preparation the forward / dgrad entry points above otherwise launch themselves, model.py:176-179).  Each job names a
 * layer (C, H, W, OC, KH, KW, S as for the matching entry point; dgrad = 0 forward, 1 input-gradient) and that entry
 * point's workspace; the layer must be one the LDS-stationary kernel handles (NatureCNN's shapes), else -1. */
typedef struct acktr_conv_pack_job {
    const float* w; void* workspace; long workspace_bytes;
    int C, H, W, OC, KH, KW, S, dgrad;
} acktr_conv_pack_job;
int acktr_conv2d_pack_weights_f32(const acktr_conv_pack_job* jobs, int count, void* stream);

/* nn.Linear forward -- model.py:180,185,208-216, distributions.py:69,83.
 * y[B,OUT] = act(x[B,IN] w[OUT,IN]^T + bias);  act: 0 none, 1 relu, 2 tanh.  workspace (nullable) enables split-K
 * for shapes with few output tiles and a long reduction (the 1568->512 layer). */
long acktr_linear_fwd_workspace_bytes(int B, int IN, int OUT);
int acktr_linear_fwd_f32(const float* x, const float* w, const float* bias, float* y, int B, int IN, int OUT, int act,
                         void* workspace, long workspace_bytes, void* stream);
/* dx[B,IN] (=|+=) (dy[B,OUT] w[OUT,IN]) * gate;  gate: 0 none, 1 relu' (gate_src > 0), 2 tanh' (1 - gate_src^2),
 * gate_src[B,IN] = the activation that feeds this layer. */
int acktr_linear_dgrad_f32(const float* dy, const float* w, const float* gate_src, int gate, float* dx, int B, int IN,
                           int OUT, int accumulate, void* stream);
/* The same, also writing dx as zero-padded planes: row b of dx read as [C, OH, OW] (IN == C*OH*OW) goes to
 * dx_padded[b, c, y+pad, x+pad] of [B, C, OH+2*pad, OW+2*pad] (interior only; the caller keeps the borders zero) --
 * the dY layout of the conv dgrad behind an FC layer (see acktr_conv2d_dgrad_ex_f32).  dx_padded nullable. */
int acktr_linear_dgrad_ex_f32(const float* dy, const float* w, const float* gate_src, int gate, float* dx, int B, int IN,
                              int OUT, int accumulate, float* dx_padded, int C, int OH, int OW, int pad, void* stream);
/* dw[OUT,IN] = dy^T x ; db[OUT] = column sums of dy (db nullable). */
long acktr_linear_wgrad_workspace_bytes(int B, int IN, int OUT);
int acktr_linear_wgrad_f32(const float* dy, const float* x, float* dw, float* db, int B, int IN, int OUT,
                           void* workspace, long workspace_bytes, void* stream);

/* Strided fp32 GEMM  C[M,N] (=|+=) alpha * sum_k A(m,k) B(n,k); each operand contiguous along k
 * (stride_k == 1) or along its row index (stride_row == 1).  Used for the KFAC Kronecker-eigenbasis
 * products of a2c_ppo_acktr/algo/kfac.py:221-224. */
int acktr_gemm_f32(const float* a, long a_stride_row, long a_stride_k, const float* b, long b_stride_row,
                   long b_stride_k, float* c, long ldc, int M, int N, int K, float alpha, int accumulate,
                   void* stream);

/* ---- action distributions and losses ------------------------------------------------------ */

/* Fused loss + backward seeds for one (mini)batch.  Rollout-side arrays (actions, old_logp, old_values,
 * returns, adv) are indexed through gather_idx (nullable) -- storage.py:127-140 without the copies.
 *   algo 0: PPO clipped surrogate + (clipped) value loss + entropy        -- algo/ppo.py:61-81
 *   algo 1: A2C  value_loss = mean((R-v)^2), action_loss = -mean(adv.detach()*logp) -- algo/a2c_acktr.py:48-51,71-72
 *   algo 2: ACKTR Fisher loss -mean(logp) - mean((v - (v+noise).detach())^2); `adv` carries the noise
 *                                                                         -- algo/a2c_acktr.py:56-65
 *   dist 0: Categorical(logits)   -- distributions.py:18-29,71-73 (torch Categorical log_prob / entropy)
 *   dist 1: DiagGaussian(mean, exp(logstd)) -- distributions.py:36-41,86-95 (torch Normal), A <= 16
 * Outputs: d(loss)/d(values), d(loss)/d(dist_in) (, d(loss)/d(logstd)), with
 * loss = value_coef*value_loss + action_loss - entropy_coef*entropy and means taken as sum*inv_batch
 * (inv_batch = 1/global batch, so data-parallel shards sum to the full-batch gradient);
 * loss_acc[0..2] += (value_loss, action_loss, entropy) of this batch (as reported by update()). */
typedef struct acktr_loss_args {
    int algo, dist, B, A;
    const float* values;   long ld_values;
    const float* dist_in;  long ld_dist;
    const float* logstd;
    const void* actions;
    const float* old_logp; const float* old_values; const float* returns; const float* adv;
    const int64_t* gather_idx;
    float clip, value_coef, entropy_coef, inv_batch;
    int use_clipped_value_loss;
    float* dvalues;  long ld_dvalues;
    float* ddist;    long ld_ddist;
    float* dlogstd;
    float* dlogstd_rows;   /* nullable [B,A]: per-sample d(loss)/d(logstd) (grad_output of the logstd AddBias, for KFAC) */
    float* logp_out;
    float* loss_acc;
    void* workspace; long workspace_bytes;
} acktr_loss_args;
long acktr_policy_loss_workspace_bytes(int B, int A);
int acktr_policy_loss_f32(const acktr_loss_args* args, void* stream);

/* The NatureCNN + Categorical step between the FC features and their gradient in one launch (+ a fixed-order finalize):
 * head = feat Wh^T + bh with Wh = [critic_linear.weight; dist.linear.weight] ([1+A, H], adjacent in the flat buffer)
 * -- model.py:185,195, distributions.py:69-73; then the loss of acktr_policy_loss_f32 (same args; `values` / `dvalues`
 * are OUTPUTS here: the [B, ld] head and d(loss)/d(head) rows, column 0 the value, 1..A the logits; `dist_in`, `ddist`
 * are ignored); dfeat[B,H] = (dhead Wh) * (feat > 0) (the FC layer's ReLU, model.py:180); dwh[1+A,H] = dhead^T feat,
 * dbh[1+A] = column sums of dhead (autograd of the two nn.Linear).  H == 512, 1+A <= 8, dist 0 only
 * (acktr_head_step_supported); args->workspace >= acktr_head_step_workspace_bytes(B, H, A), 16-byte aligned. */
int acktr_head_step_supported(int B, int H, int A, int dist);
long acktr_head_step_workspace_bytes(int B, int H, int A);
int acktr_head_step_f32(const acktr_loss_args* args, const float* feat, const float* wh, const float* bh, float* dfeat,
                        float* dwh, float* dbh, int H, void* stream);
/* The same step with its second launch (the fixed-order sum of the workgroup slabs into dwh / dbh / loss_acc) left
 * PENDING: `*pending` (host memory) describes it, and acktr_clip_adam_dev_head_f32 runs it as extra workgroups of its
 * gradient-norm launch -- one dependent launch fewer per minibatch step (algo/ppo.py:57-84 as one sequence).  dfeat is
 * complete on return of the stream work; dwh / dbh / loss_acc are NOT until the pending finalize has run.  No
 * counterpart in the reference. */
typedef struct acktr_head_fin {
    const float* slabs; int nb; long stride; int n4; int nw; int OUT;
    float* gw; float* gb; float* loss_acc; float inv_batch;
} acktr_head_fin;
int acktr_head_step_deferred_f32(const acktr_loss_args* args, const float* feat, const float* wh, const float* bh,
                                 float* dfeat, float* dwh, float* dbh, int H, acktr_head_fin* pending, void* stream);

/* The head step fed by the FC layer's split-K slabs instead of finished features (model.py:180 `Linear(32*7*7, 512)` +
 * ReLU, then model.py:185,195 / distributions.py:69-73 as above): acktr_linear_fwd_slabs_f32 runs only the product
 * x W^T into nslabs partial [B, OUT] slabs (slab stride B*OUT floats; *nslabs_out >= 1; workspace >=
 * acktr_linear_fwd_slabs_workspace_bytes), and acktr_head_step_slabs_f32 forms features = relu(sum of the slabs in order +
 * fc_bias) in the registers that feed its dot products.  The features are never written: only the head step consumes
 * them on this path (the FC weight gradient reads dfeat and the FC input).  pending: nullable, as in
 * acktr_head_step_deferred_f32.  No counterpart in the reference (one torch op per layer there). */
long acktr_linear_fwd_slabs_workspace_bytes(int B, int IN, int OUT);
int acktr_linear_fwd_slabs_f32(const float* x, const float* w, int B, int IN, int OUT, void* workspace,
                               long workspace_bytes, int* nslabs_out, void* stream);
int acktr_head_step_slabs_f32(const acktr_loss_args* args, const float* feat_slabs, int nslabs, long slab_stride,
                              const float* fc_bias, const float* wh, const float* bh, float* dfeat, float* dwh, float* dbh,
                              int H, acktr_head_fin* pending, void* stream);

/* dist.log_probs(action) and dist.entropy() per sample -- model.py:74-77; entropy_mean (nullable, one float) receives
 * dist.entropy().mean() of model.py:77 (fixed summation order). */
int acktr_dist_fwd_f32(int dist, const float* dist_in, long ld, const float* logstd, const void* actions, int B, int A,
                       float* logp, float* entropy, float* entropy_mean, void* stream);
/* Rollout-side head of Policy.act -- model.py:54-66, distributions.py:22-25,28-29,37-44: the mode (deterministic != 0)
 * or one sample per row, and dist.log_probs(action) of it.  Categorical: action_out int64[B]; deterministic == 0:
 * inverse CDF of softmax(logits) at the caller's uniform noise[B] in [0,1) (same distribution as the reference, its
 * own stream); deterministic == 2: argmax_i p_i / noise[b,i] with the caller's Exp(1) draws noise[B,A] -- the
 * one-sample fast path of torch.multinomial that FixedCategorical.sample (distributions.py:19-20) reaches, so with
 * noise = torch.empty(B, A).exponential_(1) from the CPU default generator the actions ARE the reference CPU path's.
 * DiagGaussian: action_out float[B,A]; sample (0 or 2) = mean + exp(logstd) * noise[B,A] with the caller's standard
 * normal noise (torch.normal's formula, distributions.py:41-44 through torch.distributions.Normal.sample). */
int acktr_dist_act_f32(int dist, const float* dist_in, long ld, const float* logstd, const float* noise,
                       int deterministic, int B, int A, void* action_out, float* logp, void* stream);
/* their backward: ddist = dlogp[b]*dlogp/d(dist_in) + dentropy[b]*dentropy/d(dist_in) (either seed nullable);
 * gaussian also returns dlogstd[A] (workspace >= acktr_policy_loss_workspace_bytes(B, A)) and, if dlogstd_rows is not
 * null, its per-row terms [B,A]: the grad_output of the logstd AddBias module that KFAC's backward hook reads
 * (algo/kfac.py:161-177 through distributions.py:90-95). */
int acktr_dist_bwd_f32(int dist, const float* dist_in, long ld, const float* logstd, const void* actions,
                       const float* dlogp, const float* dentropy, int B, int A, float* ddist, long ldd, float* dlogstd,
                       float* dlogstd_rows, void* workspace, long workspace_bytes, void* stream);

/* ---- frames stored as bytes on the device --------------------------------------------------------------------
 * The reference keeps observations as float32 holding 0..255 (storage.py:12, envs.py:176,188; model.py:190 divides by
 * 255): these entry points are the first NatureCNN layer (model.py:176: Conv2d(4, 32, 8, stride=4) on 84x84 frames)
 * reading uint8 frames [Bsrc,4,84,84] instead -- a quarter of the traffic of the layer's forward, its weight gradient
 * and its KFAC input factor.  The byte -> float conversion is exact and the products run in the same order, so every
 * result is BIT-IDENTICAL to the float entry points on float frames of the same values.  Arguments as in
 * acktr_conv2d_fwd_f32 / acktr_conv2d_wgrad_deferred_f32 (pending nullable) / acktr_kfac_cov_a_conv_f32 with C = 4,
 * H = W = 84, OC = 32, KH = KW = 8, S = 4 fixed; x 4-byte aligned.  The forward needs the weight-image workspace
 * (acktr_conv2d_fwd_workspace_bytes), the weight gradient acktr_conv1_wgrad_u8_workspace_bytes(B), the factor
 * acktr_kfac_cov_workspace_bytes(256, B * 400). */
int acktr_conv1_fwd_u8(const uint8_t* x, const int64_t* gather_idx, const float* w, const float* bias, float* y, int B,
                       float in_scale, int relu, void* workspace, long workspace_bytes, int prepacked, void* stream);
/* (with the gate bits of acktr_conv2d_fwd_gate_f32) */
int acktr_conv1_fwd_u8_gate(const uint8_t* x, const int64_t* gather_idx, const float* w, const float* bias, float* y,
                            int B, float in_scale, int relu, void* workspace, long workspace_bytes, int prepacked,
                            uint32_t* gate_mask, void* stream);
long acktr_conv1_wgrad_u8_workspace_bytes(int B);
int acktr_conv1_wgrad_u8(const uint8_t* x, const int64_t* gather_idx, const float* dy, float* dw, float* db, int B,
                         float in_scale, void* workspace, long workspace_bytes, acktr_slab_fin* pending, void* stream);
int acktr_kfac_cov_a_conv1_u8(const uint8_t* x, const int64_t* gather_idx, int B, float alpha, float* out,
                              void* workspace, long workspace_bytes, void* stream);

/* ---- optimisers on flat parameter buffers ------------------------------------------------- */

/* nn.utils.clip_grad_norm_ + optim.Adam.step -- algo/ppo.py:82-84 (max_grad_norm <= 0 disables clipping).
 * step_size = lr/(1-beta1^t), bias_correction2_sqrt = sqrt(1-beta2^t) are computed by the host in double
 * (torch/optim/adam.py:529-547).  grads are scaled in place by the clip coefficient, as the reference does.
 * workspace: acktr_optim_workspace_bytes(). */
long acktr_optim_workspace_bytes(void);
int acktr_clip_adam_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n, float max_grad_norm,
                        float step_size, float beta1, float beta2, float eps, float bias_correction2_sqrt,
                        float* total_norm_out, void* workspace, void* stream);
/* Same update with the step-dependent scalars taken from DEVICE memory so the launch sequence can be captured in a
 * hipGraph and replayed: hyper_dev = {lr, beta1, beta2, eps}; *step_dev is incremented first and used as Adam's t
 * (bias corrections evaluated in double on the device).  No counterpart in the reference (host loop there). */
int acktr_clip_adam_dev_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n, float max_grad_norm,
                            const float* hyper_dev, int* step_dev, void* workspace, void* stream);
/* acktr_clip_adam_dev_f32 with pending reductions carried in the norm launch: a head finalize
 * (acktr_head_step_deferred_f32) and / or a weight-gradient slab sum (acktr_conv2d_wgrad_deferred_f32); either may be
 * NULL (or, for `slab`, hold dst == NULL: nothing pending).  Their destinations must lie inside `grads`;
 * max_grad_norm > 0 required. */
int acktr_clip_adam_dev_head_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n,
                                 float max_grad_norm, const float* hyper_dev, int* step_dev,
                                 const acktr_head_fin* pending, const acktr_slab_fin* slab, void* workspace,
                                 void* stream);
/* acktr_clip_adam_dev_head_f32 that also keeps the stationary conv kernels' weight IMAGES (the re-ordered copies
 * acktr_conv2d_pack_weights_f32 writes) in step with the parameters: element i of params in [lo, hi) is additionally
 * stored to base[t1[i - lo]] and base[t2[i - lo]] (float offsets; -1 = no slot) -- its slot in the forward image and in
 * the input-gradient image of its layer.  The next forward then needs no pack launch (4 us of a 0.55 ms step).  map
 * nullable (then identical to acktr_clip_adam_dev_head_f32).  No counterpart in the reference. */
typedef struct acktr_pack_map {
    long lo, hi; const int* t1; const int* t2; float* base;
} acktr_pack_map;
int acktr_clip_adam_dev_pack_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n,
                                 float max_grad_norm, const float* hyper_dev, int* step_dev,
                                 const acktr_head_fin* pending, const acktr_slab_fin* slab, const acktr_pack_map* map,
                                 void* workspace, void* stream);

/* clip_grad_norm_ + optim.RMSprop.step -- algo/a2c_acktr.py:74-78 (torch/optim/rmsprop.py:316-339). */
int acktr_clip_rmsprop_f32(float* params, float* grads, float* square_avg, long n, float max_grad_norm, float lr,
                           float alpha, float eps, float* total_norm_out, void* workspace, void* stream);
/* grads = (update_src ? update_src : grads) * *grad_scale_dev (both nullable; p.grad <- v * nu of kfac.py:236-239,
 * nu kept on the device so no host sync is needed) followed by optim.SGD(momentum).step -- kfac.py:139-142,241 (torch/optim/sgd.py:358-380). */
int acktr_sgd_momentum_f32(float* params, float* grads, const float* update_src, float* momentum_buf, long n,
                           const float* grad_scale_dev, float lr, float momentum, int first_step, void* stream);

/* ---- KFAC (a2c_ppo_acktr/algo/kfac.py) ---------------------------------------------------- */

/* compute_cov_a for a Conv2d input -- kfac.py:16-26,29-46.  out[Kc,Kc] = alpha * sum_{b,oy,ox} p p^T over the
 * im2col patches p (index order (c,ky,kx)) of x[B,C,H,W]; the patch matrix is never materialised.
 * The reference's scaling is alpha = in_scale^2 / (B_global * (OH*OW)^2).  Same geometries as conv2d_fwd.
 * workspace: acktr_kfac_cov_workspace_bytes(Kc, B*OH*OW) for the split-K slabs. */
long acktr_kfac_cov_workspace_bytes(int D, long reduction_len);
int acktr_kfac_cov_a_conv_f32(const float* x, const int64_t* gather_idx, int B, int C, int H, int W, int KH, int KW,
                              int S, float alpha, float* out, void* workspace, long workspace_bytes, void* stream);
/* compute_cov_g for a Conv2d grad_output -- kfac.py:49-64.  out[OC,OC] = alpha * sum_{b,sp} g g^T for
 * dy[B,OC,SP]; reference scaling alpha = SP * B_global.  SP in {400, 81, 49}. */
int acktr_kfac_cov_g_conv_f32(const float* dy, int B, int OC, int SP, float alpha, float* out, void* workspace,
                              long workspace_bytes, void* stream);
/* out[D,D] = alpha * a^T a for row-major a[B,D] with leading dimension lda: Linear inputs (alpha = 1/B_global, kfac.py:46) and
 * Linear / AddBias grad_outputs (alpha = B_global, kfac.py:63-64). */
int acktr_kfac_cov_dense_f32(const float* a, long lda, int B, int D, float alpha, float* out, void* workspace,
                             long workspace_bytes, void* stream);
/* compute_cov_a with fast_cnn=True, kfac.py:33-36: out[B, C*KH*KW] = scale * the mean over the window positions of the
 * im2col patches of x[B (through gather_idx, may be NULL), C, H, W] -- one mean patch per image; the factor is then
 * acktr_kfac_cov_dense_f32 of it.  H*W <= 8192, KH*KW <= 256. */
int acktr_kfac_patch_mean_f32(const float* x, const int64_t* gather_idx, int B, int C, int H, int W, int KH, int KW, int S,
                              float scale, float* out, void* stream);
/* out[row] = sum_sp g[row, sp] -- the AddBias branch of compute_cov_g, kfac.py:59-61 (and the Conv2d branch with
 * fast_cnn=True, kfac.py:53-55). */
int acktr_spatial_sum_f32(const float* g, int rows, int SP, float* out, void* stream);
/* update_running_stat -- kfac.py:67-71: m = ((m * c1) + x) * c2, three roundings, with c1 = (float)(d/(1-d)) and
 * c2 = (float)(1-d) evaluated by the host in double like the python scalars; init != 0 first sets m = x
 * (kfac.py:156-157,174-175). */
int acktr_kfac_running_stat_f32(float* m, const float* x, long n, float c1, float c2, int init, void* stream);

/* torch.symeig(m, eigenvectors=True) + `d *= (d > thresh)` -- kfac.py:208-214.  Eigenvalues ascending;
 * eigenvectors are returned as the ROWS of qt (qt = Q^T).
 *  - batched Jacobi (hand-written, LDS resident): `descs_dev` is a device array of `count` structs
 *    {const float* a; float* d; float* qt; int n;} (natural C layout, 32 bytes), every n <= max_n <= 128;
 *  - rocSOLVER syevd for larger factors: in place on a (n x n), e_work[n] and info_dev[1] are scratch. */
int acktr_jacobi_max_n(void);
int acktr_sym_eig_jacobi_batched_f32(const void* descs_dev, int count, int max_n, float thresh, void* stream);
int acktr_sym_eig_rocsolver_f32(float* a_inout_qt, int n, float* d, float* e_work, int* info_dev, float thresh,
                                void* stream);
/* The same for `count` (<= 64) factors at once (`jobs` is a HOST array): the solves -- latency-bound chains of small
 * kernels -- run concurrently on internal side streams that fork from `stream`; the call returns after the side
 * streams have drained (host join), so work enqueued on `stream` afterwards sees the results.  e_work: sum of all n
 * floats, info_dev: count ints.  One device per process. */
typedef struct acktr_eig_job { float* a_inout_qt; float* d; int n; int reserved; } acktr_eig_job;
int acktr_sym_eig_rocsolver_multi_f32(const acktr_eig_job* jobs, int count, float* e_work, int* info_dev, float thresh,
                                      void* stream);

/* The same decomposition (kfac.py:208-214) for 32 < n <= acktr_jacobi_block_max_n() (2048) by a hand-written BLOCKED
 * one-sided Jacobi (csrc/jacobi_block.hip): 32-column blocks meet pairwise in round-robin rounds; per round one Gram
 * launch (fp32 MFMA), one 64 x 64 two-sided Jacobi launch (LDS) and one apply launch (MFMA) serve ALL `count` (<= 64)
 * matrices of the call.  `jobs` is a HOST array; a (n x n, symmetric, upper triangle authoritative like symeig) is only
 * read; d[n] ascending and thresholded; qt[n x n] rows = eigenvectors.  Relative rotation criterion: small eigenpairs
 * keep column-scaled accuracy (a tridiagonalisation-based solver's error is eps*||A|| absolute).  The call synchronises
 * `stream` once per sweep (convergence counters) and returns with the final launches enqueued on `stream`.
 * workspace: acktr_sym_eig_jacobi_block_workspace_bytes(ns, count), 256-byte aligned.  sweeps_out (host, nullable)
 * receives the sweeps each matrix took (negative: not converged within max_sweeps; max_sweeps <= 0 -> 24). */
typedef struct acktr_jacobi_job { const float* a; float* d; float* qt; int n; } acktr_jacobi_job;
int acktr_jacobi_block_max_n(void);
long acktr_sym_eig_jacobi_block_workspace_bytes(const int* ns, int count);
int acktr_sym_eig_jacobi_block_f32(const acktr_jacobi_job* jobs, int count, float thresh, void* workspace,
                                   long workspace_bytes, int max_sweeps, int* sweeps_out, void* stream);

/* v = Q_g [ (Q_g^T G Q_a) / (d_g d_a^T + damping) ] Q_a^T -- kfac.py:216-227, G = grad viewed [OUT, IN];
 * tmp1, tmp2: OUT*IN floats each. */
int acktr_kfac_precondition_f32(const float* grad, const float* qt_g, const float* d_g, const float* qt_a,
                                const float* d_a, int OUT, int IN, float damping, float* tmp1, float* tmp2,
                                float* v_out, void* stream);
/* reference kfac.py:216-227 for ALL layers of a step at once: the same four products as acktr_kfac_precondition_f32, with
 * stage s of every layer in one grouped launch (4 launches per 16 layers instead of 4 per layer).  Each job needs its OWN tmp1 / tmp2 of OUT*IN floats (the layers run concurrently); v_out may not alias them. */
typedef struct acktr_precond_job {
    const float* grad; const float* qt_g; const float* d_g; const float* qt_a; const float* d_a;
    float* tmp1; float* tmp2; float* v_out;
    int OUT; int IN;
    /* optional (both or neither, for every job of a call): the eigenvector matrices themselves, q_g [OUT,OUT] = qt_g^T and
     * q_a [IN,IN] = qt_a^T (acktr_transpose_f32 after each eigendecomposition).  With them every operand of the four
     * stages is contiguous along its output index and the stages run on stationary accumulators (kfac.hip). */
    const float* q_g; const float* q_a;
} acktr_precond_job;
int acktr_kfac_precondition_batched_f32(const acktr_precond_job* jobs, int count, float damping, void* stream);

/* out[0] = vg_sum = lr^2 * <v, g>, out[1] = nu = min(1, sqrt(kl_clip / vg_sum)) -- kfac.py:229-234, on the
 * device (the reference syncs to the host here).  workspace: acktr_optim_workspace_bytes(). */
int acktr_kfac_kl_clip_f32(const float* v, const float* g, long n, float lr, float kl_clip, float* out_vgsum_nu,
                           void* workspace, void* stream);

/* ---- GAIL discriminator: the non-GEMM pieces of a2c_ppo_acktr/algo/gail.py ---------------- */

/* gail.py:36-41,67-76: x_out[3B, Ds+Da] = rows [policy (s|a)] [expert (s|a)] [alpha*expert + (1-alpha)*policy]; alpha[B]. */
int acktr_gail_mix_concat_f32(const float* policy_state, const float* policy_action, const float* expert_state,
                              const float* expert_action, const float* alpha, int B, int Ds, int Da, float* x_out,
                              void* stream);
/* the same with the policy rows read through a minibatch index (storage.py:127-140 fused in): policy row b =
 * state_rows[idx[b], :] | action_rows[idx[b], :] (idx NULL: row b); the caller guarantees 0 <= idx[b] < rows of the two
 * arrays.  mix_copies 1 or 2: with 2, x_out has 4B rows and rows [3B, 4B) repeat the mixup rows (the stacked weight
 * gradients below pair them with the penalty's d a while rows [2B, 3B) are reused for the double backward's operands). */
int acktr_gail_mix_concat_idx_f32(const float* state_rows, const float* action_rows, const int64_t* idx,
                                  const float* expert_state, const float* expert_action, const float* alpha, int B, int Ds,
                                  int Da, int mix_copies, float* x_out, void* stream);
/* gail.py:78-85 on the trunk outputs d[3B] of those rows: loss_out[0] = BCE(expert, 1) + BCE(policy, 0) (means);
 * seed[3B] = d loss / d d for the policy / expert rows and 1 for the mixup rows (the `ones` of gail.py:45);
 * seed_bce[3B] = the same with 0 on the mixup rows.  stacked != 0: seed_bce has 4B rows, 1 on [2B, 3B), 0 on [3B, 4B). */
int acktr_gail_bce_seed_f32(const float* d, int B, int stacked, float* seed, float* seed_bce, float* loss_out, void* stream);
/* gail.py:54: loss_out[1] = lambda * mean((||g_b|| - 1)^2) for g[B, D] = d D / d x at the mixup rows, and
 * r[B, D] = d loss_out[1] / d g (the seed of the double backward).  row_scratch: B floats, needed (and used) only
 * above B = 4096. */
int acktr_gail_pen_seed_f32(const float* g, int B, int D, float lambda_, float* r, float* row_scratch, float* loss_out,
                            void* stream);
/* gail.py:93 (loss.backward()) for the three Linear layers at once, two launches: dw_l[out_l, in_l] = A_l^T X_l summed over
 * all `rows` rows of the row-stacked operands (a1, a2: [rows, H]; a3: [rows]; x: [rows, D]; h1, h2: [rows, H]) and
 * db_l = the column sums of A_l over the rows outside [skip_lo, skip_hi).  The host stacks, per layer, the ordinary
 * backward's rows, the double backward's outer-product pair (skipped by the bias sums) and the penalty's rows. */
int acktr_gail_param_grads_f32(const float* a1, const float* x, const float* a2, const float* h1, const float* a3,
                               const float* h2, int rows, int D, int H, int skip_lo, int skip_hi, float* dw1, float* db1,
                               float* dw2, float* db2, float* dw3, float* db3, void* stream);
/* element-wise links of the (double) backward through tanh layers, h = tanh(.) stored, n elements, `cols` columns:
 * mode 0: out0 = (1-h^2) in0;  1: out0 = (1-h^2) in0, out1 = in1*in0;  2: out0 = (1-h^2) in0,
 * out1 = (1-h^2)(-2 h w[col] in0);  3: out0 = (1-h^2)(in0 - 2 h in1). */
int acktr_gail_tanh_link_f32(int mode, const float* h, const float* in0, const float* in1, const float* w, long n,
                             int cols, float* out0, float* out1, void* stream);
/* gail.py:101-102: reward = log(sigmoid(d)) - log(1 - sigmoid(d)). */
int acktr_gail_reward_f32(const float* d, long n, float* reward, void* stream);
/* gail.py:107-108 for S consecutive steps of N environments (reward, masks: [S, N]): ret = ret*mask*gamma + reward per
 * step (ret_io[N] carried in/out, every step's row in ret_steps[S, N]) and the per-step batch moments
 * moments[S][2] = {mean, population variance} in fp64 that RunningMeanStd.update merges. */
int acktr_gail_returns_moments_f32(const float* reward, const float* masks, int S, int N, float gamma, float* ret_io,
                                   float* ret_steps, double* moments, void* stream);
/* gail.py:108-110 for S consecutive steps: RunningMeanStd.update_from_moments (stable_baselines3's published parallel
 * mean / variance merge, fp64) applied to moments[s] = {mean, population variance} of batch_count samples each, in step
 * order, on state_inout = {mean, var, count}; scale_out[s] = sqrt(var after step s + 1e-8).  One lane: the merge is a
 * scalar recurrence (the reference does it on the host, one call and one host sync per rollout step). */
int acktr_gail_rms_merge_f64(const double* moments, int S, double batch_count, double* state_inout, float* scale_out,
                             void* stream);
/* gail.py:110: reward[s, :] /= scale[s]. */
int acktr_gail_scale_rows_f32(float* reward, const float* scale, int S, int N, void* stream);

/* ---- one-launch minibatch step of the MLP policy (model.py:198-229 MLPBase, distributions.py:76-95 DiagGaussian, the
 * loss blocks of algo/ppo.py:57-81 and algo/a2c_acktr.py:38-51,70-72): forward of both tanh towers and heads, loss with
 * its backward seeds, every parameter gradient.  One workgroup, everything resident in LDS.  Supported: B <= 64 rows,
 * hidden 64, IN <= 64, A <= 8, algo 0 (PPO) or 1 (A2C) -- acktr_mlp_fused_step_supported() tells; other shapes use the
 * per-layer entry points.  Weights w*: [out, in] row major as nn.Linear stores them; g*: where the gradients go
 * (overwritten); loss operands as in acktr_loss_args (read through gather_idx when given); loss_acc[3] += means. */
typedef struct acktr_mlp_step_args {
    const float *wc0, *bc0, *wc2, *bc2, *wa0, *ba0, *wa2, *ba2, *wv, *bv, *wm, *bm, *logstd;
    float *gwc0, *gbc0, *gwc2, *gbc2, *gwa0, *gba0, *gwa2, *gba2, *gwv, *gbv, *gwm, *gbm, *glogstd;
    const float* obs;           /* [rows, IN] */
    const int64_t* gather_idx;  /* nullable, [B] */
    int B, IN, H, A;
    int algo;                   /* 0 PPO, 1 A2C */
    const float *actions, *old_logp, *old_values, *returns, *adv;
    float clip, value_coef, entropy_coef, inv_batch;
    int use_clipped_value_loss;
    float* loss_acc;            /* nullable */
    /* optional optimiser tail (all null / 0: off): clip_grad_norm_(opt_max_norm) + Adam over the flat buffers the g*
     * pointers above point into, in the same launch, with the arithmetic and device-side {lr, beta1, beta2, eps} / step
     * counter of acktr_clip_adam_dev_f32 (the counter is incremented here).  opt_max_norm <= 0: no clipping. */
    float *opt_params, *opt_grads, *opt_exp_avg, *opt_exp_avg_sq;
    long opt_n;
    float opt_max_norm;
    const float* opt_hyper_dev;
    int* opt_step_dev;
    /* categorical != 0: the action head is distributions.py:59-73 Categorical(logits = h W_m^T + b_m) instead of
     * DiagGaussian -- `actions_i64` [rows] replaces `actions`, `logstd` / `glogstd` are not used.  (Last, so that a
     * zero-initialised tail keeps the DiagGaussian meaning of the fields above.) */
    int categorical;
    const int64_t* actions_i64;
} acktr_mlp_step_args;
int acktr_mlp_fused_step_supported(int B, int IN, int H, int A);
int acktr_mlp_fused_step_f32(const acktr_mlp_step_args* args, void* stream);

/* ---- recurrent bases: the element-wise links of nn.GRU as driven by NNBase._forward_gru (model.py:89-95,111-166).
 * Gate order r, z, n; rows are environments, [N, H] per step; the per-step products gi = x W_ih^T + b_ih and
 * gh = (h*mask) W_hh^T + b_hh come from acktr_linear_fwd_f32.
 * mask_rows: out[n, :] = h[n, :] * m[n]                                   (hxs * masks, model.py:113,155)
 * gates_fwd: r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r*gh_n), h_out = n + z*(hm - n);
 *            gi_rzn [N,3H] holds gi on entry and (r, z, n) on return; hm_next = h_out * m_next[n] when both given.
 * gates_bwd: dh = dout + (last ? 0 : carry * m_next[n]); rzn_dgi (r,z,n on entry) <- d gi; gh_dgh (gh on entry) <-
 *            d gh; carry <- dh * z, onto which the caller accumulates d gh W_hh (acktr_linear_dgrad_f32, accumulate). */
int acktr_gru_mask_rows_f32(const float* h, const float* m, float* out, int N, int H, void* stream);
int acktr_gru_gates_fwd_f32(float* gi_rzn, const float* gh, const float* hm, const float* m_next, float* h_out,
                            float* hm_next, int N, int H, void* stream);
int acktr_gru_gates_bwd_f32(const float* dout, const float* m_next, float* rzn_dgi, float* gh_dgh, const float* hm,
                            float* carry, int last, int N, int H, void* stream);
/* One launch per time step for few rows (acktr_gru_step_supported: 1 <= N <= 64, H % 4 == 0), model.py:111-166 and its
 * autograd backward -- the per-step product and the gate arithmetic of the three entries above in one kernel:
 * step_fwd: gh[N,3H] = hm W_hh^T + b_hh (kept for the backward), then gates_fwd's arithmetic on gi_rzn / h_out / hm_next.
 * step_bwd: step t of BPTT.  carry[N,H] holds dh*z of step t+1 on entry and of step t on return; dgh_next[N,3H] is step
 *           t+1's d gh (as left by step_bwd), w_hh_t [H,3H] the TRANSPOSED W_hh (acktr_transpose_f32, once per pass):
 *           dh = dout + (last ? 0 : (carry + dgh_next w_hh_t^T) * m_next[n]), then gates_bwd's arithmetic.
 * transpose: dst[cols, rows] = src[rows, cols]^T. */
int acktr_gru_step_supported(int N, int H);
int acktr_gru_step_fwd_f32(float* gi_rzn, float* gh, const float* hm, const float* w_hh, const float* b_hh,
                           const float* m_next, float* h_out, float* hm_next, int N, int H, void* stream);
int acktr_gru_step_bwd_f32(const float* dout, const float* m_next, float* rzn_dgi, float* gh_dgh, const float* hm,
                           const float* dgh_next, const float* w_hh_t, float* carry, int last, int N, int H, void* stream);
int acktr_transpose_f32(const float* src, int rows, int cols, float* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACKTR_HIP_H */
