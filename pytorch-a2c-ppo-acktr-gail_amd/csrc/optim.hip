// Flat-buffer optimiser kernels: global-norm clip fused with Adam / RMSprop, SGD-momentum for KFAC.
// All parameters (and grads, and optimiser state) of a Policy live in ONE contiguous fp32 buffer each,
// so the reference's per-tensor loops (clip_grad_norm_ + optimizer.step, ~100 tiny launches) are two
// launches here: (1) block partials of sum g^2 in fp64, (2) fixed-order re-reduction + the update.
//
// Reference arithmetic restated:
//   torch/nn/utils/clip_grad.py:102-108,165-169  coef = min(1, max_norm / (||g||_2 + 1e-6))
//   torch/optim/adam.py:457,476,529-547          single-tensor Adam
//   torch/optim/rmsprop.py:316-339               single-tensor RMSprop (not centred, no momentum)
//   torch/optim/sgd.py:358-380                   SGD with momentum (first step: buf = grad)
#include "optim_terms.hpp"
#include "head_finalize.hpp"
#include "../../include/acktr_hip.h"

namespace acktr {

constexpr int OPT_THREADS = 256;
constexpr int OPT_MAX_PARTIALS = 1024;

// `bump` (optional): the device-side Adam step counter, incremented here by one lane so that the update kernel that
// follows reads the new value without a launch of its own (a dependent launch costs ~4.5 us inside a hipGraph).
__global__ __launch_bounds__(OPT_THREADS) void sumsq_partial_kernel(const float* g, long n, double* partial, int* bump) {
    __shared__ double sm[16];
    if (bump && blockIdx.x == 0 && threadIdx.x == 0) *bump = *bump + 1;
    double s = 0.0;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * OPT_THREADS) {
        const double x = (double)g[i];
        s += x * x;
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// The same partials with pending reductions riding in the launch (acktr_head_step_deferred_f32,
// acktr_conv2d_wgrad_deferred_f32): workgroups [0, nfin) sum the head step's slabs into dwh / dbh, [nfin, nfin + nred) a
// weight gradient's slabs into its dw | db -- all inside g -- and report the squares of what they wrote; the others walk
// g and leave those ranges out.  One dependent launch fewer per pending reduction and minibatch step.
__global__ __launch_bounds__(OPT_THREADS) void sumsq_fin_kernel(const float* g, long n, double* partial, int* bump,
                                                                const acktr_head_fin f, int nfin, long w0, long b0,
                                                                const acktr_slab_fin r, int nred, long r0) {
    __shared__ double sm[16];
    double s = 0.0;
    const int npre = nfin + nred;
    if ((int)blockIdx.x < nfin) {
        s = head_finalize_block(f, blockIdx.x);
    } else if ((int)blockIdx.x < npre) {
        s = slab_finalize_block(r, blockIdx.x - nfin);
    } else {
        const int blk = blockIdx.x - npre, nblk = gridDim.x - npre;
        if (bump && blk == 0 && threadIdx.x == 0) *bump = *bump + 1;
        const long wn = nfin ? f.nw : 0, bn = nfin ? f.OUT : 0, rn = nred ? 4 * r.n4 : 0;
        for (long i = (long)blk * OPT_THREADS + threadIdx.x; i < n; i += (long)nblk * OPT_THREADS) {
            if ((i >= w0 && i < w0 + wn) || (i >= b0 && i < b0 + bn) || (i >= r0 && i < r0 + rn)) continue;
            const double x = (double)g[i];
            s += x * x;
        }
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// clip coefficient from the partials (every block recomputes it in the same fixed order)
__device__ __forceinline__ float clip_coef_from(const double* partial, int nparts, float max_norm, float* sm2,
                                                float* norm_out) {
    if (threadIdx.x < 64) {
        // nparts <= OPT_MAX_PARTIALS = 16 * 64: a fixed, fully unrolled trip count keeps the lane's (up to) 16 loads in
        // flight together -- with a run-time bound they were 14 dependent L2 round trips in front of every workgroup
        double v[OPT_MAX_PARTIALS / 64];
#pragma unroll
        for (int j = 0; j < OPT_MAX_PARTIALS / 64; ++j) {
            const int i = threadIdx.x + 64 * j;
            v[j] = partial[i < nparts ? i : 0];
        }
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < OPT_MAX_PARTIALS / 64; ++j) s += (threadIdx.x + 64 * j < nparts) ? v[j] : 0.0;
        s = wave_sum(s);
        if (threadIdx.x == 0) {
            const float total = (float)sqrt(s);
            float c = max_norm / (total + 1e-6f);
            sm2[0] = c > 1.0f ? 1.0f : c;
            if (norm_out && blockIdx.x == 0) *norm_out = total;
        }
    }
    __syncthreads();
    return sm2[0];
}

// mode 0: Adam, 1: RMSprop.  use_clip: multiply the gradient by the clip coefficient first (and write it back,
// as clip_grad_norm_ does in place).
// hyper_dev (optional): {lr, beta1_or_alpha, beta2, eps} in device memory and step_dev the (already incremented) Adam
// step counter -- lets a captured hipGraph replay with changing bias corrections / learning rates.
template <int MODE>
__global__ __launch_bounds__(OPT_THREADS) void clip_update_kernel(float* p, float* g, float* s1, float* s2, long n,
                                                                  const double* partial, int nparts, int use_clip,
                                                                  float max_norm, float lr_or_step, float beta1,
                                                                  float beta2, float eps, float bc2_sqrt,
                                                                  float* norm_out, const float* hyper_dev,
                                                                  const int* step_dev, const acktr_pack_map map) {
    __shared__ float sm2[2];
    __shared__ float sm_adam[2];
    float coef = 1.0f;
    // The first (for this workload: only) float4 group of the thread is requested BEFORE the serial preamble below
    // (two double pow()s on one lane, the fixed-order sum of the norm partials, two barriers), which otherwise sits
    // in front of every load of the kernel.
    typedef float v4 __attribute__((ext_vector_type(4)));
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)s1 | (uintptr_t)s2) & 15) == 0);
    const long n4 = vec ? n / 4 : 0;
    const long i_first = (long)blockIdx.x * OPT_THREADS + threadIdx.x;
    v4 pv0 = {0.f, 0.f, 0.f, 0.f}, gv0 = pv0, av0 = pv0, bv0 = pv0;
    if (i_first < n4) {
        pv0 = reinterpret_cast<v4*>(p)[i_first]; gv0 = reinterpret_cast<v4*>(g)[i_first]; av0 = reinterpret_cast<v4*>(s1)[i_first];
        if (MODE == 0) bv0 = reinterpret_cast<v4*>(s2)[i_first];
    }
    if (hyper_dev) {
        const float lr = hyper_dev[0];
        eps = hyper_dev[3];
        if (MODE == 0) {
            beta1 = hyper_dev[1]; beta2 = hyper_dev[2];
            // torch/optim/adam.py:529-535 evaluates these in host doubles; one lane per pow does (a double pow per lane
            // cost more than the whole update), the rest read them from LDS.  The two lanes sit in waves 1 and 2, so the
            // pows run beside each other and beside wave 0's sum of the norm partials, behind ONE barrier.
            if (threadIdx.x == 64) {
                const double bc1 = 1.0 - pow((double)beta1, (double)*step_dev);
                sm_adam[0] = (float)((double)lr / bc1);
            }
            if (threadIdx.x == 128) {
                const double bc2 = 1.0 - pow((double)beta2, (double)*step_dev);
                sm_adam[1] = (float)sqrt(bc2);
            }
        } else {
            beta2 = hyper_dev[1];
            lr_or_step = lr;
        }
    }
    if (use_clip) coef = clip_coef_from(partial, nparts, max_norm, sm2, norm_out);     // ends on a barrier
    else __syncthreads();
    if (hyper_dev && MODE == 0) {
        lr_or_step = sm_adam[0];
        bc2_sqrt = sm_adam[1];
    }
    auto upd = [&](float& pi, float& gi, float& a1, float& a2) {
        if (use_clip) gi = gi * coef;
        if (MODE == 0) adam_elem(pi, gi, a1, a2, beta1, beta2, eps, bc2_sqrt, lr_or_step);
        else rmsprop_elem(pi, gi, a1, beta2, eps, lr_or_step);
    };
    for (long i = i_first; i < n4; i += (long)gridDim.x * OPT_THREADS) {
        v4 pv = pv0, gv = gv0, av = av0, bv = MODE == 0 ? bv0 : av0;
        if (i != i_first) {
            pv = reinterpret_cast<v4*>(p)[i]; gv = reinterpret_cast<v4*>(g)[i]; av = reinterpret_cast<v4*>(s1)[i];
            bv = av;
            if (MODE == 0) bv = reinterpret_cast<v4*>(s2)[i];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { float pj = pv[j], gj = gv[j], aj = av[j], bj = bv[j]; upd(pj, gj, aj, bj); pv[j] = pj; gv[j] = gj; av[j] = aj; bv[j] = bj; }
        reinterpret_cast<v4*>(p)[i] = pv;
        if (use_clip) reinterpret_cast<v4*>(g)[i] = gv;
        reinterpret_cast<v4*>(s1)[i] = av;
        if (MODE == 0) reinterpret_cast<v4*>(s2)[i] = bv;
        // the new values of the conv weights also go to their slots in the stationary kernels' weight images (forward
        // and input-gradient layouts), so the next step starts without a re-ordering launch
        if (map.t1 && 4 * i + 3 >= map.lo && 4 * i < map.hi) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long e = 4 * i + j - map.lo;
                if (e >= 0 && 4 * i + j < map.hi) {
                    const int d1 = map.t1[e], d2 = map.t2[e];
                    if (d1 >= 0) map.base[d1] = pv[j];
                    if (d2 >= 0) map.base[d2] = pv[j];
                }
            }
        }
    }
    for (long i = 4 * n4 + (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * OPT_THREADS) {
        float pi = p[i], gi = g[i], a1 = s1[i], a2 = MODE == 0 ? s2[i] : 0.f;
        upd(pi, gi, a1, a2);
        p[i] = pi;
        if (use_clip) g[i] = gi;
        s1[i] = a1;
        if (MODE == 0) s2[i] = a2;
        if (map.t1 && i >= map.lo && i < map.hi) {
            const int d1 = map.t1[i - map.lo], d2 = map.t2[i - map.lo];
            if (d1 >= 0) map.base[d1] = pi;
            if (d2 >= 0) map.base[d2] = pi;
        }
    }
}

// grad *= scale (device scalar, optional) ; SGD momentum: buf = first ? g : mom*buf + g ; p -= lr*buf
__global__ __launch_bounds__(OPT_THREADS) void sgd_momentum_kernel(float* p, float* g, const float* src, float* buf,
                                                                   long n, const float* grad_scale, float lr,
                                                                   float momentum, int first) {
    const float sc = grad_scale ? *grad_scale : 1.0f;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * OPT_THREADS) {
        float gi = src ? src[i] : g[i];
        if (grad_scale || src) { gi = gi * sc; g[i] = gi; }
        float b;
        if (first) b = gi;
        else b = buf[i] * momentum + gi;
        buf[i] = b;
        p[i] = p[i] - lr * b;
    }
}

__global__ void bump_step_kernel(int* step) { *step = *step + 1; }

static int grid_for(long n) {
    int b = cdiv(n, OPT_THREADS * 4);
    if (b > OPT_MAX_PARTIALS) b = OPT_MAX_PARTIALS;
    if (b < 1) b = 1;
    return b;
}
}  // namespace acktr

using namespace acktr;

extern "C" long acktr_optim_workspace_bytes(void) { return (long)OPT_MAX_PARTIALS * (long)sizeof(double); }

extern "C" int acktr_clip_adam_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n,
                                   float max_grad_norm, float step_size, float beta1, float beta2, float eps,
                                   float bias_correction2_sqrt, float* total_norm_out, void* workspace, void* stream) {
    ACKTR_ARG(params && grads && exp_avg && exp_avg_sq && workspace, "clip_adam: null pointer");
    ACKTR_ARG(n > 0, "clip_adam: n must be positive");
    hipStream_t s = (hipStream_t)stream;
    const int use_clip = max_grad_norm > 0.f;
    const int nb = grid_for(n);
    if (use_clip) {
        hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(OPT_THREADS), 0, s, grads, n, (double*)workspace, (int*)nullptr);
        ACKTR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((clip_update_kernel<0>), dim3(nb), dim3(OPT_THREADS), 0, s, params, grads, exp_avg, exp_avg_sq, n,
                       (const double*)workspace, nb, use_clip, max_grad_norm, step_size, beta1, beta2, eps,
                       bias_correction2_sqrt, total_norm_out, (const float*)nullptr, (const int*)nullptr, acktr_pack_map{});
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_clip_adam_dev_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n,
                                       float max_grad_norm, const float* hyper_dev, int* step_dev, void* workspace,
                                       void* stream) {
    ACKTR_ARG(params && grads && exp_avg && exp_avg_sq && workspace && hyper_dev && step_dev, "clip_adam_dev: null pointer");
    ACKTR_ARG(n > 0, "clip_adam_dev: n must be positive");
    hipStream_t s = (hipStream_t)stream;
    const int use_clip = max_grad_norm > 0.f;
    const int nb = grid_for(n);
    if (use_clip) {
        hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(OPT_THREADS), 0, s, grads, n, (double*)workspace, step_dev);
        ACKTR_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(1), 0, s, step_dev);
        ACKTR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((clip_update_kernel<0>), dim3(nb), dim3(OPT_THREADS), 0, s, params, grads, exp_avg, exp_avg_sq, n,
                       (const double*)workspace, nb, use_clip, max_grad_norm, 0.f, 0.f, 0.f, 0.f, 1.f, (float*)nullptr,
                       hyper_dev, (const int*)step_dev, acktr_pack_map{});
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_clip_adam_dev_head_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n,
                                            float max_grad_norm, const float* hyper_dev, int* step_dev,
                                            const acktr_head_fin* pending, const acktr_slab_fin* slab, void* workspace,
                                            void* stream) {
    return acktr_clip_adam_dev_pack_f32(params, grads, exp_avg, exp_avg_sq, n, max_grad_norm, hyper_dev, step_dev, pending,
                                        slab, nullptr, workspace, stream);
}

extern "C" int acktr_clip_adam_dev_pack_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n,
                                            float max_grad_norm, const float* hyper_dev, int* step_dev,
                                            const acktr_head_fin* pending, const acktr_slab_fin* slab,
                                            const acktr_pack_map* map, void* workspace, void* stream) {
    ACKTR_ARG(params && grads && exp_avg && exp_avg_sq && workspace && hyper_dev && step_dev, "clip_adam_dev_head: null pointer");
    acktr_pack_map pm = {};
    if (map && map->t1) {
        pm = *map;
        ACKTR_ARG(pm.t2 && pm.base && pm.lo >= 0 && pm.lo < pm.hi && pm.hi <= n, "clip_adam_dev_pack: bad weight-image map");
    }
    ACKTR_ARG(n > 0 && max_grad_norm > 0.f, "clip_adam_dev_head: needs n > 0 and a clip norm (the pending sums ride in the norm launch)");
    acktr_head_fin f = {};
    acktr_slab_fin r = {};
    int nfin = 0, nred = 0;
    if (pending) {
        f = *pending;
        ACKTR_ARG(f.slabs && f.gw && f.gb && f.nb > 0 && f.n4 > 0 && f.nw > 0 && f.OUT > 0 && f.OUT <= 8,
                  "clip_adam_dev_head: incomplete pending finalize");
        ACKTR_ARG(f.gw >= grads && f.gw + f.nw <= grads + n && f.gb >= grads && f.gb + f.OUT <= grads + n,
                  "clip_adam_dev_head: the head gradients must lie inside the flat gradient buffer");
        nfin = cdiv(f.n4, 32);
    }
    if (slab && slab->dst) {
        r = *slab;
        ACKTR_ARG(r.slabs && r.nslabs >= 16 && r.n4 > 0 && (r.stride & 3) == 0 && (((uintptr_t)r.dst | (uintptr_t)r.slabs) & 15) == 0,
                  "clip_adam_dev_head: incomplete pending slab sum");
        ACKTR_ARG(r.dst >= grads && r.dst + 4 * r.n4 <= grads + n,
                  "clip_adam_dev_head: the pending weight gradient must lie inside the flat gradient buffer");
        nred = (int)cdiv(r.n4, 16);
    }
    ACKTR_ARG(nfin + nred < OPT_MAX_PARTIALS / 2, "clip_adam_dev_head: pending reductions too large");
    hipStream_t s = (hipStream_t)stream;
    int nb = grid_for(n);
    if (nb + nfin + nred > OPT_MAX_PARTIALS) nb = OPT_MAX_PARTIALS - nfin - nred;
    const int total = nb + nfin + nred;
    hipLaunchKernelGGL(sumsq_fin_kernel, dim3(total), dim3(OPT_THREADS), 0, s, grads, n, (double*)workspace, step_dev, f, nfin,
                       nfin ? (long)(f.gw - grads) : 0L, nfin ? (long)(f.gb - grads) : 0L, r, nred,
                       nred ? (long)(r.dst - grads) : 0L);
    ACKTR_LAUNCH_CHECK();
    hipLaunchKernelGGL((clip_update_kernel<0>), dim3(grid_for(n)), dim3(OPT_THREADS), 0, s, params, grads, exp_avg, exp_avg_sq, n,
                       (const double*)workspace, total, 1, max_grad_norm, 0.f, 0.f, 0.f, 0.f, 1.f, (float*)nullptr,
                       hyper_dev, (const int*)step_dev, pm);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_clip_rmsprop_f32(float* params, float* grads, float* square_avg, long n, float max_grad_norm,
                                      float lr, float alpha, float eps, float* total_norm_out, void* workspace,
                                      void* stream) {
    ACKTR_ARG(params && grads && square_avg && workspace, "clip_rmsprop: null pointer");
    ACKTR_ARG(n > 0, "clip_rmsprop: n must be positive");
    hipStream_t s = (hipStream_t)stream;
    const int use_clip = max_grad_norm > 0.f;
    const int nb = grid_for(n);
    if (use_clip) {
        hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(OPT_THREADS), 0, s, grads, n, (double*)workspace, (int*)nullptr);
        ACKTR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((clip_update_kernel<1>), dim3(nb), dim3(OPT_THREADS), 0, s, params, grads, square_avg,
                       (float*)nullptr, n, (const double*)workspace, nb, use_clip, max_grad_norm, lr, 0.f, alpha, eps, 1.f,
                       total_norm_out, (const float*)nullptr, (const int*)nullptr, acktr_pack_map{});
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_sgd_momentum_f32(float* params, float* grads, const float* update_src, float* momentum_buf,
                                      long n, const float* grad_scale_dev, float lr, float momentum, int first_step,
                                      void* stream) {
    ACKTR_ARG(params && grads && momentum_buf, "sgd_momentum: null pointer");
    ACKTR_ARG(n > 0, "sgd_momentum: n must be positive");
    hipLaunchKernelGGL(sgd_momentum_kernel, dim3(grid_for(n)), dim3(OPT_THREADS), 0, (hipStream_t)stream, params, grads,
                       update_src, momentum_buf, n, grad_scale_dev, lr, momentum, first_step);
    ACKTR_LAUNCH_CHECK();
    return 0;
}
