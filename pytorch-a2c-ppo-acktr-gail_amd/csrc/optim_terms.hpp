// Single-element optimiser arithmetic shared by the flat-buffer kernels (optim.hip) and the fused MLP step
// (mlp_fused.hip): torch/optim/adam.py:529-547 and torch/optim/rmsprop.py:316-339 restated, same operation order.
#pragma once
#include "common.hpp"

namespace acktr {

// Adam: exp_avg.lerp_(g, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2); p.addcdiv_(exp_avg, sqrt(v)/sqrt(bc2)+eps, -lr/bc1)
__device__ __forceinline__ void adam_elem(float& pi, float gi, float& a1, float& a2, float beta1, float beta2, float eps,
                                          float bc2_sqrt, float step_size) {
    a1 = a1 + (1.0f - beta1) * (gi - a1);
    a2 = a2 * beta2 + (1.0f - beta2) * gi * gi;
    const float denom = sqrtf(a2) / bc2_sqrt + eps;
    pi = pi - step_size * (a1 / denom);
}
// RMSprop (not centred, no momentum): square_avg.mul_(alpha).addcmul_(g, g, 1-alpha); p.addcdiv_(g, sqrt(avg)+eps, -lr)
__device__ __forceinline__ void rmsprop_elem(float& pi, float gi, float& a1, float alpha, float eps, float lr) {
    a1 = a1 * alpha + (1.0f - alpha) * gi * gi;
    const float avg = sqrtf(a1) + eps;
    pi = pi - lr * (gi / avg);
}

}  // namespace acktr
