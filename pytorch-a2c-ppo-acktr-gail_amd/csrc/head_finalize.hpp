// Second half of the fused head step (head_fused.hip): the fixed-order sum of its workgroup slabs.  Shared by the stand-alone
// finalize launch and by the optimiser's gradient-norm launch (optim.hip), which can run it as extra workgroups.
#pragma once
#include "common.hpp"
#include "../../include/acktr_hip.h"

namespace acktr {

typedef float hfin4 __attribute__((ext_vector_type(4)));

// Eight lanes share one float4 column (slab s goes to lane group s % 8), then a three-step butterfly.  Element e of a slab:
// e < nw -> dW[e]; nw <= e < nw+8 -> db; nw+8.. -> the three loss sums.  `block` plays blockIdx.x of a 256-thread workgroup.
// Returns the sum of squares of the gradient values this thread wrote (dW and db; not the loss sums).
__device__ __forceinline__ double head_finalize_block(const acktr_head_fin& f, int block) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = (block * 4 + wave) * 8 + (lane & 7), sub = lane >> 3;
    hfin4 acc = hfin4{0.f, 0.f, 0.f, 0.f};
    if (i < f.n4) {
#pragma unroll 4
        for (int s = sub; s < f.nb; s += 8) acc += *reinterpret_cast<const hfin4*>(f.slabs + (long)s * f.stride + 4 * i);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = acc[j];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[j] = v;
    }
    double sq = 0.0;
    if (sub != 0 || i >= f.n4) return sq;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = 4 * i + j;
        if (e < f.nw) { f.gw[e] = acc[j]; sq += (double)acc[j] * (double)acc[j]; }
        else if (e < f.nw + 8) { if (f.gb && e - f.nw < f.OUT) { f.gb[e - f.nw] = acc[j]; sq += (double)acc[j] * (double)acc[j]; } }
        else if (e < f.nw + 11) { if (f.loss_acc) f.loss_acc[e - f.nw - 8] += acc[j] * f.inv_batch; }
    }
    return sq;
}

// slab_reduce_wave_kernel's arithmetic (linear.hip: sixteen lanes share a float4 column, lane group `sub` sums slabs sub,
// sub+16, ..., four-step butterfly) for a pending weight-gradient sum; `block` plays blockIdx.x of a 256-thread workgroup.
// Returns the sum of squares of what this thread wrote.
__device__ __forceinline__ double slab_finalize_block(const acktr_slab_fin& f, int block) {
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;      // (& 3: callers with more than 256 threads pass one
                                                                           //  virtual 256-thread block per four waves)
    const long i = ((long)block * 4 + wave) * 4 + (lane & 3);
    const int sub = lane >> 2;
    hfin4 acc = hfin4{0.f, 0.f, 0.f, 0.f};
    if (i < f.n4) {
#pragma unroll 8
        for (int k = sub; k < f.nslabs; k += 16) acc = acc + *reinterpret_cast<const hfin4*>(f.slabs + (long)k * f.stride + 4 * i);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = acc[j];
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[j] = v;
    }
    double sq = 0.0;
    if (sub != 0 || i >= f.n4) return sq;
    hfin4 res;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        res[j] = acc[j] * f.scale + 0.f;
        sq += (double)res[j] * (double)res[j];
    }
    *reinterpret_cast<hfin4*>(f.dst + 4 * i) = res;
    return sq;
}

}  // namespace acktr
