// Stationary-accumulator weight gradient of the 3x3 stride-1 layer (reference model.py:179, its autograd backward):
//   dW[oc, (c,ky,kx)] = scale * sum_{b,oy,ox} dY[b,oc,oy,ox] * X[b,c,oy+ky,ox+kx],   db[oc] = sum dY.
// The whole dW (OC x C*9 <= 32 x 576) lives in MFMA accumulators for the lifetime of a workgroup: wave w owns the
// 64 taps [64w, 64w+64) as two 32x32 tiles, and every operand goes from global memory STRAIGHT into the registers the
// MFMA reads -- no LDS, no barrier.  That works because with stride 1 four consecutive output positions of one row
// read four consecutive input floats for any tap:
//   A (dY):  lane (oc = l%32, h = l/32) holds dY[b, oc, oy, 4h .. 4h+3]            (one unaligned 16-byte load)
//   B (X):   lane (tap = l%32, h)       holds X[b, c, oy+ky, kx + 4h .. kx+4h+3]   (likewise)
// and MFMA step j pairs the j-th float of both: k = (position 4h+j).  A row has 7 positions; the eighth (h = 1, j = 3)
// is zero in A, so what B holds there does not count.  A workgroup handles a run of images and writes one
// partial dW (+db) slab; the shared fixed-order slab reduction finishes (deterministic).
// The tile-GEMM wgrad of this layer ran at 36 TFLOP/s: its K loop per workgroup was 12 tiles long between a pipeline
// fill and a 16 KB slab write.
#pragma once
#include "gemm_core.hpp"

namespace acktr {

constexpr int SA_OC = 32, SA_KK = 9, SA_H = 9, SA_OH = 7, SA_SP = 49, SA_HW = 81;

// One row of one image: the three 16-byte operand loads of a lane.  Lanes of the upper k half (h = 1) would need
// positions 4..7 of the row, of which 7 does not exist -- and whose window may lie one float past the tensor.  They
// load positions 3..6 instead and shift by one when the row is consumed, so nothing is ever read out of bounds.
// The loads are buffer loads (descriptor + 32-bit lane offset + scalar image offset): besides the cheaper addressing,
// the optimiser does not merge them across the loop back edge the way it folds a phi of two plain loads into one load
// at the top of the next iteration -- which silently undid the look-ahead below.
struct SaRow { f32x4 a, b0, b1; };
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 sa_load(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ void sa_issue(SaRow& r, __amdgpu_buffer_rsrc_t xr, __amdgpu_buffer_rsrc_t dyr, int xs, int dys, int aoff,
                                         const int (&tapoff)[2], int oy) {
    r.a = sa_load(dyr, aoff + oy * SA_OH * 4, dys);
    r.b0 = sa_load(xr, tapoff[0] + oy * SA_H * 4, xs);
    r.b1 = sa_load(xr, tapoff[1] + oy * SA_H * 4, xs);
}
__device__ __forceinline__ f32x4 sa_shift(const f32x4& v, int kh, float last) {
    f32x4 r;
    r[0] = kh ? v[1] : v[0]; r[1] = kh ? v[2] : v[1]; r[2] = kh ? v[3] : v[2]; r[3] = kh ? last : v[3];
    return r;
}
__device__ __forceinline__ void sa_consume(const SaRow& r, int kh, f32x16 (&acc)[2], float& bsum) {
    const f32x4 a = sa_shift(r.a, kh, 0.f);                   // position 7 of the row does not exist: A is zero there
    const f32x4 b0 = sa_shift(r.b0, kh, r.b0[3]), b1 = sa_shift(r.b1, kh, r.b1[3]);
    bsum += (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b0[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b1[j], acc[1], 0, 0, 0);
    }
}

// grid = number of slabs; block = (C*9/64) waves.  slab layout: [OC * C*9 | OC]
template <int C>
__global__ __launch_bounds__(C * SA_KK) void conv3_wgrad_sa_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ slabs, long slab_stride, int B,
                                                                  int imgs_per_wg, float scale) {
    static_assert((C * SA_KK) % 64 == 0, "whole waves of 64 taps");
    constexpr int KC = C * SA_KK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, kh = lane >> 5;
    int tapoff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 64 * wave + 32 * t + li;
        const int c = n / SA_KK, r9 = n % SA_KK;
        tapoff[t] = 4 * (c * SA_HW + (r9 / 3) * SA_H + (r9 % 3) + 3 * kh);   // bytes
    }
    const int aoff = 4 * (li * SA_SP + 3 * kh);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, B * C * SA_HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, B * SA_OC * SA_SP * 4, 0x00020000);
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const int b0 = blockIdx.x * imgs_per_wg;
    int b1 = b0 + imgs_per_wg;
    if (b1 > B) b1 = B;
    // a ring of seven row sets = exactly one image of look-ahead: as soon as row oy of image b has been multiplied, its
    // registers are refilled with row oy of image b+1 (unconditionally -- past the end the last image is simply
    // requested again -- so the compiler can count the loads in flight)
    SaRow ring[SA_OH];
    if (b0 < b1) {
#pragma unroll
        for (int oy = 0; oy < SA_OH; ++oy) {
            sa_issue(ring[oy], xr, dyr, b0 * C * SA_HW * 4, b0 * SA_OC * SA_SP * 4, aoff, tapoff, oy);
            __builtin_amdgcn_sched_barrier(0);                 // same issue order as in the loop, or the wait counts at the
                                                               // loop head are merged to the worse of the two
        }
    }
    for (int b = b0; b < b1; ++b) {
        const int bn = b + 1 < b1 ? b + 1 : b1 - 1;
        const int xs = bn * C * SA_HW * 4, dys = bn * SA_OC * SA_SP * 4;
#pragma unroll
        for (int oy = 0; oy < SA_OH; ++oy) {
            sa_consume(ring[oy], kh, acc, bsum);
            sa_issue(ring[oy], xr, dyr, xs, dys, aoff, tapoff, oy);
            __builtin_amdgcn_sched_barrier(0);                 // keep the refill right behind its row: the scheduler would
                                                               // otherwise collect all 21 loads at the end of the image
        }
    }
    float* slab = slabs + (long)blockIdx.x * slab_stride;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tap = 64 * wave + 32 * t + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oc = (r & 3) + 8 * (r >> 2) + 4 * kh;
            slab[(long)oc * KC + tap] = acc[t][r] * scale;
        }
    }
    if (wave == 0) {                                          // every wave saw all of dY: wave 0 reports the bias sums
        bsum += __shfl_xor(bsum, 32);
        if (kh == 0) slab[(long)SA_OC * KC + li] = bsum;
    }
}

}  // namespace acktr
