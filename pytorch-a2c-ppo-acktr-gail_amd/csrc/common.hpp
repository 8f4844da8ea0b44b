// Shared helpers for the acktr_hip C-ABI library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace acktr {

// Thread-local last-error text, surfaced through acktr_last_error().
void set_error(const char* fmt, ...);

#define ACKTR_ARG(cond, ...)                                   \
    do {                                                       \
        if (!(cond)) {                                         \
            ::acktr::set_error(__VA_ARGS__);                   \
            return -1;                                         \
        }                                                      \
    } while (0)

#define ACKTR_LAUNCH_CHECK()                                                   \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            ::acktr::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,  \
                               hipGetErrorString(e__));                        \
            return (int)e__;                                                   \
        }                                                                      \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// 64-lane wavefront reductions (no 32-lane assumptions anywhere).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x a multiple of 64 (<= 1024); result valid in thread 0.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* smem /* >= 16 entries */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    T r = 0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += smem[i];
    }
    return r;
}

}  // namespace acktr
