// Linear layers (forward / dgrad / wgrad / bias grad) on the fp32 MFMA tile GEMM, plus the shared
// fixed-order slab reduction.  Replaces the torch.nn.Linear call sites of the reference policy:
// model.py:180,185,208-216 and distributions.py:69,83 (forward) and their autograd backward.
#include <stdlib.h>
#include "gemm_core.hpp"
#include "../../include/acktr_hip.h"

namespace acktr {

// Fixed-order slab reduction in float4 units.  One pass when there are few slabs; otherwise two passes:
// pass 1 sums groups of slabs in parallel (grid.y = groups) into group partials, pass 2 sums the groups.
// Both orders are fixed, so split-K results are run-to-run deterministic.
// (Fusing the two passes into one launch -- last-arriving block sums the group partials behind an agent-scope
//  fence + atomic ticket -- was measured 2x SLOWER per wgrad: on the 8-XCD part the release fence writes back the
//  whole dirty L2, which at that moment holds megabytes of slabs.)
// Epilogue (final pass only): v = scale*sum (+ bias[i % ncols]) -> act -> (+= dst).
__global__ __launch_bounds__(256) void slab_reduce4_kernel(float* dst, const float* slabs, int first, int count,
                                                           long n4, long stride, float scale, int accumulate,
                                                           const float* bias, int ncols, int act, int group_stride,
                                                           long dst_group_stride) {
    const int g = blockIdx.y;
    const int s0 = first + g * group_stride;
    int s1 = s0 + group_stride;
    if (s1 > first + count) s1 = first + count;
    float* out = dst + (long)g * dst_group_stride;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 acc = zero4();
        int k = s0;
        for (; k + 4 <= s1; k += 4) {          // 4 independent loads in flight, summed in slab order
            const f32x4 a = *reinterpret_cast<const f32x4*>(slabs + (long)k * stride + 4 * i);
            const f32x4 b = *reinterpret_cast<const f32x4*>(slabs + (long)(k + 1) * stride + 4 * i);
            const f32x4 c = *reinterpret_cast<const f32x4*>(slabs + (long)(k + 2) * stride + 4 * i);
            const f32x4 d = *reinterpret_cast<const f32x4*>(slabs + (long)(k + 3) * stride + 4 * i);
            acc = acc + a; acc = acc + b; acc = acc + c; acc = acc + d;
        }
        for (; k < s1; ++k) acc = acc + *reinterpret_cast<const f32x4*>(slabs + (long)k * stride + 4 * i);
        // epilogue operands first, ONE 16-byte store last (a load behind a may-alias store cannot be hoisted over it)
        f32x4 bv = zero4(), prev = zero4();
        if (bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = bias[(4 * i + j) % ncols];
        }
        if (accumulate) prev = *reinterpret_cast<const f32x4*>(out + 4 * i);
        f32x4 res;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[j] * scale + bv[j];
            if (act == 1) v = v > 0.f ? v : 0.f;
            else if (act == 2) v = tanhf(v);
            res[j] = v + prev[j];
        }
        *reinterpret_cast<f32x4*>(out + 4 * i) = res;
    }
}

// Many slabs, few columns (the conv weight gradients: hundreds of slabs of 8-32 K floats): ONE pass in which sixteen
// lanes share a float4 column -- lane group `sub` sums slabs sub, sub+16, ... (all loads independent, so a lane has its
// whole share in flight at once) and a four-step butterfly adds the sixteen partials.  The order is fixed, so the
// result is run-to-run deterministic like the two-pass form it replaces (two launches of ~5 us each, both
// latency-bound).  Four adjacent columns per wave keep each request 64 contiguous bytes.
__global__ __launch_bounds__(256) void slab_reduce_wave_kernel(float* __restrict__ dst, const float* __restrict__ slabs,
                                                               int nslabs, long n4, long stride, float scale,
                                                               int accumulate, const float* __restrict__ bias, int ncols,
                                                               int act) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long i = ((long)blockIdx.x * 4 + wave) * 4 + (lane & 3);
    const int sub = lane >> 2;
    f32x4 acc = zero4();
    if (i < n4) {
#pragma unroll 8
        for (int k = sub; k < nslabs; k += 16) acc = acc + *reinterpret_cast<const f32x4*>(slabs + (long)k * stride + 4 * i);
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
    if (sub != 0 || i >= n4) return;
    f32x4 bv = zero4(), prev = zero4();
    if (bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = bias[(4 * i + j) % ncols];
    }
    if (accumulate) prev = *reinterpret_cast<const f32x4*>(dst + 4 * i);
    f32x4 res;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = acc[j] * scale + bv[j];
        if (act == 1) v = v > 0.f ? v : 0.f;
        else if (act == 2) v = tanhf(v);
        res[j] = v + prev[j];
    }
    *reinterpret_cast<f32x4*>(dst + 4 * i) = res;
}

int launch_slab_reduce_ex(float* dst, const float* slabs, int nslabs, long n, long stride, float scale, int accumulate,
                          const float* bias, int ncols, int act, float* group_ws, hipStream_t stream) {
    if (n <= 0) return 0;
    if ((n & 3) || (stride & 3) || (((uintptr_t)dst | (uintptr_t)slabs) & 15)) {
        set_error("slab_reduce: n, stride must be multiples of 4 and pointers 16-byte aligned");
        return -1;
    }
    const long n4 = n / 4;
    int bx = cdiv(n4, 256);
    if (bx > 1024) bx = 1024;
    if (nslabs >= 16 && n4 <= 16384) {
        hipLaunchKernelGGL(slab_reduce_wave_kernel, dim3(cdiv(n4, 16)), dim3(256), 0, stream, dst, slabs, nslabs, n4, stride,
                           scale, accumulate, bias, ncols, act);
        ACKTR_LAUNCH_CHECK();
        return 0;
    }
    const int groups = (group_ws && nslabs >= 16) ? (nslabs >= 128 ? 32 : 8) : 1;
    if (groups == 1) {
        hipLaunchKernelGGL(slab_reduce4_kernel, dim3(bx, 1), dim3(256), 0, stream, dst, slabs, 0, nslabs, n4, stride, scale,
                           accumulate, bias, ncols, act, nslabs, 0L);
        ACKTR_LAUNCH_CHECK();
        return 0;
    }
    const int per = cdiv(nslabs, groups);
    const int ng = cdiv(nslabs, per);
    hipLaunchKernelGGL(slab_reduce4_kernel, dim3(bx, ng), dim3(256), 0, stream, group_ws, slabs, 0, nslabs, n4, stride, 1.0f,
                       0, (const float*)nullptr, 1, 0, per, n);
    ACKTR_LAUNCH_CHECK();
    hipLaunchKernelGGL(slab_reduce4_kernel, dim3(bx, 1), dim3(256), 0, stream, dst, (const float*)group_ws, 0, ng, n4, n, scale,
                       accumulate, bias, ncols, act, ng, 0L);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

// scalar fallback for tiny / unaligned reductions (bias partials etc.)
__global__ __launch_bounds__(256) void slab_reduce_kernel(float* dst, const float* slabs, int nslabs, long n,
                                                          long stride, float scale, int accumulate) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < nslabs; ++k) s += slabs[(long)k * stride + i];   // fixed order: deterministic
        s *= scale;
        if (accumulate) s += dst[i];
        dst[i] = s;
    }
}

int launch_slab_reduce(float* dst, const float* slabs, int nslabs, long n, long stride, float scale, int accumulate,
                       hipStream_t stream) {
    if (n <= 0) return 0;
    if (!(n & 3) && !(stride & 3) && !(((uintptr_t)dst | (uintptr_t)slabs) & 15)) {
        // group partials live right behind the slabs (callers size workspaces with SLAB_GROUP_EXTRA slabs of slack)
        float* gws = const_cast<float*>(slabs) + (long)nslabs * stride;
        return launch_slab_reduce_ex(dst, slabs, nslabs, n, stride, scale, accumulate, nullptr, 1, 0,
                                     nslabs >= 16 ? gws : nullptr, stream);
    }
    int blocks = cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, stream, dst, slabs, nslabs, n, stride, scale,
                       accumulate);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

// Tile choice.  256 CUs want >= ~768 resident workgroups (3 waves per SIMD); LDS per workgroup sets how many fit:
// 128x128 (74 KB, 2/CU) only when it alone fills the chip, 64x128 (55 KB, 2/CU) when it gives >= 256 tiles,
// otherwise 64x64 (37 KB, 4/CU) -- the 1024x512x1568-class layers of this workload ran at < 1 wave per SIMD on
// 64x128.  Skinny shapes get 128x32 / 32x128.  Split-K supplies the rest (see pick_slices).
enum { TILE_128x32, TILE_32x128, TILE_64x64, TILE_64x128, TILE_128x128 };
static int env_int(const char* name) {           // tuning overrides for tools/opbench.py sweeps (unset in production)
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
}
static int tile_cfg(int M, int N) {
    if (N <= 64) return TILE_128x32;
    if (M <= 32) return TILE_32x128;
    if (const int t = env_int("ACKTR_TILE")) return t;
    if ((long)cdiv(M, 128) * cdiv(N, 128) >= 192) return TILE_128x128;
    if ((long)cdiv(M, 64) * cdiv(N, 128) >= 256) return TILE_64x128;
    return TILE_64x64;
}
template <class MOp, class NOp, class Epi>
static int gemm_dispatch(const MOp& mop, const NOp& nop, const Epi& epi, int M, int N, int K, int nslices,
                         hipStream_t s) {
    switch (tile_cfg(M, N)) {
        case TILE_128x32: return launch_gemm<128, 32, 4, 1>(mop, nop, epi, M, N, K, nslices, s);
        case TILE_32x128: return launch_gemm<32, 128, 1, 4>(mop, nop, epi, M, N, K, nslices, s);
        case TILE_64x64: return launch_gemm<64, 64, 2, 2>(mop, nop, epi, M, N, K, nslices, s);
        case TILE_64x128: return launch_gemm<64, 128, 2, 2>(mop, nop, epi, M, N, K, nslices, s);
        default: return launch_gemm<128, 128, 2, 2>(mop, nop, epi, M, N, K, nslices, s);
    }
}
long gemm_tiles(int M, int N) {
    switch (tile_cfg(M, N)) {
        case TILE_128x32: return (long)cdiv(M, 128) * cdiv(N, 32);
        case TILE_32x128: return (long)cdiv(M, 32) * cdiv(N, 128);
        case TILE_64x64: return (long)cdiv(M, 64) * cdiv(N, 64);
        case TILE_64x128: return (long)cdiv(M, 64) * cdiv(N, 128);
        default: return (long)cdiv(M, 128) * cdiv(N, 128);
    }
}

// Generic C[M,N] (row major, ldc) = A(m,k) * B(n,k) with explicit element strides per operand:
// an operand is "k contiguous" (stride_k == 1) or "row contiguous" (stride_row == 1).
template <bool VEC>
static int gemm_rowmajor_t(const float* a, long a_srow, long a_sk, const float* b, long b_srow, long b_sk,
                           const RowMajorEpi& epi, int M, int N, int K, int nslices, hipStream_t s) {
    const bool a_kc = (a_sk == 1), b_kc = (b_sk == 1);
    if (a_kc && b_kc) {
        DenseKCT<VEC> A{a, M, K, a_srow}, B{b, N, K, b_srow};
        return gemm_dispatch(A, B, epi, M, N, K, nslices, s);
    } else if (a_kc && !b_kc) {
        DenseKCT<VEC> A{a, M, K, a_srow};
        DenseRCT<VEC> B{b, N, K, b_sk};
        return gemm_dispatch(A, B, epi, M, N, K, nslices, s);
    } else if (!a_kc && b_kc) {
        DenseRCT<VEC> A{a, M, K, a_sk};
        DenseKCT<VEC> B{b, N, K, b_srow};
        return gemm_dispatch(A, B, epi, M, N, K, nslices, s);
    } else {
        DenseRCT<VEC> A{a, M, K, a_sk}, B{b, N, K, b_sk};
        return gemm_dispatch(A, B, epi, M, N, K, nslices, s);
    }
}
int gemm_rowmajor(const float* a, long a_srow, long a_sk, const float* b, long b_srow, long b_sk, const RowMajorEpi& epi,
                  int M, int N, int K, int nslices, hipStream_t s) {
    const bool a_kc = (a_sk == 1), b_kc = (b_sk == 1);
    if (!a_kc && a_srow != 1) { set_error("gemm: operand A must be contiguous along k or rows"); return -1; }
    if (!b_kc && b_srow != 1) { set_error("gemm: operand B must be contiguous along k or rows"); return -1; }
    // vector (branch-free) loaders need 16-byte aligned rows and a contiguous extent that is a multiple of 4
    const bool va = a_kc ? (vec_ok(a, a_srow) && K % 4 == 0) : (vec_ok(a, a_sk) && M % 4 == 0);
    const bool vb = b_kc ? (vec_ok(b, b_srow) && K % 4 == 0) : (vec_ok(b, b_sk) && N % 4 == 0);
    if (va && vb) return gemm_rowmajor_t<true>(a, a_srow, a_sk, b, b_srow, b_sk, epi, M, N, K, nslices, s);
    return gemm_rowmajor_t<false>(a, a_srow, a_sk, b, b_srow, b_sk, epi, M, N, K, nslices, s);
}

__global__ __launch_bounds__(256) void symmetrize_lower_kernel(float* c, int n) {
    const long total = (long)n * n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int i = (int)(e / n), j = (int)(e - (long)i * n);
        if (i > j) c[e] = c[(long)j * n + i];
    }
}
int launch_symmetrize_lower(float* c, int n, hipStream_t stream) {
    int blocks = cdiv((long)n * n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(symmetrize_lower_kernel, dim3(blocks), dim3(256), 0, stream, c, n);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

int pick_slices(long tiles, long K, long max_by_ws) {
    long want = 768 / (tiles > 0 ? tiles : 1);
    if (tiles >= 160 && K <= 2048) want = 1;   // the tiles alone nearly fill the chip and K is short: a split would only
                                               // add slab traffic and a reduction launch (FC wgrad, 200 tiles: 36.2 vs 39.5 us)
    long by_k = K / (4 * GK);              // at least 4 K tiles per slice
    long n = want;
    if (n > by_k) n = by_k;
    if (const int f = env_int("ACKTR_WGRAD_SLICES")) n = f;
    if (n > max_by_ws) n = max_by_ws;
    if (n > 1024) n = 1024;
    if (n < 1) n = 1;
    return (int)n;
}

}  // namespace acktr

using namespace acktr;

static int fwd_slices(int B, int IN, int OUT, long ws_floats) {
    if (((long)B * OUT) % 4) return 1;
    long max_by_ws = ws_floats / ((long)B * OUT) - SLAB_GROUP_EXTRA;
    long want = 768 / gemm_tiles(B, OUT);
    long by_k = IN / (12 * GK);                // at least 12 K tiles per slice (FC 1568 -> 512 at B = 1024: 4 slices 31.9 us,
                                               // 6 slices 34.6, 8 slices 36.9 -- the slabs cost more than the extra waves buy)
    long n = want < by_k ? want : by_k;
    if (const int f = env_int("ACKTR_FWD_SLICES")) n = f;
    if (n > max_by_ws) n = max_by_ws;
    if (n > 16) n = 16;
    return n < 2 ? 1 : (int)n;
}

// ==========================================================================================
// Skinny layers (OUT <= 8: the fused critic+actor head [1+A, hidden], the MLP critic head).  A 32x32 MFMA tile
// would be 3/4 padding and the launch is latency-bound, so these run on the vector ALUs, streaming each
// activation once with 16-byte loads.  Requirements: IN % 4 == 0, 16-byte aligned rows (else the tile GEMM runs).
// ==========================================================================================
constexpr int SK_MAX_OUT = 8;
static bool skinny_ok(const void* a, const void* b, const void* c, int IN, int OUT) {
    return OUT <= SK_MAX_OUT && IN % 4 == 0 && IN >= 4 &&
           ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}

// y[b, o] = act(sum_i x[b,i] w[o,i] + bias[o]); one wave per row, lanes stride the reduction in float4s
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int B,
                                                         int IN, int OUT, int act) {
    constexpr int RW = 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * 4 + wave) * RW;
    if (r0 >= B) return;
    float acc[RW][SK_MAX_OUT];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int o = 0; o < SK_MAX_OUT; ++o) acc[r][o] = 0.f;
    for (int k = lane * 4; k < IN; k += 256) {
        f32x4 xv[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int rr = r0 + r < B ? r0 + r : B - 1;
            xv[r] = *reinterpret_cast<const f32x4*>(x + (long)rr * IN + k);
        }
#pragma unroll
        for (int o = 0; o < SK_MAX_OUT; ++o) {
            if (o < OUT) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (long)o * IN + k);
#pragma unroll
                for (int r = 0; r < RW; ++r)
                    acc[r][o] += xv[r][0] * wv[0] + xv[r][1] * wv[1] + xv[r][2] * wv[2] + xv[r][3] * wv[3];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int o = 0; o < SK_MAX_OUT; ++o) {
            if (o < OUT) {                                   // wave-uniform
                float v = wave_sum(acc[r][o]);
                if (lane == 0 && r0 + r < B) {
                    v += bias ? bias[o] : 0.f;
                    if (act == 1) v = v > 0.f ? v : 0.f;
                    else if (act == 2) v = tanhf(v);
                    y[(long)(r0 + r) * OUT + o] = v;
                }
            }
        }
}

// dx[b, i..i+3] (=|+=) (sum_o dy[b,o] w[o,i..i+3]) * gate
__global__ __launch_bounds__(256) void skinny_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                           const float* __restrict__ gate_src, int gate,
                                                           float* __restrict__ dx, int B, int IN, int OUT,
                                                           int accumulate) {
    const int q = IN / 4;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)B * q) return;
    const int b = (int)(t / q), i = (int)(t % q) * 4;
    f32x4 a = zero4();
#pragma unroll
    for (int o = 0; o < SK_MAX_OUT; ++o) {
        if (o < OUT) {
            const float d = dy[(long)b * OUT + o];
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (long)o * IN + i);
            a += d * wv;
        }
    }
    const long off = (long)b * IN + i;
    if (gate) {
        const f32x4 h = *reinterpret_cast<const f32x4*>(gate_src + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = gate == 1 ? (h[j] > 0.f ? a[j] : 0.f) : a[j] * (1.f - h[j] * h[j]);
    }
    if (accumulate) a += *reinterpret_cast<const f32x4*>(dx + off);
    *reinterpret_cast<f32x4*>(dx + off) = a;
}

// per row slice: pw[slice][o][i] = sum_{b in slice} dy[b,o] x[b,i];  pb[slice][o] = sum_b dy[b,o]
// A thread owns one 16-byte column quad; when the layer has fewer than 128 quads (the 64-wide MLP heads: 16) the other
// threads of the workgroup take every nsub-th row of the slice and the partials meet in LDS in sub-slice order -- with one
// row walker per quad the 65536-row minibatches of configs[4] kept 16 lanes of a workgroup busy for 241 us.
__global__ __launch_bounds__(128) void skinny_wgrad_partial_kernel(const float* __restrict__ dy,
                                                                  const float* __restrict__ x, float* __restrict__ pw,
                                                                  float* __restrict__ pb, int B, int IN, int OUT,
                                                                  int rows_per_slice) {
    __shared__ f32x4 red[128][SK_MAX_OUT];                             // 16 KB
    __shared__ float redb[128][SK_MAX_OUT];
    const int q = IN / 4;
    const int nsub = q < 128 ? 128 / q : 1;                            // row walkers per quad (grid.x == 1 when q < 128)
    const int sub = q < 128 ? threadIdx.x / q : 0;
    const int cq = q < 128 ? threadIdx.x % q : blockIdx.x * 128 + threadIdx.x;
    const int c4 = cq * 4;
    const bool live = sub < nsub && c4 < IN;
    const int r0 = blockIdx.y * rows_per_slice;
    int r1 = r0 + rows_per_slice;
    if (r1 > B) r1 = B;
    f32x4 acc[SK_MAX_OUT];
    float bs[SK_MAX_OUT];
#pragma unroll
    for (int o = 0; o < SK_MAX_OUT; ++o) { acc[o] = zero4(); bs[o] = 0.f; }
    const bool bias_lane = pb && cq == 0 && blockIdx.x == 0;           // one lane per row walker also sums dy
    if (live) {
        for (int r = r0 + sub; r < r1; r += nsub) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long)r * IN + c4);
#pragma unroll
            for (int o = 0; o < SK_MAX_OUT; ++o)
                if (o < OUT) {
                    const float d = dy[(long)r * OUT + o];
                    acc[o] += d * xv;
                    if (bias_lane) bs[o] += d;
                }
        }
    }
    if (nsub == 1) {                                                   // (uniform) one walker per quad: nothing to combine
        if (live) {
#pragma unroll
            for (int o = 0; o < SK_MAX_OUT; ++o)
                if (o < OUT) *reinterpret_cast<f32x4*>(pw + ((long)blockIdx.y * OUT + o) * IN + c4) = acc[o];
            if (bias_lane)
#pragma unroll
                for (int o = 0; o < SK_MAX_OUT; ++o)
                    if (o < OUT) pb[(long)blockIdx.y * OUT + o] = bs[o];
        }
        return;
    }
#pragma unroll
    for (int o = 0; o < SK_MAX_OUT; ++o) { red[threadIdx.x][o] = acc[o]; redb[threadIdx.x][o] = bs[o]; }
    __syncthreads();
    if (threadIdx.x < q) {                                             // sub-slices added in order: deterministic
#pragma unroll
        for (int o = 0; o < SK_MAX_OUT; ++o)
            if (o < OUT) {
                f32x4 t = red[threadIdx.x][o];
                for (int k = 1; k < nsub; ++k) t += red[k * q + threadIdx.x][o];
                *reinterpret_cast<f32x4*>(pw + ((long)blockIdx.y * OUT + o) * IN + c4) = t;
            }
    }
    if (pb && threadIdx.x < OUT) {
        float t = 0.f;
        for (int k = 0; k < nsub; ++k) t += redb[k * q][threadIdx.x];
        pb[(long)blockIdx.y * OUT + threadIdx.x] = t;
    }
}
// db[o] = sum over slices of pb[slice][o], fixed order: thread (g, o) sums slices g, g+32, ..., then the 32 groups in order
__global__ __launch_bounds__(256) void skinny_bias_reduce_kernel(const float* __restrict__ pb, int nsl, int OUT,
                                                                 float* __restrict__ db) {
    __shared__ float sm[32][SK_MAX_OUT];
    const int o = threadIdx.x & 7, g = threadIdx.x >> 3;
    float v = 0.f;
    if (o < OUT)
        for (int k = g; k < nsl; k += 32) v += pb[(long)k * OUT + o];
    sm[g][o] = v;
    __syncthreads();
    if (threadIdx.x < OUT) {
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += sm[k][threadIdx.x];
        db[threadIdx.x] = t;
    }
}
static int skinny_slices(int B) {
    int n = B / 8;
    if (n < 1) n = 1;
    if (n > 128) n = B >= 16384 ? 512 : 128;          // (65536-row minibatches: 128 slices left 512 rows to every workgroup)
    return n;
}

// ==========================================================================================
// Small layers (K <= 128, N <= 128, few thousand rows at most: the GAIL discriminator's 23 -> 100 -> 100 trunk, MLP
// policies at rollout batch sizes).  These launches are pure latency: the tile GEMM's LDS staging and its barrier per K tile
// cost ~9.5 us for 10 MFLOP.  Here ONE wave owns a 32x32 output tile and its operands go from global memory straight into
// the MFMA operand registers, all loads in flight at once, no LDS, no barrier:
//   A (activations, k contiguous): lane (li, kh) loads 16-byte quads 2j + kh of row m0 + li; element e of quad j feeds
//     MFMA step 4j + e, i.e. that step's k is 8j + 4kh + e -- the sum over k does not care about the order as long as B agrees;
//   B k-contiguous (forward, w[n, k]): the same quads of row n0 + li;  B n-contiguous (dgrad, w[k, n]): dword
//     w[(8j + 4kh + e) * N + n0 + li], coalesced across li.
// Rows that are not 16-byte aligned or K % 4 != 0 (the 23-wide first layer) take dword loads with k = 2s + kh.
// ==========================================================================================
constexpr int SMALL_MAX_K = 128, SMALL_MAX_N = 128, SMALL_MAX_M = 8192;
template <bool VEC, bool BKC>
__global__ __launch_bounds__(64) void small_linear_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ bias, const float* __restrict__ gate_src,
                                                          float* __restrict__ c, int M, int N, int K, int act, int gate,
                                                          int accumulate) {
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int lane = threadIdx.x, li = lane & 31, kh = lane >> 5;
    const int row = m0 + li, col = n0 + li;
    const bool rok = row < M, cok = col < N;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    if constexpr (VEC) {
        const int nq = K / 4;
        const float* ap = a + (long)(rok ? row : 0) * K;
        f32x4 av[SMALL_MAX_K / 8], bv[SMALL_MAX_K / 8];
#pragma unroll
        for (int j = 0; j < SMALL_MAX_K / 8; ++j) {                  // (no early exit: the arrays must keep static indices)
            const int q = 2 * j + kh;
            const bool ok = q < nq;
            av[j] = (ok && rok) ? *reinterpret_cast<const f32x4*>(ap + 4 * q) : zero4();
            if constexpr (BKC) {
                bv[j] = (ok && cok) ? *reinterpret_cast<const f32x4*>(b + (long)col * K + 4 * q) : zero4();
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[j][e] = (ok && cok) ? b[(long)(4 * q + e) * N + col] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < SMALL_MAX_K / 8; ++j) {
            if (2 * j < nq) {                                        // wave-uniform
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][e], bv[j][e], acc, 0, 0, 0);
            }
        }
    } else {
        const int ns = (K + 1) / 2;
        for (int s0 = 0; s0 < ns; s0 += 16) {
            float av[16], bv[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = 2 * (s0 + s) + kh;
                const bool ok = k < K;
                av[s] = (ok && rok) ? a[(long)row * K + k] : 0.f;
                bv[s] = (ok && cok) ? (BKC ? b[(long)col * K + k] : b[(long)k * N + col]) : 0.f;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s)
                if (s0 + s < ns) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
        }
    }
    if (!cok) return;
    const float bn = bias ? bias[col] : 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int m = m0 + 4 * kh + (v & 3) + 8 * (v >> 2);
        if (m >= M) continue;
        const long off = (long)m * N + col;
        float r = acc[v] + bn;
        if (act == 1) r = r > 0.f ? r : 0.f;
        else if (act == 2) r = tanhf(r);
        if (gate) {
            const float h = gate_src[off];
            r = gate == 1 ? (h > 0.f ? r : 0.f) : r * (1.f - h * h);
        }
        if (accumulate) r += c[off];
        c[off] = r;
    }
}
static bool small_ok(int M, int N, int K) {
    static const bool off = getenv("ACKTR_LINEAR_SMALL_OFF") != nullptr;      // tuning / comparison switch
    return !off && K <= SMALL_MAX_K && N <= SMALL_MAX_N && N > SK_MAX_OUT && M <= SMALL_MAX_M;
}
// c[M, N] = epilogue(sum_k a[m, k] * B(n, k)); b_kc: B(n, k) = b[n*K + k], else b[k*N + n]
static int small_linear(const float* a, const float* b, bool b_kc, const float* bias, const float* gate_src, float* c, int M,
                        int N, int K, int act, int gate, int accumulate, hipStream_t s) {
    const bool vec = K % 4 == 0 && (((uintptr_t)a) & 15) == 0 && (!b_kc || (((uintptr_t)b) & 15) == 0);
    const dim3 grid(cdiv(M, 32), cdiv(N, 32)), block(64);
    if (vec && b_kc) hipLaunchKernelGGL((small_linear_kernel<true, true>), grid, block, 0, s, a, b, bias, gate_src, c, M, N, K, act, gate, accumulate);
    else if (vec) hipLaunchKernelGGL((small_linear_kernel<true, false>), grid, block, 0, s, a, b, bias, gate_src, c, M, N, K, act, gate, accumulate);
    else if (b_kc) hipLaunchKernelGGL((small_linear_kernel<false, true>), grid, block, 0, s, a, b, bias, gate_src, c, M, N, K, act, gate, accumulate);
    else hipLaunchKernelGGL((small_linear_kernel<false, false>), grid, block, 0, s, a, b, bias, gate_src, c, M, N, K, act, gate, accumulate);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" long acktr_linear_fwd_workspace_bytes(int B, int IN, int OUT) {
    const int ns = fwd_slices(B, IN, OUT, 1L << 40);
    return ns > 1 ? (long)(ns + SLAB_GROUP_EXTRA) * B * OUT * (long)sizeof(float) : 0;
}

extern "C" int acktr_linear_fwd_f32(const float* x, const float* w, const float* bias, float* y, int B, int IN,
                                    int OUT, int act, void* workspace, long workspace_bytes, void* stream) {
    ACKTR_ARG(x && w && y, "linear_fwd: null pointer");
    ACKTR_ARG(B > 0 && IN > 0 && OUT > 0, "linear_fwd: bad sizes B=%d IN=%d OUT=%d", B, IN, OUT);
    ACKTR_ARG(act >= 0 && act <= 2, "linear_fwd: act must be 0 (none), 1 (relu) or 2 (tanh)");
    hipStream_t s = (hipStream_t)stream;
    if (skinny_ok(x, w, x, IN, OUT)) {
        hipLaunchKernelGGL(skinny_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, x, w, bias, y, B, IN, OUT, act);
        ACKTR_LAUNCH_CHECK();
        return 0;
    }
    if (small_ok(B, OUT, IN)) return small_linear(x, w, true, bias, nullptr, y, B, OUT, IN, act, 0, 0, s);
    float* ws = (float*)workspace;
    const int ns = (ws && (((uintptr_t)ws | (uintptr_t)y) & 15) == 0)
                       ? fwd_slices(B, IN, OUT, workspace_bytes / (long)sizeof(float)) : 1;
    if (ns > 1) {   // few output tiles, long K: split K into slabs, bias + activation fused into the reduction
        RowMajorEpi e = row_major_epi(ws, B, OUT, OUT);
        e.slab_stride = (long)B * OUT;
        int rc = gemm_rowmajor(x, IN, 1, w, IN, 1, e, B, OUT, IN, ns, s);
        if (rc) return rc;
        return launch_slab_reduce_ex(y, ws, ns, (long)B * OUT, (long)B * OUT, 1.f, 0, bias, OUT, act, nullptr, s);
    }
    RowMajorEpi e = row_major_epi(y, B, OUT, OUT);
    e.bias_n = bias;
    e.act = act;
    return gemm_rowmajor(x, IN, 1, w, IN, 1, e, B, OUT, IN, 1, s);
}

extern "C" long acktr_linear_fwd_slabs_workspace_bytes(int B, int IN, int OUT) {
    const int ns = fwd_slices(B, IN, OUT, 1L << 40);
    return (long)(ns + SLAB_GROUP_EXTRA) * B * OUT * (long)sizeof(float);
}

extern "C" int acktr_linear_fwd_slabs_f32(const float* x, const float* w, int B, int IN, int OUT, void* workspace,
                                          long workspace_bytes, int* nslabs_out, void* stream) {
    ACKTR_ARG(x && w && workspace && nslabs_out, "linear_fwd_slabs: null pointer");
    ACKTR_ARG(B > 0 && IN > 0 && OUT > 0, "linear_fwd_slabs: bad sizes B=%d IN=%d OUT=%d", B, IN, OUT);
    ACKTR_ARG(((uintptr_t)workspace & 15) == 0 && workspace_bytes >= (long)B * OUT * (long)sizeof(float),
              "linear_fwd_slabs: workspace must be 16-byte aligned and hold at least one [B, OUT] slab");
    float* ws = (float*)workspace;
    const int ns = fwd_slices(B, IN, OUT, workspace_bytes / (long)sizeof(float) + (long)SLAB_GROUP_EXTRA * B * OUT);
    RowMajorEpi e = row_major_epi(ws, B, OUT, OUT);
    e.slab_stride = (long)B * OUT;
    *nslabs_out = ns;
    return gemm_rowmajor(x, IN, 1, w, IN, 1, e, B, OUT, IN, ns, (hipStream_t)stream);
}

extern "C" int acktr_linear_dgrad_f32(const float* dy, const float* w, const float* gate_src, int gate, float* dx,
                                      int B, int IN, int OUT, int accumulate, void* stream) {
    return acktr_linear_dgrad_ex_f32(dy, w, gate_src, gate, dx, B, IN, OUT, accumulate, nullptr, 0, 0, 0, 0, stream);
}

extern "C" int acktr_linear_dgrad_ex_f32(const float* dy, const float* w, const float* gate_src, int gate, float* dx,
                                         int B, int IN, int OUT, int accumulate, float* dx_padded, int C, int OH, int OW,
                                         int pad, void* stream) {
    ACKTR_ARG(dy && w && dx, "linear_dgrad: null pointer");
    ACKTR_ARG(B > 0 && IN > 0 && OUT > 0, "linear_dgrad: bad sizes");
    ACKTR_ARG(gate >= 0 && gate <= 2 && (gate == 0 || gate_src), "linear_dgrad: gate needs gate_src");
    ACKTR_ARG(!dx_padded || (C > 0 && OH > 0 && OW > 0 && pad > 0 && pad <= 4 && (long)C * OH * OW == IN && OUT > SK_MAX_OUT &&
                             (long)B * C * (OH + 2 * pad) * (OW + 2 * pad) < (1L << 31)),
              "linear_dgrad: dx_padded needs IN == C*OH*OW, 1 <= pad <= 4 and OUT > %d", SK_MAX_OUT);
    if (!dx_padded && skinny_ok(w, dx, gate_src, IN, OUT)) {
        hipLaunchKernelGGL(skinny_dgrad_kernel, dim3(cdiv((long)B * (IN / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                           dy, w, gate_src, gate, dx, B, IN, OUT, accumulate);
        ACKTR_LAUNCH_CHECK();
        return 0;
    }
    if (!dx_padded && small_ok(B, IN, OUT))
        return small_linear(dy, w, false, nullptr, gate_src, dx, B, IN, OUT, 0, gate, accumulate, (hipStream_t)stream);
    RowMajorEpi e = row_major_epi(dx, B, IN, IN);
    e.gate = gate; e.gate_src = gate_src; e.ldg = IN; e.accumulate = accumulate;
    if (dx_padded) {                       // second copy as zero-padded [B, C, OH+2p, OW+2p] planes (interior only)
        const int pw = OW + 2 * pad, psz = (OH + 2 * pad) * pw;
        e.c2 = dx_padded; e.ldc2 = (long)C * psz; e.c2_sp = OH * OW; e.c2_ow = OW; e.c2_pw = pw; e.c2_psz = psz;
        e.c2_off0 = pad * pw + pad;
    }
    // dX[b, i] = sum_o dY[b, o] * W[o, i]   : A = dY (k contiguous), B(n=i, k=o) = W[o*IN + i] (row contiguous)
    return gemm_rowmajor(dy, OUT, 1, w, 1, IN, e, B, IN, OUT, 1, (hipStream_t)stream);
}

// ==========================================================================================
// Weight gradient of a mid-size layer whose 64x64 output tiles number about one per CU (the policy's 1568 -> 512 FC
// layer: 8 x 25 tiles): dW[o, i] = sum_b dY[b, o] X[b, i], db[o] = sum_b dY[b, o].  Both operands are contiguous along
// the OUTPUT indices, so the MFMA operands of a 32x32x2 step are plain coalesced dword loads (lane (li, kh): row
// b = 2s + kh, column o0 + li / i0 + li) -- no LDS staging, no barrier in the K loop.  One workgroup per tile, its eight
// waves split the batch rows eight ways (the tile GEMM at 200 tiles either ran one 4-wave workgroup per CU over the
// whole K, or paid a slab round trip for split K), partial tiles meet in LDS and are added in wave order.
// ==========================================================================================
constexpr int FW_WAVES = 8, FW_LDS_BYTES = FW_WAVES * 64 * 64 * 4;
// V2 (OUT and IN even, 8-byte aligned operands): a lane loads TWO adjacent outputs with one 8-byte load and the two 32-wide
// MFMA blocks of a tile side are interleaved (block a = outputs m0 + 2i + a): half the load instructions per MFMA -- with
// dword loads the kernel ran at the same ~48 TFLOP/s for 1024 and for 8192 batch rows, paced by load issue, not by the MFMAs.
// U = k pairs per register block: 2 measured best (26.0 us for the policy's FC layer; 1: 27.7, 3: 26.4, 4: 27.0, 8: 31.0;
// 16 waves with U = 4 / 2: 33.2 / 30.1) -- the loads alone and the MFMAs alone each take the same ~13.7 us of this kernel's
// loop (64x64 tiles: 16 flop per loaded byte, and a CU draws ~16 B/clk from L2), so what counts is how finely the two
// interleave, not how far ahead the loads run.
template <bool V2, int W = FW_WAVES, int U = 2>
__global__ __launch_bounds__(64 * W) void fc_wgrad_sa_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                      float* __restrict__ dw, float* __restrict__ db, int B,
                                                                      int IN, int OUT, int tiles_n) {
    typedef float fw2 __attribute__((ext_vector_type(2)));
    typedef fw2 fw2u __attribute__((aligned(8)));
    extern __shared__ __attribute__((aligned(16))) float fw_smem[];            // [wave][blk 4][reg 16][lane 64]
    // workgroups go round-robin over the 8 XCDs: XCD j takes the column tiles j, j+8, ... of every row tile, so its L2 keeps
    // 1/8 of X next to all of dY (X + dY = 8.4 MB for the policy's FC layer against 4 MB of L2 per XCD)
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, slots = (tiles_n + 7) >> 3;
    const int nt = xcd + 8 * (local % slots);
    if (nt >= tiles_n) return;                                                  // (uniform: before any barrier)
    const int m0 = (local / slots) * 64, n0 = nt * 64;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, li = lane & 31, kh = lane >> 5;
    const int per = ((B + 2 * W - 1) / (2 * W)) * 2;                            // rows per wave, even
    const int k0 = wave * per, k1 = min(B, k0 + per);
    // column of block a held by this lane: V2: m0 + 2 li + a, else m0 + 32 a + li
    const int mc0 = V2 ? m0 + 2 * li : m0 + li, mc1 = V2 ? mc0 + 1 : mc0 + 32;
    const int nc0 = V2 ? n0 + 2 * li : n0 + li, nc1 = V2 ? nc0 + 1 : nc0 + 32;
    const bool mok0 = mc0 < OUT, mok1 = mc1 < OUT, nok0 = nc0 < IN, nok1 = nc1 < IN;
    const float* ap = dy + mc0;
    const float* xp = x + nc0;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    float bs0 = 0.f, bs1 = 0.f;
    float av[2][U][2], xv[2][U][2];
    auto load_blk = [&](int buf, int k) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k + 2 * u + kh;
            const bool ok = kk < k1;
            const long ra = (long)(ok ? kk : 0) * OUT, rx = (long)(ok ? kk : 0) * IN;
            if constexpr (V2) {                                                 // (pairs are whole: OUT, IN, m0, n0 even)
                const fw2 a2 = (ok && mok0) ? *reinterpret_cast<const fw2u*>(ap + ra) : fw2{0.f, 0.f};
                const fw2 x2 = (ok && nok0) ? *reinterpret_cast<const fw2u*>(xp + rx) : fw2{0.f, 0.f};
                av[buf][u][0] = a2[0]; av[buf][u][1] = a2[1];
                xv[buf][u][0] = x2[0]; xv[buf][u][1] = x2[1];
            } else {
                av[buf][u][0] = (ok && mok0) ? ap[ra] : 0.f;
                av[buf][u][1] = (ok && mok1) ? ap[ra + 32] : 0.f;
                xv[buf][u][0] = (ok && nok0) ? xp[rx] : 0.f;
                xv[buf][u][1] = (ok && nok1) ? xp[rx + 32] : 0.f;
            }
        }
    };
    auto mul_blk = [&](int buf) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bs0 += av[buf][u][0];
            bs1 += av[buf][u][1];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u][a], xv[buf][u][b], acc[a][b], 0, 0, 0);
        }
    };
    if (k0 < k1) {
        load_blk(0, k0);
        for (int k = k0; k < k1; k += 4 * U) {                                  // two register blocks per trip: static indices
            load_blk(1, k + 2 * U);
            __builtin_amdgcn_sched_barrier(0);
            mul_blk(0);
            __builtin_amdgcn_sched_barrier(0);
            load_blk(0, k + 4 * U);
            __builtin_amdgcn_sched_barrier(0);
            mul_blk(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* mine = fw_smem + (long)(wave & (FW_WAVES - 1)) * 4096;
    __shared__ float bsum[W][2][64];
    bsum[wave][0][lane] = bs0;
    bsum[wave][1][lane] = bs1;
    if (wave < FW_WAVES) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int v = 0; v < 16; ++v) mine[((a * 2 + b) * 16 + v) * 64 + lane] = acc[a][b][v];
    }
    __syncthreads();
    if constexpr (W > FW_WAVES) {                                               // waves 8..15 add into the slot of wave - 8
        if (wave >= FW_WAVES) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int v = 0; v < 16; ++v) mine[((a * 2 + b) * 16 + v) * 64 + lane] += acc[a][b][v];
        }
        __syncthreads();
    }
    // the eight partial tiles added in wave order; accumulator (a, b, v, l) is row i = 4 (l/32) + (v&3) + 8 (v>>2), column
    // j = l%32 of block (a, b)
    if constexpr (V2) {
        for (int e = threadIdx.x; e < 2048; e += 64 * W) {                     // (a, v, l): both b of a lane, one 8-byte store
            const int a = e >> 10, v = (e >> 6) & 15, l = e & 63;
            fw2 sum = fw2{0.f, 0.f};
#pragma unroll
            for (int w = 0; w < FW_WAVES; ++w) {
                sum[0] += fw_smem[w * 4096 + ((a * 2 + 0) * 16 + v) * 64 + l];
                sum[1] += fw_smem[w * 4096 + ((a * 2 + 1) * 16 + v) * 64 + l];
            }
            const int m = m0 + 2 * (4 * (l >> 5) + (v & 3) + 8 * (v >> 2)) + a, n = n0 + 2 * (l & 31);
            if (m < OUT && n < IN) *reinterpret_cast<fw2u*>(dw + (long)m * IN + n) = sum;
        }
    } else {
        for (int e = threadIdx.x; e < 4096; e += 64 * W) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < FW_WAVES; ++w) sum += fw_smem[w * 4096 + e];
            const int blk = e >> 10, v = (e >> 6) & 15, l = e & 63;
            const int m = m0 + 32 * (blk >> 1) + 4 * (l >> 5) + (v & 3) + 8 * (v >> 2), n = n0 + 32 * (blk & 1) + (l & 31);
            if (m < OUT && n < IN) dw[(long)m * IN + n] = sum;
        }
    }
    if (db && n0 == 0 && threadIdx.x < 64) {                                    // both k halves of a lane, eight waves
        const int a = threadIdx.x >> 5, l = threadIdx.x & 31;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) sum += bsum[w][a][l] + bsum[w][a][l + 32];
        const int m = V2 ? m0 + 2 * l + a : m0 + 32 * a + l;
        if (m < OUT) db[m] = sum;
    }
}
namespace acktr { int sw_num_cus(); }
static bool fc_wgrad_sa_ok(int B, int IN, int OUT) {
    static const bool off = getenv("ACKTR_FC_WGRAD_TILE") != nullptr;          // tuning / comparison switch
    const long tiles = (long)cdiv(OUT, 64) * cdiv(IN, 64);
    return !off && B >= 256 && B <= 4096 && OUT > SK_MAX_OUT && tiles >= 128 && tiles <= 2L * sw_num_cus();   // (B = 8192: the tile GEMM's long K loop is 0.3 % ahead)
}
template <bool V2, int W = FW_WAVES, int U = 2>
static int fc_wgrad_sa_t(const float* dy, const float* x, float* dw, float* db, int B, int IN, int OUT, hipStream_t s) {
    static bool attr_set = false;                                               // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fc_wgrad_sa_kernel<V2, W, U>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, FW_LDS_BYTES);
        if (e != hipSuccess) { set_error("fc_wgrad_sa: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int tn = cdiv(IN, 64);
    hipLaunchKernelGGL((fc_wgrad_sa_kernel<V2, W, U>), dim3(cdiv(OUT, 64) * cdiv(tn, 8) * 8), dim3(64 * W), FW_LDS_BYTES, s, dy,
                       x, dw, db, B, IN, OUT, tn);
    ACKTR_LAUNCH_CHECK();
    return 0;
}
static int fc_wgrad_sa(const float* dy, const float* x, float* dw, float* db, int B, int IN, int OUT, hipStream_t s) {
    const bool v2 = OUT % 2 == 0 && IN % 2 == 0 && ((((uintptr_t)dy) | ((uintptr_t)x) | ((uintptr_t)dw)) & 7) == 0;
    return v2 ? fc_wgrad_sa_t<true>(dy, x, dw, db, B, IN, OUT, s) : fc_wgrad_sa_t<false>(dy, x, dw, db, B, IN, OUT, s);
}

extern "C" long acktr_linear_wgrad_workspace_bytes(int B, int IN, int OUT) {
    if (OUT <= SK_MAX_OUT && IN % 4 == 0) {
        const long nsl = skinny_slices(B) + SLAB_GROUP_EXTRA;
        return (nsl * OUT * IN + nsl * OUT + 4) * (long)sizeof(float);
    }
    const long tiles = gemm_tiles(OUT, IN);
    const int ns = pick_slices(tiles, B, 1024);
    return (long)(ns + SLAB_GROUP_EXTRA) * ((long)OUT * IN + (OUT + 3) / 4 * 4) * (long)sizeof(float);
}

extern "C" int acktr_linear_wgrad_f32(const float* dy, const float* x, float* dw, float* db, int B, int IN, int OUT,
                                      void* workspace, long workspace_bytes, void* stream) {
    ACKTR_ARG(dy && x && dw, "linear_wgrad: null pointer");
    ACKTR_ARG(B > 0 && IN > 0 && OUT > 0, "linear_wgrad: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const long ws_floats = workspace ? workspace_bytes / (long)sizeof(float) : 0;
    if (skinny_ok(x, dw, ws, IN, OUT)) {
        const int nsl0 = skinny_slices(B);
        const int rps = cdiv(B, nsl0);
        const int nsl = cdiv(B, rps);
        const long wpart = (long)(nsl + SLAB_GROUP_EXTRA) * OUT * IN;
        const long need = wpart + (long)(nsl + SLAB_GROUP_EXTRA) * OUT;
        if (ws && ws_floats >= need) {
            float* pb = db ? ws + wpart : nullptr;
            hipLaunchKernelGGL(skinny_wgrad_partial_kernel, dim3(cdiv(IN / 4, 128), nsl), dim3(128), 0, s, dy, x, ws, pb,
                               B, IN, OUT, rps);
            ACKTR_LAUNCH_CHECK();
            int rc = launch_slab_reduce(dw, ws, nsl, (long)OUT * IN, (long)OUT * IN, 1.f, 0, s);
            if (rc) return rc;
            if (db) {
                hipLaunchKernelGGL(skinny_bias_reduce_kernel, dim3(1), dim3(256), 0, s, pb, nsl, OUT, db);
                ACKTR_LAUNCH_CHECK();
            }
            return 0;
        }
    }
    if (fc_wgrad_sa_ok(B, IN, OUT)) return fc_wgrad_sa(dy, x, dw, db, B, IN, OUT, s);
    // dW[o, i] = sum_b dY[b, o] * X[b, i] : A(m=o, k=b) = dy[b*OUT + o], B(n=i, k=b) = x[b*IN + i];
    // db[o] = sum_b dY[b, o] is the GEMM's M-operand row sum.  Slab layout [OUT*IN | OUT padded to 4] per K slice.
    const long wk = (long)OUT * IN, slab = wk + (OUT + 3) / 4 * 4;
    const long tiles = gemm_tiles(OUT, IN);
    long max_by_ws = ws_floats / slab - SLAB_GROUP_EXTRA;
    int ns = pick_slices(tiles, B, max_by_ws < 1 ? 1 : max_by_ws);
    if (ns > 1 && max_by_ws < ns) ns = 1;
    if (ns == 1) {
        RowMajorEpi e = row_major_epi(dw, OUT, IN, IN);
        e.rowsum_dst = db;
        return gemm_rowmajor(dy, 1, OUT, x, 1, IN, e, OUT, IN, B, 1, s);
    }
    RowMajorEpi e = row_major_epi(ws, OUT, IN, IN);
    e.slab_stride = slab;
    e.rowsum_dst = db ? ws + wk : nullptr;
    e.rowsum_stride = slab;
    int rc = gemm_rowmajor(dy, 1, OUT, x, 1, IN, e, OUT, IN, B, ns, s);
    if (rc) return rc;
    if (db == dw + wk && OUT % 4 == 0 && wk % 4 == 0) return launch_slab_reduce(dw, ws, ns, wk + OUT, slab, 1.f, 0, s);
    rc = launch_slab_reduce(dw, ws, ns, wk, slab, 1.f, 0, s);
    if (rc || !db) return rc;
    return launch_slab_reduce(db, ws + wk, ns, OUT, slab, 1.f, 0, s);
}

/* Generic strided fp32 GEMM: C[M,N] = alpha * A(m,k) B(n,k) (+ beta*C), used by the KFAC preconditioner.
 * Each operand must be contiguous along k or along its row index. */
extern "C" int acktr_gemm_f32(const float* a, long a_stride_row, long a_stride_k, const float* b, long b_stride_row,
                              long b_stride_k, float* c, long ldc, int M, int N, int K, float alpha, int accumulate,
                              void* stream) {
    ACKTR_ARG(a && b && c, "gemm: null pointer");
    ACKTR_ARG(M > 0 && N > 0 && K > 0 && ldc >= N, "gemm: bad sizes M=%d N=%d K=%d ldc=%ld", M, N, K, ldc);
    RowMajorEpi e = row_major_epi(c, M, N, ldc);
    e.scale = alpha;
    e.accumulate = accumulate;
    return gemm_rowmajor(a, a_stride_row, a_stride_k, b, b_stride_row, b_stride_k, e, M, N, K, 1, (hipStream_t)stream);
}
