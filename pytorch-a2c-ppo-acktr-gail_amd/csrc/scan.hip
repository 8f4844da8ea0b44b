// Return / GAE reverse scan, advantage normalisation, minibatch row gather.
//
// compute_returns restates a2c_ppo_acktr/storage.py:66-105 of the reference:
// a serial-in-t recurrence per environment whose fp32 association must be kept
// (bit-exact contract), so time cannot be parallelised.  Design for MI355X:
//   * one lane per environment, 32 (or 16, see ScanCfg) environments per workgroup (128-B rows);
//   * all 4 waves of the workgroup prefetch the NEXT 64-step chunk straight to
//     registers and pre-compute everything that does not depend on the carried
//     value (delta, gamma*lambda*mask, (1-bad)*V) -> LDS, double buffered;
//   * wave 0 runs the dependent chain (2-5 flops/step) out of LDS and streams
//     the returns rows back with coalesced stores.
// The chain latency (T x ~20 cycles) is the floor; HBM traffic is the
// algorithmic minimum (every input element read once, every output written once).
//
// No FMA contraction anywhere in this file (pragma below + -ffp-contract=off on the command line):
// products and sums are rounded separately, exactly as the reference's separate torch ops do.
// Plain operators are used on purpose: the __f*_rn() header wrappers are compiled under the
// header's own contraction mode and DO fuse into v_fma after inlining.
#include <stdlib.h>
#include <mutex>
#include <type_traits>

#include "common.hpp"
#include "../../include/acktr_hip.h"

#pragma clang fp contract(off)

namespace acktr {

constexpr int SCAN_LOADERS = 256;                            // waves 1..4: HBM loads, staging, result write-back
constexpr int SCAN_THREADS = 64 + SCAN_LOADERS;              // wave 0: the dependent chain, nothing else
typedef float scan_f4 __attribute__((ext_vector_type(4)));

// TE environments per workgroup x TC time steps per chunk.  32 x 64 (128-byte rows) when N gives >= 128 workgroups;
// below that 16 x 128: the chain is latency-bound whatever the number of active lanes, and twice as many workgroups
// (CUs) pull the rollout from HBM -- at T = 2048 x N = 1024 the 32 workgroups of the 32-wide form were load-bound
// (44.5 us; 16 x 128: 33.2 us = 16 ns per step, 8 x 128: 32.6 us -- the dependent chain of 3 roundings per step is the
// floor now; at N = 65536 the narrow forms lose 35-60 % of the bandwidth, so they are used for small N only).
template <bool GAE, bool PROPER, int TE_, int TC_>
struct ScanCfg {
    static constexpr int TE = TE_, TC = TC_;
    static constexpr int EPT = TE * TC / SCAN_LOADERS;       // elements per loader thread per chunk (8)
    static constexpr int PITCH = TC + 4;                     // LDS row (one environment's chunk) pitch, 16-byte multiple
    static constexpr int NARR = PROPER ? 4 : (GAE ? 3 : 2);
    static constexpr int IN_FLOATS = 2 * NARR * TE * PITCH;  // double-buffered chain operands
    static constexpr int OUT_FLOATS = 2 * TE * PITCH;        // double-buffered results
    static constexpr int LDS_BYTES = (IN_FLOATS + OUT_FLOATS) * (int)sizeof(float);
    static_assert(EPT % 4 == 0 && TE % 4 == 0 && (TE * TC) % SCAN_LOADERS == 0 && TE <= 64, "scan tile");
};

// Roles inside a workgroup, per 64-step chunk c (all overlapped, one barrier per chunk):
//   wave 0      : ONLY the dependent chain of chunk c (raised priority), operands and results through LDS as 16-byte
//                 vectors laid out time-contiguous per environment (one ds_read_b128 = 4 steps of one operand), fetched
//                 two groups ahead of their use
//   waves 1..4  : request chunk c+3 from HBM into registers (branch-free; THREE chunks = 120 loads per thread in flight, so
//                 a load has two full chunk periods to land before it is staged), write the results of chunk c-1 to HBM (coalesced 128-byte
//                 rows, straight-line predicated stores), pre-compute chunk c+1's carry-independent terms into the other
//                 LDS buffer
// (Round 1 had wave 0 share the load / staging duty and a run-time write-back loop whose back edge drained the
// prefetch: 31 ns per step at T = 2048 x N = 1024; the chain alone is ~10 ns.)
// VEC = 4 (N % 4 == 0, 16-byte aligned arrays): a loader thread's 8 elements per chunk are two quads of adjacent
// environments, one 16-byte load per array and quad -- 10 loads per chunk instead of 40, so three chunks of prefetch fit
// under the 64 outstanding vector-memory operations a wave can have (with 40 per chunk the issue itself stalled and the
// loaders, not the chain, set the pace: 52 us against 28 us for the chain alone at T = 2048).  VEC = 1: any N.
template <bool GAE, bool PROPER, int VEC, int TE, int TC>
__global__ __launch_bounds__(SCAN_THREADS) void returns_scan_kernel(
    const float* rewards, float* value_preds, float* returns, const float* masks, const float* bad_masks,
    const float* next_value, int T, int N, float g, float gl) {
    using Cfg = ScanCfg<GAE, PROPER, TE, TC>;
    constexpr int NARR = Cfg::NARR, SCAN_TE = TE, SCAN_TC = TC, SCAN_EPT = Cfg::EPT, SCAN_PITCH = Cfg::PITCH;
    extern __shared__ __attribute__((aligned(16))) float scan_lds[];
    float* lin = scan_lds;                          // [2][NARR][SCAN_TE][SCAN_PITCH]
    float* lout = scan_lds + Cfg::IN_FLOATS;        // [2][SCAN_TE][SCAN_PITCH]
    auto in_at = [&](int buf, int k, int le) { return lin + ((buf * NARR + k) * SCAN_TE + le) * SCAN_PITCH; };
    auto out_at = [&](int buf, int le) { return lout + (buf * SCAN_TE + le) * SCAN_PITCH; };

    const int tid = threadIdx.x;
    const int lt = tid - 64;                        // loader thread index (>= 0 for waves 1..4)
    const int e0 = blockIdx.x * SCAN_TE;
    const int nchunks = (T + SCAN_TC - 1) / SCAN_TC;

    // side effect of the reference: value_preds[-1] = next_value (GAE) / returns[-1] = next_value
    if (tid < SCAN_TE && e0 + tid < N) {
        const float nv = next_value[e0 + tid];
        if (GAE) value_preds[(long)T * N + e0 + tid] = nv;
        else returns[(long)T * N + e0 + tid] = nv;
    }

    float st[3][SCAN_EPT][5];                       // three chunks of loads in flight (set = chunk % 3)
    // element i of this loader thread -> (time row j of the chunk, local environment le)
    auto el_j = [&](int i) { return VEC == 4 ? (lt + SCAN_LOADERS * (i >> 2)) / (SCAN_TE / 4) : (lt + SCAN_LOADERS * i) / SCAN_TE; };
    auto el_le = [&](int i) {
        return VEC == 4 ? ((lt + SCAN_LOADERS * (i >> 2)) % (SCAN_TE / 4)) * 4 + (i & 3) : (lt + SCAN_LOADERS * i) % SCAN_TE;
    };

    // branch-free: out-of-range (t < 0 on the last chunk, n >= N on the last workgroup, chunks past the end) elements are
    // fetched from a clamped, valid address and never stored, so the loads of a chunk are issued back to back
    auto issue_loads = [&](auto P, int c) {
        constexpr int p = decltype(P)::value;
        const int cc = c < nchunks ? c : nchunks - 1;
        const int tlo = T - (cc + 1) * SCAN_TC;
        if constexpr (VEC == 4) {
#pragma unroll
            for (int u = 0; u < SCAN_EPT / 4; ++u) {
                const int j = el_j(4 * u), le = el_le(4 * u);
                const int t = (tlo + j) < 0 ? 0 : tlo + j;
                const int n = (e0 + le) < N ? e0 + le : N - 4;           // N % 4 == 0: a quad is wholly in or out
                const long o = (long)t * N + n;
                auto put = [&](int k, const float* base) {
                    const scan_f4 q = *reinterpret_cast<const scan_f4*>(base);
#pragma unroll
                    for (int v = 0; v < 4; ++v) st[p][4 * u + v][k] = q[v];
                };
                put(0, rewards + o);
                put(1, masks + o + N);
                if (GAE || PROPER) put(2, value_preds + o);
                if (GAE) put(3, value_preds + o + N);
                if (PROPER) put(4, bad_masks + o + N);
            }
        } else {
#pragma unroll
            for (int i = 0; i < SCAN_EPT; ++i) {
                const int j = el_j(i), le = el_le(i);
                const int t = (tlo + j) < 0 ? 0 : tlo + j;
                const int n = (e0 + le) < N ? e0 + le : N - 1;
                const long o = (long)t * N + n;
                st[p][i][0] = rewards[o];
                st[p][i][1] = masks[o + N];                                  // masks[t+1]
                if (GAE || PROPER) st[p][i][2] = value_preds[o];             // V[t]
                if (GAE) st[p][i][3] = value_preds[o + N];                   // V[t+1]  (t = T-1 is patched for chunk 0)
                if (PROPER) st[p][i][4] = bad_masks[o + N];                  // bad_masks[t+1]
            }
        }
    };
    auto stage_to_lds = [&](auto P, int buf) {
        constexpr int p = decltype(P)::value;
#pragma unroll
        for (int i = 0; i < SCAN_EPT; ++i) {
            const int j = el_j(i), le = el_le(i);
            const float r = st[p][i][0], m = st[p][i][1];
            if (GAE) {
                // delta = (r + (g*V[t+1])*m) - V[t]          storage.py:77-79 / 95-97
                const float delta = ((((r) + (((((g) * (st[p][i][3]))) * (m))))) - (st[p][i][2]));
                in_at(buf, 0, le)[j] = delta;
                in_at(buf, 1, le)[j] = ((gl) * (m));                 // (gamma*lambda)*mask
                in_at(buf, 2, le)[j] = st[p][i][2];
                if (PROPER) in_at(buf, 3, le)[j] = st[p][i][4];
            } else {
                in_at(buf, 0, le)[j] = m;
                in_at(buf, 1, le)[j] = r;
                if (PROPER) {
                    in_at(buf, 2, le)[j] = st[p][i][4];
                    in_at(buf, 3, le)[j] = ((((1.0f) - (st[p][i][4]))) * (st[p][i][2]));   // (1-b)*V[t]
                }
            }
        }
    };
    // results of chunk c (LDS) -> returns rows: a fixed number of predicated stores per loader thread (no loop back edge
    // for the memory-counter bookkeeping to drain the prefetch at)
    auto flush = [&](int c) {
        const int buf = c & 1;
        const int tlo = T - (c + 1) * SCAN_TC;
        // GAE: the chain leaves gae[t]; R = gae + V[t] (storage.py:83 / 100) is taken here, off the chain wave.  V[t] sits in
        // the operand buffer of the same parity, at the very slot this thread overwrites right after (stage_to_lds of
        // chunk c+2 uses the same element mapping), so the read cannot race with another thread's write.
        float v[SCAN_EPT];
#pragma unroll
        for (int i = 0; i < SCAN_EPT; ++i) {
            const int j = el_j(i), le = el_le(i);
            v[i] = out_at(buf, le)[j];
            if (GAE) v[i] = ((v[i]) + (in_at(buf, 2, le)[j]));
        }
        if constexpr (VEC == 4) {
#pragma unroll
            for (int u = 0; u < SCAN_EPT / 4; ++u) {
                const int j = el_j(4 * u), le = el_le(4 * u);
                if (tlo + j >= 0 && e0 + le < N)
                    *reinterpret_cast<scan_f4*>(returns + (long)(tlo + j) * N + e0 + le) =
                        scan_f4{v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]};
            }
        } else {
#pragma unroll
            for (int i = 0; i < SCAN_EPT; ++i) {
                const int j = el_j(i), le = el_le(i);
                if (tlo + j >= 0 && e0 + le < N) returns[(long)(tlo + j) * N + e0 + le] = v[i];
            }
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;

    float carry = 0.f;   // gae (GAE) or returns[t+1] (RET): lanes 0..31 of wave 0
    if (tid >= 64) {
        issue_loads(P0{}, 0);
        issue_loads(P1{}, 1);
        issue_loads(P2{}, 2);
        if (GAE) {      // V[T]: the reference reads next_value there (written above, possibly by another lane: use the source)
#pragma unroll
            for (int i = 0; i < SCAN_EPT; ++i) {
                const int j = el_j(i), le = el_le(i);
                const int n = (e0 + le) < N ? e0 + le : N - 1;
                if (j == SCAN_TC - 1) st[0][i][3] = next_value[n];
            }
        }
        stage_to_lds(P0{}, 0);
    } else {
        __builtin_amdgcn_s_setprio(3);
        if (!GAE && tid < SCAN_TE) carry = next_value[(e0 + tid) < N ? e0 + tid : N - 1];
    }
    __syncthreads();

    // chain of chunk c by lanes 0..31 of wave 0: operands fetched two 4-step groups ahead
    auto chain = [&](int buf) {
        const float* p0 = in_at(buf, 0, tid);
        const float* p1 = in_at(buf, 1, tid);
        const float* p2 = in_at(buf, NARR > 2 ? 2 : 0, tid);
        const float* p3 = in_at(buf, NARR > 3 ? 3 : 0, tid);
        float* po = out_at(buf, tid);
        scan_f4 a0[3], a1[3], a2[3], a3[3];
        auto fetch = [&](int set, int j0) {
            a0[set] = *reinterpret_cast<const scan_f4*>(p0 + j0);
            a1[set] = *reinterpret_cast<const scan_f4*>(p1 + j0);
            if (NARR > 2 && !GAE) a2[set] = *reinterpret_cast<const scan_f4*>(p2 + j0);   // GAE: V[t] is added at write-back
            if (NARR > 3) a3[set] = *reinterpret_cast<const scan_f4*>(p3 + j0);
        };
        constexpr int NG = SCAN_TC / 4;
        fetch((NG - 1) % 3, 4 * (NG - 1));
        fetch((NG - 2) % 3, 4 * (NG - 2));
#pragma unroll
        for (int q = NG - 1; q >= 0; --q) {
            const int set = q % 3;
            if (q >= 2) fetch((q - 2) % 3, 4 * (q - 2));
            // keep the fetch two groups ahead of its use: without the fence the scheduler sinks the ds_reads next to their
            // consumers and the ~50-cycle LDS latency lands on the dependent chain once per group
            __builtin_amdgcn_sched_barrier(0);
            scan_f4 res;
#pragma unroll
            for (int u = 3; u >= 0; --u) {
                if (GAE) {
                    carry = ((a0[set][u]) + (((a1[set][u]) * (carry))));     // gae = delta + (gl*m)*gae
                    if (PROPER) carry = ((carry) * (a3[set][u]));            // gae *= bad_mask
                    res[u] = carry;                                          // (R = gae + V[t]: see flush)
                } else {
                    carry = ((((((carry) * (g))) * (a0[set][u]))) + (a1[set][u]));   // (R*g)*m + r
                    if (PROPER) carry = ((((carry) * (a2[set][u]))) + (a3[set][u]));
                    res[u] = carry;
                }
            }
            *reinterpret_cast<scan_f4*>(po + 4 * q) = res;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // loaders at chunk c (set P = c % 3): request chunk c+3 into set P (chunk c's values were staged one chunk ago), write
    // chunk c-1's results out, stage chunk c+1 (set (c+1) % 3, requested two chunk periods ago) into the other LDS buffer
    auto loader_step = [&](auto P, int c) {
        constexpr int p = decltype(P)::value;
        issue_loads(P, c + 3);
        if (c > 0) flush(c - 1);
        stage_to_lds(std::integral_constant<int, (p + 1) % 3>{}, (c + 1) & 1);
    };
    auto step = [&](auto P, int c) {
        if (tid >= 64) loader_step(P, c);
        else if (tid < SCAN_TE) chain(c & 1);
        __syncthreads();
    };
    for (int c = 0; c < nchunks; c += 3) {
        step(P0{}, c);
        if (c + 1 < nchunks) step(P1{}, c + 1);
        if (c + 2 < nchunks) step(P2{}, c + 2);
    }
    if (tid >= 64) flush(nchunks - 1);
}

// ---------------------------------------------------------------------------
// Advantage normalisation (ppo.py:35-37): adv = R - V; (adv - mean)/(std_unbiased + 1e-5)
// pass 1: per-block fp64 partial (sum, sum of squares); pass 2: every block re-reduces the
// partials in a fixed order (deterministic) and normalises its slice.
// ---------------------------------------------------------------------------
constexpr int ADV_THREADS = 256;
constexpr int ADV_MAX_BLOCKS = 1024;

__global__ __launch_bounds__(ADV_THREADS) void adv_partial_kernel(const float* returns, const float* value_preds,
                                                                 long n, double* partial) {
    __shared__ double sm[16];
    double s = 0.0, s2 = 0.0;
    for (long i = (long)blockIdx.x * ADV_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * ADV_THREADS) {
        const float a = ((returns[i]) - (value_preds[i]));
        s += (double)a;
        s2 += (double)a * (double)a;
    }
    s = block_sum(s, sm);
    s2 = block_sum(s2, sm);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = s;
        partial[2 * blockIdx.x + 1] = s2;
    }
}

__global__ __launch_bounds__(ADV_THREADS) void adv_normalize_kernel(const float* returns, const float* value_preds,
                                                                   float* adv, long n, const double* partial,
                                                                   int nparts) {
    __shared__ float ms[2];
    if (threadIdx.x < 64) {
        double s = 0.0, s2 = 0.0;
        for (int i = threadIdx.x; i < nparts; i += 64) {
            s += partial[2 * i];
            s2 += partial[2 * i + 1];
        }
        s = wave_sum(s);
        s2 = wave_sum(s2);
        if (threadIdx.x == 0) {
            const double mean = s / (double)n;
            double var = (s2 - s * mean) / (double)(n - 1);   // unbiased
            if (var < 0.0) var = 0.0;
            ms[0] = (float)mean;
            ms[1] = (float)sqrt(var);
        }
    }
    __syncthreads();
    const float mean = ms[0];
    const float denom = ((ms[1]) + (1e-5f));
    for (long i = (long)blockIdx.x * ADV_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * ADV_THREADS) {
        const float a = ((returns[i]) - (value_preds[i]));
        adv[i] = ((((a) - (mean))) / (denom));
    }
}

// ---------------------------------------------------------------------------
// Row gather (storage.py:127-140): dst[i, :] = src[idx[i], :]
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* src, const int64_t* idx, T* dst, long rows,
                                                          long row_elems, long src_rows) {
    const long row = blockIdx.x;
    const long s = idx[row];
    if (s < 0 || s >= src_rows) return;   // host validates; never fault
    const T* sp = src + s * row_elems;
    T* dp = dst + row * row_elems;
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < row_elems; i += (long)gridDim.y * 256) dp[i] = sp[i];
}

}  // namespace acktr

using namespace acktr;

extern "C" int acktr_compute_returns_f32(const float* rewards, float* value_preds, float* returns,
                                         const float* masks, const float* bad_masks, const float* next_value,
                                         int T, int N, float gamma32, float gamma_lambda32, int use_gae,
                                         int use_proper_time_limits, void* stream) {
    ACKTR_ARG(rewards && value_preds && returns && masks && next_value, "compute_returns: null pointer");
    ACKTR_ARG(!use_proper_time_limits || bad_masks, "compute_returns: bad_masks required");
    ACKTR_ARG(T > 0 && N > 0, "compute_returns: T=%d N=%d must be positive", T, N);
    ACKTR_ARG((long)(T + 1) * N < (1L << 31), "compute_returns: (T+1)*N too large");
    hipStream_t s = (hipStream_t)stream;
    dim3 block(SCAN_THREADS);
    int te = (cdiv(N, 32) < 128 && N > 16) ? 16 : 32;         // few environments: 16 per workgroup, 128-step chunks
    if (const char* ev = getenv("ACKTR_SCAN_TE")) te = atoi(ev);   // tuning override (tools/scan_bench.py)
    const uintptr_t al = (uintptr_t)rewards | (uintptr_t)value_preds | (uintptr_t)returns | (uintptr_t)masks |
                         (uintptr_t)(use_proper_time_limits ? bad_masks : rewards);
    const bool vec4 = (N % 4 == 0) && (al % 16 == 0);
#define LAUNCH(G, P, V, TE, TC)                                                                                      \
    do {                                                                                                             \
        constexpr int lds_bytes = ScanCfg<G, P, TE, TC>::LDS_BYTES;                                                  \
        static bool attr_set = false;                                                                                \
        if (!attr_set && lds_bytes > 48 * 1024) {                                                                    \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(returns_scan_kernel<G, P, V, TE, TC>),  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);               \
            ACKTR_ARG(e == hipSuccess, "compute_returns: hipFuncSetAttribute failed: %s", hipGetErrorString(e));     \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((returns_scan_kernel<G, P, V, TE, TC>), dim3(cdiv(N, TE)), block, lds_bytes, s, rewards,  \
                           value_preds, returns, masks, bad_masks, next_value, T, N, gamma32, gamma_lambda32);       \
    } while (0)
#define LAUNCHV(G, P)                                                             \
    do {                                                                          \
        if (te == 16) { if (vec4) LAUNCH(G, P, 4, 16, 128); else LAUNCH(G, P, 1, 16, 128); } \
        else { if (vec4) LAUNCH(G, P, 4, 32, 64); else LAUNCH(G, P, 1, 32, 64); }  \
    } while (0)
    if (use_gae) {
        if (use_proper_time_limits) LAUNCHV(true, true); else LAUNCHV(true, false);
    } else {
        if (use_proper_time_limits) LAUNCHV(false, true); else LAUNCHV(false, false);
    }
#undef LAUNCHV
#undef LAUNCH
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" long acktr_adv_normalize_workspace_bytes(void) { return (long)ADV_MAX_BLOCKS * 2 * sizeof(double); }

namespace acktr {
static int adv_normalize_launch(const float* returns, const float* value_preds, float* adv, long n, double* partial,
                                hipStream_t s) {
    int blocks = cdiv(n, ADV_THREADS * 8);
    if (blocks > ADV_MAX_BLOCKS) blocks = ADV_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adv_partial_kernel, dim3(blocks), dim3(ADV_THREADS), 0, s, returns, value_preds, n, partial);
    ACKTR_LAUNCH_CHECK();
    int nblocks2 = cdiv(n, ADV_THREADS * 4);
    if (nblocks2 > 2048) nblocks2 = 2048;
    hipLaunchKernelGGL(adv_normalize_kernel, dim3(nblocks2), dim3(ADV_THREADS), 0, s, returns, value_preds, adv, n,
                       (const double*)partial, blocks);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

// Scratch of the allocation-free entry point: the (sum, sum of squares) partials of at most ADV_MAX_BLOCKS blocks,
// 16 KiB per GPU, resident for the life of the process.  The name is a host-side stand-in: host code only ever uses
// the per-device address that hipGetSymbolAddress returns (each GPU has its own copy at its own address).
static __device__ __align__(256) double adv_noalloc_scratch[ADV_MAX_BLOCKS * 2];
constexpr int NOALLOC_MAX_DEVICES = 64;
static std::mutex g_noalloc_mu;
static double* g_noalloc_ptr[NOALLOC_MAX_DEVICES] = {nullptr};

// The scratch address of the CURRENT device.  `s` (nullable): the stream the kernels will be launched on -- it must
// belong to the current device, or the launch would carry another GPU's address.
static int adv_noalloc_scratch_ptr(double** out, hipStream_t s) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { set_error("adv_normalize_noalloc: hipGetDevice failed: %s", hipGetErrorString(e)); return (int)e; }
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) { set_error("adv_normalize_noalloc: device %d out of range", dev); return -1; }
    if (s) {
        hipDevice_t sdev = -1;
        e = hipStreamGetDevice(s, &sdev);
        if (e == hipSuccess && (int)sdev != dev) {
            set_error("adv_normalize_noalloc: the stream belongs to device %d but device %d is current; make the stream's "
                      "device current before the call", (int)sdev, dev);
            return -1;
        }
    }
    std::lock_guard<std::mutex> lk(g_noalloc_mu);
    if (!g_noalloc_ptr[dev]) {                       // once per device, with that device current
        double* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(adv_noalloc_scratch));
        if (e != hipSuccess || !p) {
            set_error("adv_normalize_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return e != hipSuccess ? (int)e : -1;
        }
        g_noalloc_ptr[dev] = p;
    }
    *out = g_noalloc_ptr[dev];
    return 0;
}
}  // namespace acktr

extern "C" int acktr_adv_normalize_f32(const float* returns, const float* value_preds, float* adv, long n,
                                       void* workspace, void* stream) {
    ACKTR_ARG(returns && value_preds && adv && workspace, "adv_normalize: null pointer");
    ACKTR_ARG(n >= 2, "adv_normalize: need n >= 2 for the unbiased std (n=%ld)", n);
    return adv_normalize_launch(returns, value_preds, adv, n, (double*)workspace, (hipStream_t)stream);
}

extern "C" long acktr_adv_normalize_noalloc_max_n(void) { return 1L << 28; }

extern "C" int acktr_adv_normalize_noalloc_init(void) {
    double* p = nullptr;
    return adv_noalloc_scratch_ptr(&p, nullptr);
}

extern "C" int acktr_adv_normalize_noalloc_f32(const float* returns, const float* value_preds, float* adv, long n,
                                               void* stream) {
    ACKTR_ARG(returns && value_preds && adv, "adv_normalize_noalloc: null pointer");
    ACKTR_ARG(n >= 2, "adv_normalize_noalloc: need n >= 2 for the unbiased std (n=%ld)", n);
    ACKTR_ARG(n <= acktr_adv_normalize_noalloc_max_n(),
              "adv_normalize_noalloc: n=%ld exceeds the maximum of %ld elements of this allocation-free entry point; "
              "use acktr_adv_normalize_f32 with a caller workspace", n, acktr_adv_normalize_noalloc_max_n());
    double* partial = nullptr;
    int rc = adv_noalloc_scratch_ptr(&partial, (hipStream_t)stream);
    if (rc) return rc;
    // every partial a launch reads was written by the same call's first kernel: nothing has to start at zero
    return adv_normalize_launch(returns, value_preds, adv, n, partial, (hipStream_t)stream);
}

extern "C" int acktr_gather_rows(const void* src, const int64_t* idx, void* dst, long rows, long row_bytes,
                                 long src_rows, void* stream) {
    ACKTR_ARG(src && idx && dst, "gather_rows: null pointer");
    ACKTR_ARG(rows >= 0 && row_bytes > 0 && src_rows > 0, "gather_rows: bad sizes");
    if (rows == 0) return 0;
    ACKTR_ARG(rows < (1L << 31), "gather_rows: too many rows");
    hipStream_t s = (hipStream_t)stream;
    const bool v16 = (row_bytes % 16 == 0) && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    if (v16) {
        const long ne = row_bytes / 16;
        int bx = cdiv(ne, 256 * 4); if (bx < 1) bx = 1; if (bx > 64) bx = 64;
        hipLaunchKernelGGL((gather_rows_kernel<float4>), dim3((unsigned)rows, bx), dim3(256), 0, s,
                           (const float4*)src, idx, (float4*)dst, rows, ne, src_rows);
    } else {
        ACKTR_ARG(row_bytes % 4 == 0, "gather_rows: row_bytes must be a multiple of 4");
        const long ne = row_bytes / 4;
        int bx = cdiv(ne, 256 * 4); if (bx < 1) bx = 1; if (bx > 64) bx = 64;
        hipLaunchKernelGGL((gather_rows_kernel<float>), dim3((unsigned)rows, bx), dim3(256), 0, s,
                           (const float*)src, idx, (float*)dst, rows, ne, src_rows);
    }
    ACKTR_LAUNCH_CHECK();
    return 0;
}
