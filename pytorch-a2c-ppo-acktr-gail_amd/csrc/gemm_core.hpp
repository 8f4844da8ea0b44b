// fp32 MFMA tile-GEMM skeleton with pluggable operand loaders and epilogues (gfx950).
//
//   C(m, n) = sum_k  Mop(m, k) * Nop(n, k)          m: 32x32 tile rows (accumulator registers)
//                                                   n: 32x32 tile cols (lanes -> coalesced stores)
//
// Every dense contraction of the update path (implicit-GEMM convolution forward / dgrad /
// wgrad, linear layers, KFAC factor SYRKs, KFAC preconditioning) is this one skeleton with a
// different pair of *operand functors*: an operand never has to exist in memory as a matrix --
// im2col patches, transposed views, gathered minibatch rows are address functions evaluated by
// the loader, so no patch matrix is ever materialised in HBM.
//
// MI355X specifics
//   * v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD = the fp32 matrix peak);
//     64-wide waves, 4 waves / workgroup arranged WM x WN over the block tile.
//   * K-permutation trick: lane half h owns k = 8q+4h .. 8q+4h+3 of every 8-wide k group, so one
//     ds_read_b128 per operand feeds FOUR MFMA k-steps (both operands use the same permutation,
//     so products pair up correctly; only the fp32 summation order changes).
//   * LDS layouts follow the operand's memory contiguity so global loads stay 16 B wide:
//       INNER_IS_K  : [rows][32+4]   (k contiguous)  fragment = 1 x ds_read_b128, conflict free
//       !INNER_IS_K : [32][rows+4]   (row contiguous) fragment = 4 x ds_read_b32, conflict free
//   * register-staged double buffering: global loads of tile t+1 are issued before the MFMAs of
//     tile t and written to the other LDS stage afterwards; one barrier per K tile.
//   * optional split-K over blockIdx.z into per-slice slabs (deterministic fixed-order reduce).
#pragma once
#include <type_traits>
#include "common.hpp"

namespace acktr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GK = 32;        // K tile
constexpr int PK = GK + 4;    // pitch (floats) of the k-contiguous LDS layout

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// ------------------------------------------------------------------------------------------
// Operand tile: global -> registers -> LDS -> MFMA fragments
// Operand concept:
//   static constexpr bool INNER_IS_K;
//   struct Outer;  Outer prep(int outer) const;
//   struct Inner;  Inner prep_inner(int inner4) const;      // decode of the inner index; for row-contiguous operands
//                                                           // it is K-invariant and evaluated ONCE per thread
//   f32x4 ld4(const Outer&, const Inner&, int& meta) const; // RAW 4 consecutive "inner" elements from a safe address
//   static f32x4 post(const f32x4& raw, int meta);          // bounds select / shifts, applied when the registers are
//                                                           // committed to LDS -- never right behind the load, so that
//                                                           // the loads of a whole tile stay in flight together
//   INNER_IS_K : outer = row (m or n), inner = k        !INNER_IS_K : outer = k, inner = row
// ------------------------------------------------------------------------------------------
template <int BR, class Op>
struct OperandTile {
    static constexpr int UNITS = BR / 32;            // float4 units per thread per K tile
    static constexpr int PITCH_R = BR + 4;
    static constexpr int SMEM = Op::INNER_IS_K ? BR * PK : GK * PITCH_R;
    static constexpr int RQ = BR / 4;                // row-contiguous layout: float4 groups per k row
    // row-contiguous mapping: thread -> ONE k row (tid / 8) and UNITS row groups (tid % 8 + 8u), so the outer
    // state of that k (an im2col position decode + gather-index load) is prepared once per thread per K tile
    static_assert(RQ % 8 == 0, "row tile must be a multiple of 32");

    f32x4 regs[2][UNITS];                            // two tiles in flight (register-staged prefetch, depth 2)
    int meta[2][UNITS];
    typename Op::Outer outer[Op::INNER_IS_K ? UNITS : 1];
    typename Op::Inner inner[Op::INNER_IS_K ? 1 : UNITS];     // row-contiguous: the rows' decode never changes

    __device__ __forceinline__ void init(const Op& op, int row_base, int tid) {
        if constexpr (Op::INNER_IS_K) {
#pragma unroll
            for (int u = 0; u < UNITS; ++u) outer[u] = op.prep(row_base + (tid >> 3) + 32 * u);
        } else {
#pragma unroll
            for (int u = 0; u < UNITS; ++u) inner[u] = op.prep_inner(row_base + 4 * ((tid & 7) + 8 * u));
        }
    }
    template <int P>
    __device__ __forceinline__ void load(const Op& op, int row_base, int k_base, int tid) {
        if constexpr (Op::INNER_IS_K) {
            const typename Op::Inner in = op.prep_inner(k_base + 4 * (tid & 7));
#pragma unroll
            for (int u = 0; u < UNITS; ++u) regs[P][u] = op.ld4(outer[u], in, meta[P][u]);
        } else {
            const typename Op::Outer o = op.prep(k_base + (tid >> 3));
#pragma unroll
            for (int u = 0; u < UNITS; ++u) regs[P][u] = op.ld4(o, inner[u], meta[P][u]);
        }
    }
    template <int P>
    __device__ __forceinline__ void finish() {       // raw -> operand values (bounds select etc.)
#pragma unroll
        for (int u = 0; u < UNITS; ++u) regs[P][u] = Op::post(regs[P][u], meta[P][u]);
    }
    template <int P>
    __device__ __forceinline__ void store(float* s, int tid) const {
        if constexpr (Op::INNER_IS_K) {
#pragma unroll
            for (int u = 0; u < UNITS; ++u)
                *reinterpret_cast<f32x4*>(s + ((tid >> 3) + 32 * u) * PK + 4 * (tid & 7)) = regs[P][u];
        } else {
#pragma unroll
            for (int u = 0; u < UNITS; ++u)
                *reinterpret_cast<f32x4*>(s + (tid >> 3) * PITCH_R + 4 * ((tid & 7) + 8 * u)) = regs[P][u];
        }
    }
    // fragment for tile-local row `row`, k group base kk = 8q + 4h
    __device__ static __forceinline__ f32x4 frag(const float* s, int row, int kk) {
        if constexpr (Op::INNER_IS_K) {
            return *reinterpret_cast<const f32x4*>(s + row * PK + kk);
        } else {
            f32x4 v;
            v[0] = s[(kk + 0) * PITCH_R + row];
            v[1] = s[(kk + 1) * PITCH_R + row];
            v[2] = s[(kk + 2) * PITCH_R + row];
            v[3] = s[(kk + 3) * PITCH_R + row];
            return v;
        }
    }
};

// ------------------------------------------------------------------------------------------
// Epilogue concept:
//   struct Col;  Col col(int n) const;                       // per-lane column state (validity, offsets)
//   void st(const Col&, int m, float acc, int kslice) const; // bounds-checks m itself
//   static constexpr bool HAS_ROWSUM;  bool wants_rowsum() const;  void rowsum(int m, float v, int kslice) const;
//     optional: sum over k of the M operand's rows (the bias gradient of a wgrad GEMM, where the M operand is dY),
//     accumulated from the registers the loader holds anyway -- dY is not read a second time for it
// ------------------------------------------------------------------------------------------
// Symmetric outputs (C = X X^T, the KFAC factors): an epilogue may declare a tile redundant -- every element of it lies
// strictly below the diagonal; launch_symmetrize_lower() fills the lower triangle from the upper one at the end
// (mirroring inside the epilogue was tried first: its column-strided stores cost what the skipped tiles saved).
// Epilogues that declare GATE_PREFETCH offer pre(col, m) -> float and take that value as st()'s last argument.
template <class Epi, class = void>
struct EpiGatePrefetch : std::false_type {};
template <class Epi>
struct EpiGatePrefetch<Epi, std::enable_if_t<Epi::GATE_PREFETCH>> : std::true_type {};

// Batched launches (ZBATCH): an operand may carry its own reduction length per blockIdx.z (zK()); others use the launch's K.
template <class Op, class = void>
struct OpZK {
    __device__ static __forceinline__ int get(const Op&, int K) { return K; }
};
template <class Op>
struct OpZK<Op, std::void_t<decltype(&Op::zK)>> {
    __device__ static __forceinline__ int get(const Op& op, int) { return op.zK(); }
};

template <class Epi>
struct EpiSym {
    __device__ static __forceinline__ bool skip(const Epi&, int, int, int) { return false; }
};

template <int BM, int BN, int WM, int WN, class MOp, class NOp, class Epi, bool ZBATCH = false>
__global__ __launch_bounds__(256) void gemm_mfma_f32_kernel(const MOp mop, const NOp nop, const Epi epi, int K,
                                                            int kchunk) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "tile / wave layout");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    using MT = OperandTile<BM, MOp>;
    using NT = OperandTile<BN, NOp>;
    constexpr int STAGE = MT::SMEM + NT::SMEM;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    // Workgroups are dispatched round-robin over the 8 XCDs (each with its own L2).  Renumber them so that workgroups
    // that are neighbours in (x, y, z) -- same operand tiles, same K slice -- run on the SAME XCD.
    int vx = blockIdx.x, vy = blockIdx.y, vz = blockIdx.z;
    if constexpr (!ZBATCH) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        if (total % 8 == 0) {
            const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            const unsigned Lp = (L % 8) * (total / 8) + L / 8;
            vx = Lp % gridDim.x; vy = (Lp / gridDim.x) % gridDim.y; vz = Lp / (gridDim.x * gridDim.y);
        }
    }
    const int n0 = vx * BN, m0 = vy * BM;
    if (EpiSym<Epi>::skip(epi, m0, n0, BN)) return;        // workgroup-uniform, before any barrier
    const int kbeg = ZBATCH ? 0 : vz * kchunk;     // ZBATCH: z is a phase index the functors read themselves
    int kend = kbeg + kchunk;
    if (kend > K) kend = K;
    if constexpr (ZBATCH) kend = OpZK<MOp>::get(mop, kend);
    const int nk = (kend - kbeg + GK - 1) / GK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    MT mt;
    NT nt;
    mt.init(mop, m0, tid);
    nt.init(nop, n0, tid);

    bool do_rowsum = false;
    if constexpr (Epi::HAS_ROWSUM) do_rowsum = epi.wants_rowsum() && vx == 0;     // workgroup-uniform
    f32x4 rs[MT::UNITS];
#pragma unroll
    for (int u = 0; u < MT::UNITS; ++u) rs[u] = zero4();
    // Pipeline (depth 2): while tile kt is multiplied out of LDS, tile kt+1 sits in register set (kt+1)&1 waiting to
    // be stored and tile kt+2 is in flight into the other set.  With one tile of look-ahead every K step exposed a
    // full memory latency: the short reductions of this workload (a few K tiles per workgroup) ran latency-bound.
    // request tile t into register set P.  UNCONDITIONAL (past the end the last tile is simply requested again and
    // never committed): behind a branch the compiler cannot count the loads in flight and drains them all.
    auto issue = [&](auto P, int t) {
        const int tc = t < nk ? t : (nk > 0 ? nk - 1 : 0);       // (an empty K slice requests k >= K: masked, safe)
        mt.template load<decltype(P)::value>(mop, m0, kbeg + tc * GK, tid);
        nt.template load<decltype(P)::value>(nop, n0, kbeg + tc * GK, tid);
        __builtin_amdgcn_sched_barrier(0);
    };
    // register set P (tile t) -> LDS stage t&1 (+ bias row sums).  Also unconditional: past the end it stores the
    // re-requested last tile into a stage that nobody reads any more.
    auto commit = [&](auto P, int t) {
        mt.template finish<decltype(P)::value>();
        nt.template finish<decltype(P)::value>();
        if constexpr (Epi::HAS_ROWSUM) {
            if (do_rowsum && t < nk) {
#pragma unroll
                for (int u = 0; u < MT::UNITS; ++u) rs[u] += mt.regs[decltype(P)::value][u];
            }
        }
        float* d = smem + (t & 1) * STAGE;
        mt.template store<decltype(P)::value>(d, tid);
        nt.template store<decltype(P)::value>(d + MT::SMEM, tid);
    };
    auto multiply = [&](int kt) {
        const float* sM = smem + (kt & 1) * STAGE;
        const float* sN = sM + MT::SMEM;
        // fragments are fetched from LDS one k group ahead of the MFMAs that use them
        f32x4 af[2][TM], bf[2][TN];
        auto load_frags = [&](int q) {
            const int kk = 8 * q + 4 * lh;
#pragma unroll
            for (int a = 0; a < TM; ++a) af[q & 1][a] = MT::frag(sM, wm * (TM * 32) + a * 32 + li, kk);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[q & 1][b] = NT::frag(sN, wn * (TN * 32) + b * 32 + li, kk);
        };
        load_frags(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < GK / 8; ++q) {
            if (q + 1 < GK / 8) load_frags(q + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][a][t], bf[q & 1][b][t], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    issue(P0{}, 0);
    issue(P1{}, 1);
    commit(P0{}, 0);
    issue(P0{}, 2);
    __syncthreads();

    // the loop body is one straight line (no exits, no conditional memory operations), so the compiler's count of
    // outstanding loads stays exact across the back edge
    const int npairs = nk / 2;
    for (int i = 0; i < npairs; ++i) {
        const int kt = 2 * i;
        multiply(kt);                 // set 1 holds tile kt+1, set 0 is receiving tile kt+2
        commit(P1{}, kt + 1);
        issue(P1{}, kt + 3);
        __syncthreads();
        multiply(kt + 1);             // set 0 holds tile kt+2, set 1 is receiving tile kt+3
        commit(P0{}, kt + 2);
        issue(P0{}, kt + 4);
        __syncthreads();
    }
    if (nk & 1) multiply(nk - 1);
    __syncthreads();                  // LDS is reused by the row-sum reduction below

    if constexpr (Epi::HAS_ROWSUM) {
        if (do_rowsum) {
            if constexpr (MOp::INNER_IS_K) {     // thread holds 4 k of row (tid>>3)+32u; the row's 8 threads are adjacent lanes
#pragma unroll
                for (int u = 0; u < MT::UNITS; ++u) {
                    float v = (rs[u][0] + rs[u][1]) + (rs[u][2] + rs[u][3]);
                    v += __shfl_xor(v, 1);
                    v += __shfl_xor(v, 2);
                    v += __shfl_xor(v, 4);
                    if ((tid & 7) == 0) epi.rowsum(m0 + (tid >> 3) + 32 * u, v, vz);
                }
            } else {                             // thread holds rows 4((tid&7)+8u)..+3 at k = tid>>3: reduce over k through LDS
                float* red = smem;               // the main loop ended on a barrier: LDS is free
#pragma unroll
                for (int u = 0; u < MT::UNITS; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) red[(tid >> 3) * BM + 4 * ((tid & 7) + 8 * u) + j] = rs[u][j];
                __syncthreads();
                if (tid < BM) {
                    float v = 0.f;
#pragma unroll 8
                    for (int k = 0; k < 32; ++k) v += red[k * BM + tid];
                    epi.rowsum(m0 + tid, v, vz);
                }
            }
        }
    }

    // epilogue: lane -> column n, registers -> rows m
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + wn * (TN * 32) + b * 32 + li;
        const typename Epi::Col col = epi.col(n);
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int mb = m0 + wm * (TM * 32) + a * 32 + 4 * lh;
            if constexpr (EpiGatePrefetch<Epi>::value && TM * TN <= 2) {
                float h[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) h[r] = epi.pre(col, mb + (r & 3) + 8 * (r >> 2));
#pragma unroll
                for (int r = 0; r < 16; ++r) epi.st(col, mb + (r & 3) + 8 * (r >> 2), acc[a][b][r], vz, h[r]);
            } else if constexpr (EpiGatePrefetch<Epi>::value) {
                // 128x128 tiles (the KFAC factor SYRKs): the unrolled epilogue would keep all 64 prefetched values live
                // on top of 64 accumulators -- 276 VGPRs, one wave per SIMD instead of two.  Load at the point of use.
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    epi.st(col, m, acc[a][b][r], vz, epi.pre(col, m));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) epi.st(col, mb + (r & 3) + 8 * (r >> 2), acc[a][b][r], vz);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, class MOp, class NOp, class Epi>
int launch_gemm(const MOp& mop, const NOp& nop, const Epi& epi, int M, int N, int K, int nslices,
                hipStream_t stream) {
    using MT = OperandTile<BM, MOp>;
    using NT = OperandTile<BN, NOp>;
    constexpr int smem_bytes = 2 * (MT::SMEM + NT::SMEM) * (int)sizeof(float);
    static_assert(smem_bytes <= 160 * 1024, "LDS budget");
    if (M <= 0 || N <= 0) return 0;
    if (nslices < 1) nslices = 1;
    int kchunk = (K + nslices - 1) / nslices;
    kchunk = ((kchunk + GK - 1) / GK) * GK;
    if (kchunk < GK) kchunk = GK;
    auto kern = gemm_mfma_f32_kernel<BM, BN, WM, WN, MOp, NOp, Epi>;
    if (smem_bytes > 48 * 1024) {
        static bool attr_set = false;   // per instantiation
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
            if (e != hipSuccess) {
                set_error("hipFuncSetAttribute(%d B LDS) failed: %s", smem_bytes, hipGetErrorString(e));
                return (int)e;
            }
            attr_set = true;
        }
    }
    dim3 grid(cdiv(N, BN), cdiv(M, BM), nslices);
    if (grid.y > 65535 || grid.z > 65535) {
        set_error("gemm grid too large (%u,%u,%u)", grid.x, grid.y, grid.z);
        return -1;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), smem_bytes, stream, mop, nop, epi, K, kchunk);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

// Same skeleton with blockIdx.z as a batch / phase index (functors read blockIdx.z); K is not split.
template <int BM, int BN, int WM, int WN, class MOp, class NOp, class Epi>
int launch_gemm_z(const MOp& mop, const NOp& nop, const Epi& epi, int M, int N, int K, int nz, hipStream_t stream) {
    using MT = OperandTile<BM, MOp>;
    using NT = OperandTile<BN, NOp>;
    constexpr int smem_bytes = 2 * (MT::SMEM + NT::SMEM) * (int)sizeof(float);
    if (M <= 0 || N <= 0 || nz <= 0) return 0;
    int kchunk = ((K + GK - 1) / GK) * GK;
    auto kern = gemm_mfma_f32_kernel<BM, BN, WM, WN, MOp, NOp, Epi, true>;
    if (smem_bytes > 48 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
            if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
            attr_set = true;
        }
    }
    dim3 grid(cdiv(N, BN), cdiv(M, BM), nz);
    if (grid.y > 65535 || grid.z > 65535) { set_error("gemm grid too large"); return -1; }
    hipLaunchKernelGGL(kern, grid, dim3(256), smem_bytes, stream, mop, nop, epi, K, kchunk);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

// ==========================================================================================
// Dense operands
// ==========================================================================================
// Loader rule: the common path of every operand is BRANCH-FREE -- one unconditional 16-byte load from a clamped
// (always valid) address; the bounds predicate travels in `meta` and is applied by post() at LDS-store time.
// Loads under per-lane branches end up in separate basic blocks (they cannot be batched, and the compiler drains
// the memory counter at every join); a select right behind the load makes the wave wait for it on the spot.
__device__ __forceinline__ f32x4 sel4(bool ok, const f32x4& v) {
    f32x4 r;
    r[0] = ok ? v[0] : 0.f; r[1] = ok ? v[1] : 0.f; r[2] = ok ? v[2] : 0.f; r[3] = ok ? v[3] : 0.f;
    return r;
}

// element (row, k) at p[row*ld + k]  (k contiguous).  VEC: p 16-byte aligned, ld % 4 == 0, K % 4 == 0.
template <bool VEC>
struct DenseKCT {
    static constexpr bool INNER_IS_K = true;
    const float* p; int rows; int K; long ld;
    struct Outer { const float* rp; int ok; };
    __device__ __forceinline__ Outer prep(int row) const {
        Outer o; o.ok = row < rows; o.rp = p + (long)(o.ok ? row : 0) * ld; return o;
    }
    struct Inner { int k0; };
    __device__ __forceinline__ Inner prep_inner(int k0) const { return Inner{k0}; }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, const Inner& in, int& meta) const {
        const int k0 = in.k0;
        if constexpr (VEC) {
            const bool ok = o.ok && k0 < K;
            meta = ok;
            return *reinterpret_cast<const f32x4*>(o.rp + (ok ? k0 : 0));
        } else {
            meta = 1;
            f32x4 v = zero4();
            if (!o.ok || k0 >= K) return v;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (k0 + j < K) v[j] = o.rp[k0 + j];
            return v;
        }
    }
    __device__ static __forceinline__ f32x4 post(const f32x4& v, int meta) { return sel4(meta & 1, v); }
};
// element (row, k) at p[k*ld + row]  (row contiguous).  VEC: p 16-byte aligned, ld % 4 == 0, rows % 4 == 0.
template <bool VEC>
struct DenseRCT {
    static constexpr bool INNER_IS_K = false;
    const float* p; int rows; int K; long ld;
    struct Outer { const float* kp; int ok; };
    __device__ __forceinline__ Outer prep(int k) const {
        Outer o; o.ok = k < K; o.kp = p + (long)(o.ok ? k : 0) * ld; return o;
    }
    struct Inner { int r0; };
    __device__ __forceinline__ Inner prep_inner(int r0) const { return Inner{r0}; }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, const Inner& in, int& meta) const {
        const int r0 = in.r0;
        if constexpr (VEC) {
            const bool ok = o.ok && r0 < rows;
            meta = ok;
            return *reinterpret_cast<const f32x4*>(o.kp + (ok ? r0 : 0));
        } else {
            meta = 1;
            f32x4 v = zero4();
            if (!o.ok || r0 >= rows) return v;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (r0 + j < rows) v[j] = o.kp[r0 + j];
            return v;
        }
    }
    __device__ static __forceinline__ f32x4 post(const f32x4& v, int meta) { return sel4(meta & 1, v); }
};
using DenseKC = DenseKCT<true>;      // the conv / KFAC call sites use the vector forms (their shapes qualify)
using DenseRC = DenseRCT<true>;
static inline int vec_ok(const float* p, long ld) { return (((uintptr_t)p & 15) == 0) && (ld % 4 == 0); }

// ==========================================================================================
// Row-major epilogue: C[slab][m*ldc + n] = post(acc)
//   v = acc*scale (+ bias_n[n]) (+ bias_m[m]);  act: 0 none, 1 relu, 2 tanh
//   gate: 0 none, 1 relu'  (v *= gate_src[m*ldg+n] > 0), 2 tanh' (v *= 1 - gate_src^2)
//   accumulate: C += v (single slice only)
// ==========================================================================================
struct RowMajorEpi {
    float* c; int M; int N; long ldc; long slab_stride;
    const float* bias_n; const float* bias_m; float scale; int act;
    const float* gate_src; long ldg; int gate; int accumulate;
    const float* div_m; const float* div_n; float div_add;     // v /= (div_m[m]*div_n[n] + div_add)  (KFAC damping)
    float* rowsum_dst; long rowsum_stride;                     // optional: rowsum_dst[slab*stride + m] = sum_k Mop(m, k)
    int symmetric;                                             // C = X X^T: tiles strictly below the diagonal are skipped;
                                                               // the caller mirrors the upper triangle afterwards
    // optional second copy of C with every row read as [ch][oh][ow] planes (c2_sp = oh*ow) and written zero-padded:
    // c2[m*ldc2 + ch*c2_psz + y*c2_pw + x + c2_off0] -- the layout the conv dgrad behind an FC dgrad reads
    float* c2; long ldc2; int c2_sp, c2_ow, c2_pw, c2_psz, c2_off0;
    static constexpr bool HAS_ROWSUM = true;
    __device__ __forceinline__ bool wants_rowsum() const { return rowsum_dst != nullptr; }
    __device__ __forceinline__ void rowsum(int m, float v, int ks) const {
        if (m < M) rowsum_dst[(long)ks * rowsum_stride + m] = v;
    }
    struct Col { int n; int ok; float bn; float dn; int off2; };
    __device__ __forceinline__ Col col(int n) const {
        Col cc; cc.n = n; cc.ok = n < N; cc.bn = (cc.ok && bias_n) ? bias_n[n] : 0.f;
        cc.dn = (cc.ok && div_n) ? div_n[n] : 0.f;
        cc.off2 = 0;
        if (c2 && cc.ok) { const int ch = n / c2_sp, e = n % c2_sp; cc.off2 = ch * c2_psz + (e / c2_ow) * c2_pw + e % c2_ow + c2_off0; }
        return cc;
    }
    // The gate operand of the 16 rows a lane stores is fetched by the kernel BEFORE the first store (GATE_PREFETCH):
    // inside st() every load sat behind the previous row's stores, which the compiler cannot prove disjoint from it --
    // sixteen dependent round trips per lane in the epilogue of every gated dgrad.
    static constexpr bool GATE_PREFETCH = true;
    __device__ __forceinline__ float pre(const Col& cc, int m) const {
        return (gate && cc.ok && m < M) ? gate_src[(long)m * ldg + cc.n] : 1.f;
    }
    __device__ __forceinline__ void st(const Col& cc, int m, float acc, int ks, float h) const {
        if (!cc.ok || m >= M) return;
        float v = acc * scale + cc.bn;
        if (bias_m) v += bias_m[m];
        if (div_m) v = v / (div_m[m] * cc.dn + div_add);
        if (act == 1) v = v > 0.f ? v : 0.f;
        else if (act == 2) v = tanhf(v);
        if (gate == 1) v = h > 0.f ? v : 0.f;
        else if (gate == 2) v = v * (1.f - h * h);
        float* d = c + (long)ks * slab_stride + (long)m * ldc + cc.n;
        if (accumulate) v += *d;
        *d = v;
        if (c2) c2[(long)m * ldc2 + cc.off2] = v;
    }
};
template <>
struct EpiSym<RowMajorEpi> {
    __device__ static __forceinline__ bool skip(const RowMajorEpi& e, int m0, int n0, int BN) {
        return e.symmetric && m0 > n0 + BN - 1;
    }
};
static inline RowMajorEpi row_major_epi(float* c, int M, int N, long ldc) {
    RowMajorEpi e; e.c = c; e.M = M; e.N = N; e.ldc = ldc; e.slab_stride = 0; e.bias_n = nullptr; e.bias_m = nullptr;
    e.scale = 1.f; e.act = 0; e.gate_src = nullptr; e.ldg = 0; e.gate = 0; e.accumulate = 0;
    e.div_m = nullptr; e.div_n = nullptr; e.div_add = 0.f; e.rowsum_dst = nullptr; e.rowsum_stride = 0; e.symmetric = 0;
    e.c2 = nullptr; e.ldc2 = 0; e.c2_sp = 1; e.c2_ow = 1; e.c2_pw = 0; e.c2_psz = 0; e.c2_off0 = 0;
    return e;
}

// c[i][j] = c[j][i] for i > j (square, row major, leading dimension n)
int launch_symmetrize_lower(float* c, int n, hipStream_t stream);

// Fixed-order slab reduction: dst[i] (=|+=) scale * sum_s slabs[s*stride + i].  With >= 16 slabs it runs in two
// passes and keeps up to SLAB_GROUP_EXTRA group partials right behind the slabs: size workspaces accordingly.
constexpr int SLAB_GROUP_EXTRA = 32;
int launch_slab_reduce(float* dst, const float* slabs, int nslabs, long n, long stride, float scale, int accumulate,
                       hipStream_t stream);
// Same with a fused epilogue on the final pass: (+ bias[i % ncols]) -> act (0 none, 1 relu, 2 tanh).
int launch_slab_reduce_ex(float* dst, const float* slabs, int nslabs, long n, long stride, float scale, int accumulate,
                          const float* bias, int ncols, int act, float* group_ws, hipStream_t stream);

}  // namespace acktr
