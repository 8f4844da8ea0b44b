// EXPERIMENT (not part of the library; driver: tools/gemm_direct_probe.hip).  Measured on MI355X at the FC shapes of
// NatureCNN, B = 1024 (rocprofv3 kernel time): forward 24.3-25.4 us (LDS-tiled production kernel: 24.3), dgrad 26.6
// without the gate epilogue (32.8 with it), wgrad 27.1 + slab reduction (35.9).  With the stores removed and every load
// hitting one cache line the forward still takes 22 us: 6-8 us are the fixed cost of a kernel that loads, multiplies and
// stores (K = 32 alone: 8.5 us), the K loop itself runs at 106 TFLOP/s.  A 1.6 GFLOP GEMM is overhead-bound on this
// chip whatever the skeleton, so the production path keeps ONE GEMM skeleton.  `dbg` bits: 1 no stores, 2 all loads
// from k = 0, 4 run the K loop four times.
//
// Register-direct fp32 MFMA GEMM for the mid-size dense layers (NatureCNN's 1568 <-> 512 FC layer at B ~ 1024 and the
// KFAC preconditioner products): C(m,n) = sum_k A(m,k) * B(n,k).
//
// Why a second GEMM skeleton.  The LDS-tiled kernel (gemm_core.hpp) runs these 1.6 GFLOP problems at 44-54 TFLOP/s: a
// workgroup's K loop is only 8-16 tiles long and its four waves meet at a barrier every 16 MFMAs.  Here NO operand
// passes through LDS and there is NO barrier: v_mfma_f32_32x32x2_f32 wants, per lane, ONE float of A (row l%32, k
// l/32) and ONE float of B (column l%32, k l/32) -- exactly what a lane can fetch for itself:
//   * operand contiguous along k ("KC": X[b][k], W[o][k], dY[b][o]): lane (r = l%32, h = l/32) loads the 16 bytes
//     k0 + 8j + 4h .. +3 of ITS row; the four floats feed four MFMA steps (the k order inside a 32-block is permuted
//     identically for both operands, which a sum does not notice bit-wise only if both use it -- they do).  The four
//     loads j = 0..3 of a 32-k block are issued back to back and together cover whole 128-byte lines of 32 rows.
//   * operand contiguous along its row index ("RC": W[k][n] in dgrad, dY[k][m] / X[k][n] in wgrad): lane (c, h) loads
//     BLK consecutive floats  n0 + BLK*c .. of row k; float q belongs to MFMA block q, i.e. the BLK 32-column blocks of
//     a wave are INTERLEAVED (block q = columns congruent q mod BLK), one 8/16-byte coalesced load serves all blocks
//     and the epilogue stores BLK adjacent floats per lane.
// A wave owns a (32*MB) x (32*NB) tile and streams 32-k blocks through two register sets (block t+1 in flight while
// block t is multiplied: 16*MB*NB MFMAs = 1-4 k cycles, far above the L2 latency); a workgroup is WM x WN such waves
// that share nothing but the L1.  K is split over blockIdx.z into slabs (fixed-order reduction: linear.hip).
// Requirements (checked by the launcher): K % 32 == 0 per slice granularity, rows 16-byte aligned; M / N tails are
// handled by clamping load addresses and masking stores.
#pragma once
#include "gemm_core.hpp"

namespace acktr {

template <bool KC, int BLK>
struct DirectOperand;

// k-contiguous: element (row, k) at p[row * ld + k]
template <int BLK>
struct DirectOperand<true, BLK> {
    f32x4 r[BLK][4];
    const float* base[BLK];
    __device__ __forceinline__ void init(const float* p, long ld, int row0, int rows, int li, int lh) {
#pragma unroll
        for (int b = 0; b < BLK; ++b) {
            int row = row0 + 32 * b + li;
            row = row < rows ? row : rows - 1;
            base[b] = p + (long)row * ld + 4 * lh;
        }
    }
    __device__ __forceinline__ void load(int k0) {
#pragma unroll
        for (int b = 0; b < BLK; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) r[b][j] = *reinterpret_cast<const f32x4*>(base[b] + k0 + 8 * j);
    }
    __device__ __forceinline__ float frag(int b, int j, int s) const { return r[b][j][s]; }
    // tile row of MFMA block b, MFMA row index i (0..31) -> row offset inside the wave tile
    __device__ static __forceinline__ int row_of(int b, int i) { return 32 * b + i; }
};

// row-contiguous: element (row, k) at p[k * ld + row]; BLK interleaved blocks, BLK in {1, 2, 4}
template <int BLK>
struct DirectOperand<false, BLK> {
    typedef float vec __attribute__((ext_vector_type(BLK)));
    vec r[4][4];
    const float* base;
    long ld;
    __device__ __forceinline__ void init(const float* p, long ld_, int row0, int rows, int li, int lh) {
        int row = row0 + BLK * li;
        row = row + BLK <= rows ? row : rows - BLK;
        base = p + (long)(4 * lh) * ld_ + row;
        ld = ld_;
    }
    __device__ __forceinline__ void load(int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) r[j][s] = *reinterpret_cast<const vec*>(base + (long)(k0 + 8 * j + s) * ld);
    }
    __device__ __forceinline__ float frag(int b, int j, int s) const { return r[j][s][b]; }
    __device__ static __forceinline__ int row_of(int b, int i) { return BLK * i + b; }
};
template <>
struct DirectOperand<false, 1> {
    float r[4][4];
    const float* base;
    long ld;
    __device__ __forceinline__ void init(const float* p, long ld_, int row0, int rows, int li, int lh) {
        int row = row0 + li;
        row = row < rows ? row : rows - 1;
        base = p + (long)(4 * lh) * ld_ + row;
        ld = ld_;
    }
    __device__ __forceinline__ void load(int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) r[j][s] = base[(long)(k0 + 8 * j + s) * ld];
    }
    __device__ __forceinline__ float frag(int, int j, int s) const { return r[j][s]; }
    __device__ static __forceinline__ int row_of(int, int i) { return i; }
};

// Plain epilogue: C (row major, ldc) or slab z of it.
struct DirectStore {
    float* c;
    long ldc, slab_stride;
    __device__ __forceinline__ void st(int m, int n, float v, int z) const { c[(long)z * slab_stride + (long)m * ldc + n] = v; }
};

template <int MB, int NB, bool AKC, bool BKC, int WM, int WN, class Epi>
__global__ __launch_bounds__(64 * WM * WN) void gemm_direct_kernel(const float* __restrict__ A, long lda,
                                                                   const float* __restrict__ B, long ldb, const Epi epi,
                                                                   int M, int N, int kblocks, int tiles_n, int dbg = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware numbering (workgroups go round-robin over the 8 XCDs): consecutive virtual ids share an XCD
    unsigned L = blockIdx.x;
    const unsigned total = gridDim.x;
    if (total % 8 == 0) L = (L % 8) * (total / 8) + L / 8;
    const int tn = L % tiles_n, tm = L / tiles_n;
    const int z = blockIdx.y, nz = gridDim.y;
    const int kb0 = (int)((long)kblocks * z / nz), kb1 = (int)((long)kblocks * (z + 1) / nz);
    const int m0 = (tm * WM + wm) * (32 * MB), n0 = (tn * WN + wn) * (32 * NB);

    DirectOperand<AKC, MB> a[2];
    DirectOperand<BKC, NB> b[2];
    a[0].init(A, lda, m0, M, li, lh);
    b[0].init(B, ldb, n0, N, li, lh);
    a[1] = a[0];
    b[1] = b[0];

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto issue = [&](auto P, int kb) {          // unconditional (past the end: the last block again, never multiplied)
        const int kc = kb < kb1 ? kb : kb1 - 1;
        a[decltype(P)::value].load((dbg & 2) ? 0 : 32 * kc);
        b[decltype(P)::value].load((dbg & 2) ? 0 : 32 * kc);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto multiply = [&](auto P) {
        constexpr int p = decltype(P)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                        acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p].frag(i, j, s), b[p].frag(q, j, s), acc[i][q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    for (int rep = 0; rep < ((dbg & 4) ? 4 : 1); ++rep)
    if (kb1 > kb0) {
        issue(P0{}, kb0);
        const int npairs = (kb1 - kb0) / 2;
        int kb = kb0;
        for (int i = 0; i < npairs; ++i, kb += 2) {
            issue(P1{}, kb + 1);
            multiply(P0{});
            issue(P0{}, kb + 2);
            multiply(P1{});
        }
        if ((kb1 - kb0) & 1) multiply(P0{});
    }
    // epilogue: lane -> column index li of each block, registers -> rows
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + DirectOperand<AKC, MB>::row_of(i, 4 * lh + (r & 3) + 8 * (r >> 2));
                const int n = n0 + DirectOperand<BKC, NB>::row_of(q, li);
                if (m < M && n < N && (!(dbg & 1) || acc[i][q][r] == 12345.678f)) epi.st(m, n, acc[i][q][r], z);
            }
}

template <int MB, int NB, bool AKC, bool BKC, int WM, int WN, class Epi>
int launch_gemm_direct(const float* A, long lda, const float* B, long ldb, const Epi& epi, int M, int N, int K, int nslices,
                       hipStream_t stream, int dbg = 0) {
    if (K % 32 || K < 32) { set_error("gemm_direct: K must be a multiple of 32"); return -1; }
    const int tiles_m = cdiv(M, 32 * MB * WM), tiles_n = cdiv(N, 32 * NB * WN);
    if (nslices < 1) nslices = 1;
    if (nslices > K / 32) nslices = K / 32;
    auto kern = gemm_direct_kernel<MB, NB, AKC, BKC, WM, WN, Epi>;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, nslices), dim3(64 * WM * WN), 0, stream, A, lda, B, ldb, epi, M, N, K / 32,
                       tiles_n, dbg);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

}  // namespace acktr
