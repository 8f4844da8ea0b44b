// Stationary-accumulator KFAC input factor of the 8x8 stride-4 layer on 84x84x4 frames (reference kfac.py:16-46 for
// NatureCNN's first convolution):   A = alpha * P^T P,   P = im2col patches [B*400, 256]   (never materialised).
//
// The tile-GEMM SYRK ran this at 48 TFLOP/s on the symmetric-half count (0.31 of the fp32 MFMA peak): both operand
// tiles are gathered im2col tiles, three 128x128 output tiles exist, so the reduction is cut into 256 slices of
// 128x128-tile workgroups with two barriers per K-tile pair.  Here (same idea as conv1_wgrad_sa_kernel):
//   * v_mfma_f32_32x32x2_f32 takes A as "lane (row l%32, k l/32)" and B as "lane (column l%32, k l/32)": for a Gram
//     matrix both are the SAME register -- lane (li, h) holds the patch element (tap of li, position of h), and tile
//     (i, j) of the factor is  acc += mfma(block_i, block_j).  Every loaded float serves as row AND column operand;
//   * one aligned 16-byte load of a frame row gives a lane four taps kx = 4(li%2) .. +3, i.e. one float for each of four
//     interleaved 32-tap blocks; two loads (channel halves) give all eight blocks = 256 taps of one position;
//   * the 36 upper-triangle 32x32 tiles are dealt to the four waves of a workgroup, nine accumulators each (the wave index
//     selects one of four fully unrolled code paths so that every register index is static); all waves load the same
//     patches (the L1 serves three of the four), nine MFMAs behind every two loads;
//   * one workgroup per CU owns a contiguous range of row-pair groups (40 positions, see conv1_wgrad_sa_kernel) and
//     writes ONE slab; the shared fixed-order slab reduction applies alpha.  The tiles are "upper" in BLOCK order
//     (block = 4*(channel half) + kx%4), which is not the tap order: cov_a_conv1_mirror_kernel then fills element
//     (r, c) with block(r) > block(c) from (c, r) (those slab elements are never written, and what the reduction summed
//     there is overwritten).
#pragma once
#include "gemm_core.hpp"

namespace acktr {

constexpr int C1A_H = 84, C1A_C = 4, C1A_D = 256, C1A_GROUPS = 10;

struct C1aBlock { f32x4 v[2][4]; };      // [channel half][position s of the lane's four]

// k-th upper-triangle tile pair (i <= j over 8 blocks, row-major enumeration) of wave W: pair number 4k + W
constexpr int c1a_pair(int W, int k, int which) {
    int p = 0;
    for (int i = 0; i < 8; ++i)
        for (int j = i; j < 8; ++j) {
            if (p == 4 * k + W) return which ? j : i;
            ++p;
        }
    return 0;
}

template <int W>
struct C1aTab {
    int i[9], j[9];
    constexpr C1aTab() : i{}, j{} {
        for (int k = 0; k < 9; ++k) { i[k] = c1a_pair(W, k, 0); j[k] = c1a_pair(W, k, 1); }
    }
};

// tap index of lane li in interleaved block blk = 4*nh + q
__device__ __forceinline__ int c1a_tap(int blk, int li) {
    return (((2 * (blk >> 2) + (li >> 4)) * 8 + ((li >> 1) & 7)) * 8) + (li & 1) * 4 + (blk & 3);
}

template <int W>
__device__ __forceinline__ void c1a_wave(const float* __restrict__ x, const int64_t* __restrict__ idx, float* __restrict__ slab,
                                         int g0, int g1, int li, int h) {
    constexpr C1aTab<W> tab{};
    int boff[2][5];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int p = 8 * j + 4 * h, oy = p / 20, ox0 = p % 20;
            boff[nh][j] = ((2 * nh + (li >> 4)) * C1A_H + oy * 4 + ((li >> 1) & 7)) * C1A_H + ox0 * 4 + (li & 1) * 4;
        }
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    C1aBlock ring[5];
    auto ptr = [&](int g) {                      // group g -> image, row pair (wave-uniform)
        const int b = g / C1A_GROUPS, t = g - b * C1A_GROUPS;
        const long img = idx ? idx[b] : (long)b;
        return x + img * (C1A_C * C1A_H * C1A_H) + t * (8 * C1A_H);
    };
    if (g0 < g1) {
        const float* xb = ptr(g0);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) ring[j].v[nh][s] = *reinterpret_cast<const f32x4*>(xb + boff[nh][j] + 4 * s);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int g = g0; g < g1; ++g) {
            const float* xn = ptr(g + 1 < g1 ? g + 1 : g1 - 1);     // past the end: the last group again (never multiplied)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                C1aBlock& blk = ring[j];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const int bi = tab.i[k], bj = tab.j[k];
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(blk.v[bi >> 2][s][bi & 3], blk.v[bj >> 2][s][bj & 3], acc[k], 0, 0, 0);
                    }
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) blk.v[nh][s] = *reinterpret_cast<const f32x4*>(xn + boff[nh][j] + 4 * s);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int bi = tab.i[k], bj = tab.j[k];
        const int col = c1a_tap(bj, li);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = c1a_tap(bi, 4 * h + (r & 3) + 8 * (r >> 2));
            slab[row * C1A_D + col] = acc[k][r];
        }
    }
}

// grid = slabs (one workgroup per CU), block = 4 waves
__global__ __launch_bounds__(256) void cov_a_conv1_sa_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                             float* __restrict__ slabs, int B) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, h = lane >> 5;
    const long G = (long)B * C1A_GROUPS;
    const int g0 = (int)(G * blockIdx.x / gridDim.x), g1 = (int)(G * (blockIdx.x + 1) / gridDim.x);
    float* slab = slabs + (long)blockIdx.x * (C1A_D * C1A_D);
    if (wave == 0) c1a_wave<0>(x, idx, slab, g0, g1, li, h);
    else if (wave == 1) c1a_wave<1>(x, idx, slab, g0, g1, li, h);
    else if (wave == 2) c1a_wave<2>(x, idx, slab, g0, g1, li, h);
    else c1a_wave<3>(x, idx, slab, g0, g1, li, h);
}

// out[r][c] = out[c][r] wherever block(r) > block(c)
__global__ __launch_bounds__(256) void cov_a_conv1_mirror_kernel(float* out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int r = e >> 8, c = e & 255;
    const int br = ((r >> 7) << 2) | (r & 3), bc = ((c >> 7) << 2) | (c & 3);
    if (br > bc) out[e] = out[c * C1A_D + r];
}

}  // namespace acktr
