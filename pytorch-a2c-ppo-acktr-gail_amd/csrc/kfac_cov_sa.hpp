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
#include <type_traits>
#include "gemm_core.hpp"

namespace acktr {

constexpr int C1A_H = 84, C1A_C = 4, C1A_D = 256, C1A_GROUPS = 10;

// [channel half][position s of the lane's four]; XT = uint8_t (frames stored as bytes): the four kx of a position are
// one dword, a channel half's four positions ONE 16-byte load, converted exactly where the MFMA consumes them
template <class XT> struct C1aBlock { f32x4 v[2][4]; };
typedef unsigned int c1a_u32x4 __attribute__((ext_vector_type(4)));
template <> struct C1aBlock<uint8_t> { c1a_u32x4 v[2]; };
__device__ __forceinline__ float c1a_val(const C1aBlock<float>& b, int nh, int s, int q) { return b.v[nh][s][q]; }
__device__ __forceinline__ float c1a_val(const C1aBlock<uint8_t>& b, int nh, int s, int q) {
    return (float)((b.v[nh][s] >> (8 * q)) & 0xffu);
}

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

template <int W, class XT>
__device__ __forceinline__ void c1a_wave(const XT* __restrict__ x, const int64_t* __restrict__ idx, float* __restrict__ slab,
                                         int g0, int g1, int li, int h) {
    constexpr C1aTab<W> tab{};
    constexpr bool U8 = std::is_same<XT, uint8_t>::value;
    struct __attribute__((packed, aligned(4))) UQ { c1a_u32x4 v; };
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
    C1aBlock<XT> ring[5];
    auto ptr = [&](int g) {                      // group g -> image, row pair (wave-uniform)
        const int b = g / C1A_GROUPS, t = g - b * C1A_GROUPS;
        const long img = idx ? idx[b] : (long)b;
        return x + img * (C1A_C * C1A_H * C1A_H) + t * (8 * C1A_H);
    };
    if (g0 < g1) {
        const XT* xb = ptr(g0);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if constexpr (U8) {
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) ring[j].v[nh] = reinterpret_cast<const UQ*>(xb + boff[nh][j])->v;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) ring[j].v[nh][s] = *reinterpret_cast<const f32x4*>(xb + boff[nh][j] + 4 * s);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int g = g0; g < g1; ++g) {
            const XT* xn = ptr(g + 1 < g1 ? g + 1 : g1 - 1);     // past the end: the last group again (never multiplied)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                C1aBlock<XT>& blk = ring[j];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const int bi = tab.i[k], bj = tab.j[k];
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1a_val(blk, bi >> 2, s, bi & 3), c1a_val(blk, bj >> 2, s, bj & 3), acc[k], 0, 0, 0);
                    }
                    if constexpr (!U8) {
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh) blk.v[nh][s] = *reinterpret_cast<const f32x4*>(xn + boff[nh][j] + 4 * s);
                    } else if (s == 3) {
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh) blk.v[nh] = reinterpret_cast<const UQ*>(xn + boff[nh][j])->v;
                    }
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
template <class XT>
__global__ __launch_bounds__(256) void cov_a_conv1_sa_kernel(const XT* __restrict__ x, const int64_t* __restrict__ idx,
                                                             float* __restrict__ slabs, int B) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, h = lane >> 5;
    const long G = (long)B * C1A_GROUPS;
    const int g0 = (int)(G * blockIdx.x / gridDim.x), g1 = (int)(G * (blockIdx.x + 1) / gridDim.x);
    float* slab = slabs + (long)blockIdx.x * (C1A_D * C1A_D);
    if (wave == 0) c1a_wave<0, XT>(x, idx, slab, g0, g1, li, h);
    else if (wave == 1) c1a_wave<1, XT>(x, idx, slab, g0, g1, li, h);
    else if (wave == 2) c1a_wave<2, XT>(x, idx, slab, g0, g1, li, h);
    else c1a_wave<3, XT>(x, idx, slab, g0, g1, li, h);
}

// out[r][c] = out[c][r] wherever block(r) > block(c)
__global__ __launch_bounds__(256) void cov_a_conv1_mirror_kernel(float* out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int r = e >> 8, c = e & 255;
    const int br = ((r >> 7) << 2) | (r & 3), bc = ((c >> 7) << 2) | (c & 3);
    if (br > bc) out[e] = out[c * C1A_D + r];
}

// ------------------------------------------------------------------------------------------------------------------
// Same idea for the 4x4 stride-2 layer on 20x20x32 activations (NatureCNN's second convolution): A = alpha * P^T P,
// P = patches [B*81, 512].  (Tile GEMM: 342 us = 40 TFLOP/s on the symmetric-half count.)
//   * lane (li, h) stands for (channel 8g + li/4, ky = li%4) of channel group g (4 groups) and loads the four kx of its
//     patch row with ONE 16-byte load (8-byte aligned: the stride is 2) -- float q belongs to block 4g + q, 16 blocks;
//   * the 136 upper 32x32 tiles are cut by GROUP PAIR (ga <= gb): six cross pairs of 16 tiles, four diagonal pairs of 10.
//     A workgroup works on ONE pair: its four waves run the same program (16 or 10 accumulators, two loads or one per
//     k-pair step) on four different image ranges, add their partial sums through LDS in a fixed order and write one
//     64 KB region slab.  The host deals workgroups to the ten pairs in proportion to their MFMA counts;
//   * an image has 81 positions = ten blocks of 8 + one position: the eleventh block is a single k-pair step in which
//     the upper k half is zeroed;  twelve virtual blocks per image walk a ring of three register slots, all offsets are
//     compile-time constants selected by h;
//   * c2a_reduce_kernel sums the region slabs in order, applies alpha and writes element (r, c) and its mirror.
// The 3x3 stride-1 layer on 9x9x64 activations (third convolution, 576 taps, 49 positions) runs the same program with
// three kx per lane (one 12-byte load), six groups of 32 (channel, ky) rows and 21 group pairs of 9 / 6 tiles.
struct CovGeom2 {      // 4x4 stride 2 on 20x20x32
    static constexpr int NQ = 4, NG = 4, NPOS = 81, OW = 9, SY = 40, SX = 2, IMG = 32 * 400, D = 512, VBLOCKS = 12;
    typedef f32x4 vec;
    __device__ static __forceinline__ int lane_base(int, int li) { return (li >> 2) * 400 + (li & 3) * 20; }
    __device__ static __forceinline__ long group_base(int g) { return (long)g * (8 * 400); }
    __device__ static __forceinline__ int tap(int g, int q, int l) { return (((8 * g + (l >> 2)) * 4 + (l & 3)) * 4) + q; }
};
struct CovF3 { float v[3]; __device__ __forceinline__ float operator[](int i) const { return v[i]; } };
struct CovGeom3 {      // 3x3 stride 1 on 9x9x64
    static constexpr int NQ = 3, NG = 6, NPOS = 49, OW = 7, SY = 9, SX = 1, IMG = 64 * 81, D = 576, VBLOCKS = 9;
    typedef CovF3 vec;
    __device__ static __forceinline__ int lane_base(int g, int li) { const int pr = 32 * g + li; return (pr / 3) * 81 + (pr % 3) * 9; }
    __device__ static __forceinline__ long group_base(int) { return 0; }
    __device__ static __forceinline__ int tap(int g, int q, int l) { return (32 * g + l) * 3 + q; }
};
constexpr int COVSA_MAX_PAIRS = 21;
template <class G> struct CovSa {
    static constexpr int PAIRS = G::NG * (G::NG + 1) / 2;
    static constexpr int NT_CROSS = G::NQ * G::NQ, NT_DIAG = G::NQ * (G::NQ + 1) / 2;
    static constexpr int SLAB = NT_CROSS * 1024;                   // floats per region slab
    static constexpr int LDS_BYTES = 2 * NT_CROSS * 1024 * 4;      // two waves' accumulators
    static constexpr int NBLK = (G::NPOS + 7) / 8;                 // real 8-position blocks per image (the last: 1 position)
    static_assert(G::NPOS % 8 == 1 && G::VBLOCKS % 3 == 0 && G::VBLOCKS >= NBLK, "position blocking");
};
struct CovPlan {
    int wg0[COVSA_MAX_PAIRS + 1];      // workgroups [wg0[p], wg0[p+1]) work on pair p; slab index == workgroup index
    int ga[COVSA_MAX_PAIRS], gb[COVSA_MAX_PAIRS];
};

template <class G> constexpr int cov_pos_off(int p) { return (p / G::OW) * G::SY + (p % G::OW) * G::SX; }

template <class G, bool DIAG>
struct CovSlot;
template <class G>
struct CovSlot<G, true> { typename G::vec a[4]; };
template <class G>
struct CovSlot<G, false> { typename G::vec a[4], b[4]; };

template <class G, bool DIAG>
__device__ __forceinline__ void cov_sa_wave(const float* __restrict__ x, float* __restrict__ slab, float* lds, int ga, int gb,
                                            int img0, int img1, int lane, int wave) {
    using S = CovSa<G>;
    using Vec = typename G::vec;
    constexpr int NQ = G::NQ, NT = DIAG ? S::NT_DIAG : S::NT_CROSS, LAST = S::NBLK - 1, VB = G::VBLOCKS;
    const int li = lane & 31, h = lane >> 5;
    const int lba = G::lane_base(ga, li), lbb = G::lane_base(gb, li);
    f32x16 acc[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    CovSlot<G, DIAG> ring[3];
    const float* xa = x + G::group_base(ga);
    const float* xb = x + G::group_base(gb);
    // virtual block vb of the image at element offset io: blocks past LAST do not exist, block LAST has one step
    auto issue = [&](auto VBc, CovSlot<G, DIAG>& sl, long io) {
        constexpr int vb = decltype(VBc)::value;
        if constexpr (vb <= LAST) {
#pragma unroll
            for (int s = 0; s < (vb == LAST ? 1 : 4); ++s) {
                const int p0 = 8 * vb + s, p1 = (vb == LAST) ? G::NPOS - 1 : 8 * vb + 4 + s;
                const int po = h ? cov_pos_off<G>(p1) : cov_pos_off<G>(p0);
                sl.a[s] = *reinterpret_cast<const Vec*>(xa + io + lba + po);
                if constexpr (!DIAG) sl.b[s] = *reinterpret_cast<const Vec*>(xb + io + lbb + po);
            }
        }
    };
    auto consume = [&](auto VBc, CovSlot<G, DIAG>& sl) {
        constexpr int vb = decltype(VBc)::value;
        if constexpr (vb <= LAST) {
#pragma unroll
            for (int s = 0; s < (vb == LAST ? 1 : 4); ++s) {
                float va[NQ], vbv[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    va[q] = sl.a[s][q];
                    if constexpr (DIAG) vbv[q] = va[q]; else vbv[q] = sl.b[s][q];
                    if constexpr (vb == LAST) {               // positions past the image's last do not exist: zero k half 1
                        va[q] = h ? 0.f : va[q];
                        vbv[q] = h ? 0.f : vbv[q];
                    }
                }
                int k = 0;
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
                    for (int qj = (DIAG ? qi : 0); qj < NQ; ++qj) {
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[qi], vbv[qj], acc[k], 0, 0, 0);
                        ++k;
                    }
            }
        }
    };
    auto step = [&](auto VBc, long io_cur, long io_next) {     // consume virtual block vb, refill its slot with vb + 3
        constexpr int vb = decltype(VBc)::value;
        CovSlot<G, DIAG>& sl = ring[vb % 3];
        consume(VBc, sl);
        if constexpr (vb + 3 < VB) issue(std::integral_constant<int, vb + 3>{}, sl, io_cur);
        else issue(std::integral_constant<int, vb + 3 - VB>{}, sl, io_next);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (img0 < img1) {
        const long io0 = (long)img0 * G::IMG;
        issue(std::integral_constant<int, 0>{}, ring[0], io0);
        issue(std::integral_constant<int, 1>{}, ring[1], io0);
        issue(std::integral_constant<int, 2>{}, ring[2], io0);
        __builtin_amdgcn_sched_barrier(0);
        for (int img = img0; img < img1; ++img) {
            const long io = (long)img * G::IMG, ion = (long)(img + 1 < img1 ? img + 1 : img1 - 1) * G::IMG;
            step(std::integral_constant<int, 0>{}, io, ion);
            step(std::integral_constant<int, 1>{}, io, ion);
            step(std::integral_constant<int, 2>{}, io, ion);
            step(std::integral_constant<int, 3>{}, io, ion);
            step(std::integral_constant<int, 4>{}, io, ion);
            step(std::integral_constant<int, 5>{}, io, ion);
            step(std::integral_constant<int, 6>{}, io, ion);
            step(std::integral_constant<int, 7>{}, io, ion);
            step(std::integral_constant<int, 8>{}, io, ion);
            if constexpr (VB > 9) {
                step(std::integral_constant<int, 9 % VB>{}, io, ion);
                step(std::integral_constant<int, 10 % VB>{}, io, ion);
                step(std::integral_constant<int, 11 % VB>{}, io, ion);
            }
        }
    }
    // fixed-order sum of the four waves: (w0 + w2) + (w1 + w3), through LDS in two rounds
    float* mine = lds + (size_t)(wave & 1) * (NT * 1024);
    if (wave >= 2) {
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) mine[(k * 16 + i) * 64 + lane] = acc[k][i];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[k][i] += mine[(k * 16 + i) * 64 + lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) lds[(k * 16 + i) * 64 + lane] = acc[k][i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) slab[(k * 16 + i) * 64 + lane] = acc[k][i] + lds[(k * 16 + i) * 64 + lane];
    }
}

template <class G>
__global__ __launch_bounds__(256) void cov_a_sa_kernel(const float* __restrict__ x, float* __restrict__ slabs, int B,
                                                       const CovPlan plan) {
    extern __shared__ __attribute__((aligned(16))) float cov_sa_lds[];
    using S = CovSa<G>;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int p = 0;
#pragma unroll
    for (int i = 1; i < S::PAIRS; ++i) p += (int)blockIdx.x >= plan.wg0[i];
    const int nw = plan.wg0[p + 1] - plan.wg0[p], w = (int)blockIdx.x - plan.wg0[p];
    const int ranges = nw * 4, r = w * 4 + wave;
    const int img0 = (int)((long)B * r / ranges), img1 = (int)((long)B * (r + 1) / ranges);
    float* slab = slabs + (long)blockIdx.x * S::SLAB;
    const int ga = plan.ga[p], gb = plan.gb[p];
    if (ga == gb) cov_sa_wave<G, true>(x, slab, cov_sa_lds, ga, gb, img0, img1, lane, wave);
    else cov_sa_wave<G, false>(x, slab, cov_sa_lds, ga, gb, img0, img1, lane, wave);
}

// out[r][c] = out[c][r] = alpha * sum over the pair's slabs, in slab order.  grid = (NT_CROSS * 4, PAIRS), block 256
template <class G>
__global__ __launch_bounds__(256) void cov_sa_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, float alpha,
                                                            const CovPlan plan) {
    using S = CovSa<G>;
    const int p = blockIdx.y, ga = plan.ga[p], gb = plan.gb[p];
    const int e = blockIdx.x * 256 + threadIdx.x;              // (tile slot k, register i, lane)
    const int k = e >> 10, i = (e >> 6) & 15, lane = e & 63;
    int qi = 0, qj = 0;
    if (ga == gb) {                                            // k-th pair (qi <= qj), row-major
        if (k >= S::NT_DIAG) return;
        int c = 0;
        for (int a = 0; a < G::NQ; ++a)
            for (int b = a; b < G::NQ; ++b) { if (c == k) { qi = a; qj = b; } ++c; }
    } else { qi = k / G::NQ; qj = k % G::NQ; }
    float s = 0.f;
    for (int w = plan.wg0[p]; w < plan.wg0[p + 1]; ++w) s += slabs[(long)w * S::SLAB + e];
    s *= alpha;
    const int row = G::tap(ga, qi, 4 * (lane >> 5) + (i & 3) + 8 * (i >> 2)), col = G::tap(gb, qj, lane & 31);
    out[row * G::D + col] = s;
    out[col * G::D + row] = s;
}

}  // namespace acktr
