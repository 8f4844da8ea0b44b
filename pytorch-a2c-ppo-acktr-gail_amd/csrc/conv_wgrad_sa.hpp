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
#include <type_traits>
#include "gemm_core.hpp"

namespace acktr {

constexpr int SA_OC = 32, SA_KK = 9, SA_H = 9, SA_OH = 7, SA_SP = 49, SA_HW = 81, SA3_WAVES = 10;

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
template <bool FULL>
__device__ __forceinline__ void sa_consume(const SaRow& r, int kh, f32x16 (&acc)[2], float& bsum) {
    const f32x4 a = sa_shift(r.a, kh, 0.f);                   // position 7 of the row does not exist: A is zero there
    const f32x4 b0 = sa_shift(r.b0, kh, r.b0[3]), b1 = sa_shift(r.b1, kh, r.b1[3]);
    bsum += (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b0[j], acc[0], 0, 0, 0);
        if constexpr (FULL) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b1[j], acc[1], 0, 0, 0);
    }
}

// grid = number of slabs; slab layout: [OC * C*9 | OC].  The C*9/32 = 18 tap tiles go to TEN waves: eight with two tiles
// and two with one (waves 8 and 9, which land on different SIMDs) -- nine two-tile waves put three on one SIMD and two on
// the others (672 MFMAs per image run against 448), ten put 2.5 on two of them (560).
template <int C>
__global__ __launch_bounds__(64 * SA3_WAVES) void conv3_wgrad_sa_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                       float* __restrict__ slabs, long slab_stride, int B,
                                                                       int imgs_per_wg, float scale) {
    static_assert(C * SA_KK == 32 * (2 * (SA3_WAVES - 2) + 2), "eight two-tile waves + two one-tile waves");
    constexpr int KC = C * SA_KK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const bool full = wave < SA3_WAVES - 2;
    const int tile0 = full ? 2 * wave : wave + (SA3_WAVES - 2);                // first 32-tap tile of this wave
    int tapoff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 32 * (tile0 + (full ? t : 0)) + li;                     // (a one-tile wave loads its tile twice)
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
    auto run = [&](auto full_c) {
        for (int b = b0; b < b1; ++b) {
            const int bn = b + 1 < b1 ? b + 1 : b1 - 1;
            const int xs = bn * C * SA_HW * 4, dys = bn * SA_OC * SA_SP * 4;
#pragma unroll
            for (int oy = 0; oy < SA_OH; ++oy) {
                sa_consume<decltype(full_c)::value>(ring[oy], kh, acc, bsum);
                sa_issue(ring[oy], xr, dyr, xs, dys, aoff, tapoff, oy);
                __builtin_amdgcn_sched_barrier(0);             // keep the refill right behind its row: the scheduler would
                                                               // otherwise collect all 21 loads at the end of the image
            }
        }
    };
    if (full) run(std::true_type{}); else run(std::false_type{});
    float* slab = slabs + (long)blockIdx.x * slab_stride;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t == 1 && !full) break;
        const int tap = 32 * (tile0 + t) + li;
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



// ------------------------------------------------------------------------------------------------------------------
// Stationary-accumulator weight gradient of the 8x8 stride-4 layer on 84x84x4 frames (reference model.py:176, its
// autograd backward; the dominant kernel of the PPO step):
//   dW[oc, (c,ky,kx)] = scale * sum_{b,oy,ox} dY[b,oc,oy,ox] * X[idx[b],c,4oy+ky,4ox+kx],   db[oc] = sum dY.
// The tile GEMM ran this 32 x 256 x (B*400) product at 74 TFLOP/s: two output tiles, so the reduction is cut into
// ~400 slices whose workgroups each pay a pipeline fill, two barriers per K-tile pair and an im2col address decode per
// element.  Here the whole dW lives in accumulators and both operands go from global memory straight into the
// registers the MFMA reads (no LDS, no barrier in the loop):
//   * a wave owns 32 oc x 128 taps = channels {2nh, 2nh+1} as FOUR interleaved 32x32 tiles: lane (li, h) stands for the
//     tap group (c = 2nh + li/16, ky = (li/2)%8, kx = 4(li%2) .. +3) and tile q takes kx = 4(li%2) + q -- the four
//     floats of ONE aligned 16-byte load of the frame row are the B values of the four tiles;
//   * the reduction runs over blocks of 8 output positions; lane half h takes positions 4h .. 4h+3 of the block, which
//     lie in one output row (20 % 4 == 0), so they are 4 consecutive 16-byte loads of the same frame row (one
//     64-byte run per lane) and ONE 16-byte load of dY[b, li, ...] for A;
//   * five blocks = 40 positions = two output rows: with the loop unrolled by five every lane offset is a constant of
//     the ring slot, only the image pointer and the row pair t move (scalar).  The ring keeps five blocks (25 loads)
//     in flight per wave, two waves per SIMD.
// A workgroup is 8 waves = 2 channel halves x 4 reduction sub-ranges; the four partial sums of a half are added in LDS
// in a fixed order and ONE slab per workgroup (one workgroup per CU) goes to the shared slab reduction: 8 MB of slabs
// instead of 12.6.  Measured at B = 1024 (MI355X): 90.4 -> 77.5 us per call inside the PPO step (86.6 TFLOP/s); with the
// MFMAs removed the loop takes 30 us (115 MB of frames + 52 MB of dY from HBM), with the loads removed 45 us -- the two
// overlap only partly, the sum of both pipes is what is left to gain.
constexpr int S1_C = 4, S1_H = 84, S1_OC = 32, S1_SP = 400, S1_KC = 256, S1_GROUPS = 10, S1_WAVES = 8;
constexpr int S1_LDS_BYTES = S1_WAVES * 4 * 16 * 64 * 4;

// XT = uint8_t (frames stored as bytes): the lane's 16 frame values of a block are ONE 16-byte
// load instead of four, converted (exactly) where the MFMA consumes them; same products in the same order.
template <class XT> struct S1Block { f32x4 a, b[4]; };
template <> struct S1Block<uint8_t> { f32x4 a; u32x4 bq; };
__device__ __forceinline__ float s1_b(const S1Block<float>& blk, int s, int q) { return blk.b[s][q]; }
__device__ __forceinline__ float s1_b(const S1Block<uint8_t>& blk, int s, int q) { return (float)((blk.bq[s] >> (8 * q)) & 0xffu); }

template <class XT>
__global__ __launch_bounds__(64 * S1_WAVES) void conv1_wgrad_sa_kernel(const XT* __restrict__ x, const int64_t* __restrict__ idx,
                                                                       const float* __restrict__ dy, float* __restrict__ slabs,
                                                                       long slab_stride, int B, float scale) {
    constexpr bool U8 = std::is_same<XT, uint8_t>::value;
    extern __shared__ __attribute__((aligned(16))) float s1_red[];
    // (wave index through readfirstlane: the compiler then keeps the reduction range, the image pointers and the gather
    //  index in scalar registers -- as lane values they made the loop a divergent one with a vector load of idx[b] and a
    //  full vmcnt(0) drain per group)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, h = lane >> 5;
    const int nh = wave & 1, ks = wave >> 1;
    const long G = (long)B * S1_GROUPS;
    const int R = gridDim.x * 4, r = blockIdx.x * 4 + ks;
    const int g0 = (int)(G * r / R), g1 = (int)(G * (r + 1) / R);

    int boff[5], aoff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int p = 8 * j + 4 * h, oy = p / 20, ox0 = p % 20;
        boff[j] = ((2 * nh + (li >> 4)) * S1_H + oy * 4 + ((li >> 1) & 7)) * S1_H + ox0 * 4 + (li & 1) * 4;
        aoff[j] = li * S1_SP + p;
    }
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    float bsum = 0.f;

    S1Block<XT> ring[5];
    auto ptrs = [&](int g, const XT*& xb, const float*& dyb) {      // group g -> image b, row pair t (wave-uniform)
        const int b = g / S1_GROUPS, t = g - b * S1_GROUPS;
        const long img = idx ? idx[b] : (long)b;
        xb = x + img * (S1_C * S1_H * S1_H) + t * (8 * S1_H);
        dyb = dy + (long)b * (S1_OC * S1_SP) + t * 40;
    };
    struct __attribute__((packed, aligned(4))) UQ { u32x4 v; };
    auto issue = [&](S1Block<XT>& blk, int j, const XT* xb, const float* dyb) {
        blk.a = *reinterpret_cast<const f32x4*>(dyb + aoff[j]);
        if constexpr (U8) blk.bq = reinterpret_cast<const UQ*>(xb + boff[j])->v;
        else {
#pragma unroll
            for (int s = 0; s < 4; ++s) blk.b[s] = *reinterpret_cast<const f32x4*>(xb + boff[j] + 4 * s);
        }
    };
    if (g0 < g1) {
        const XT* xb;
        const float* dyb;
        ptrs(g0, xb, dyb);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            issue(ring[j], j, xb, dyb);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int g = g0; g < g1; ++g) {
            const XT* xn;
            const float* dyn;
            ptrs(g + 1 < g1 ? g + 1 : g1 - 1, xn, dyn);      // past the end: the last group again (never multiplied)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                // one refill load behind every four MFMAs (a burst of five loads behind sixteen MFMAs left the matrix
                // pipe idle while the wave sat in the address path's queue)
                S1Block<XT>& blk = ring[j];
                const f32x4 a = blk.a;
                bsum += (a[0] + a[1]) + (a[2] + a[3]);
                blk.a = *reinterpret_cast<const f32x4*>(dyn + aoff[j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], s1_b(blk, s, q), acc[q], 0, 0, 0);
                    if constexpr (!U8) blk.b[s] = *reinterpret_cast<const f32x4*>(xn + boff[j] + 4 * s);
                    else if (s == 3) blk.bq = reinterpret_cast<const UQ*>(xn + boff[j])->v;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    // fixed-order sum of the four sub-range partials of each channel half through LDS, one slab per workgroup
    float* mine = s1_red + (size_t)wave * (4 * 16 * 64);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) mine[(q * 16 + i) * 64 + lane] = acc[q][i];
    __syncthreads();
    float* slab = slabs + (long)blockIdx.x * slab_stride;
    for (int e = threadIdx.x; e < 2 * 16 * 64; e += 64 * S1_WAVES) {
        const int el = e & 63, ei = (e >> 6) & 15, enh = e >> 10;
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += s1_red[(size_t)((2 * k + enh) * 4 + q) * (16 * 64) + ei * 64 + el];
            v[q] = sum * scale;
        }
        const int eli = el & 31, oc = (ei & 3) + 8 * (ei >> 2) + 4 * (el >> 5);
        const int n = ((2 * enh + (eli >> 4)) * 8 + ((eli >> 1) & 7)) * 8 + (eli & 1) * 4;
        *reinterpret_cast<f32x4*>(slab + oc * S1_KC + n) = v;
    }
    // bias: lanes (li, h) of the nh == 0 waves hold the sums over their positions of dY[., li, .]
    __syncthreads();
    if (nh == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (h == 0) s1_red[ks * 32 + li] = bsum;
    }
    __syncthreads();
    if (threadIdx.x < 32)
        slab[S1_OC * S1_KC + threadIdx.x] = ((s1_red[threadIdx.x] + s1_red[32 + threadIdx.x]) + s1_red[64 + threadIdx.x]) + s1_red[96 + threadIdx.x];
}

// ------------------------------------------------------------------------------------------------------------------
// Second form (round 3, the default for float frames): ALL 256 taps in one wave -- eight 32x32 tiles, 128 accumulator
// registers; lane (li, h) stands for the frame row (c = li/8, ky = li%8) and tile q for kx = q, so position s and tap kx read
// float 4s + kx of the lane's run -- and the frames go through a PRIVATE LDS ring of the wave.  What limits the kernel
// above is not bytes but cache-line look-ups: a lane stands for a frame row, so one 16-byte load instruction touches
// 32-64 different 128-byte lines, and the two kx halves of a tile side sit in neighbouring lanes, so every frame float
// goes through the vector memory path four times (512 MB per launch at B = 1024).  An intermediate form (eight tiles per
// wave, register-direct loads, one wave per SIMD) loaded each float twice and ran 72 us stand-alone against 84 -- with
// its MFMAs removed still 51 us for 360 MB, ~27 B/clk per CU.  Here the unit of work is one OUTPUT ROW oy of one image (20 positions, 80 MFMAs per wave): it reads the
// frame rows 4oy .. 4oy+7 of the four channels = the two "quads" oy and oy+1 (a quad = rows 4k .. 4k+3, 4 x 1344
// contiguous bytes), and the next row re-uses quad oy+1.  A wave walks consecutive rows, so every step brings in ONE
// new quad with fully coalesced 16-byte loads (eight lanes per line; 5.25 loads per lane), parks it in a register set for
// one step (the latency budget), writes it to slot (step % 3) of its ring and reads its own 44 floats per lane from there
// (eleven ds_read_b128).  Every frame float crosses the vector memory path once per workgroup: 115 MB per launch, as from
// HBM.  No barrier in the loop (the ring is private), the compiler counts every load (all plain loads).  Steps: image b has
// quads k = 0 .. 20; step (b, k) loads quad k and, for k >= 1, multiplies row oy = k - 1; global step gs = 21 b + k.
// dY (A operand) stays a direct load: lane (oc, h) reads its 10 positions 10h .. 10h+9 of the row (16 + 16 + 8 bytes).
constexpr int S9_WAVES = 8, S9_QF = 4 * 4 * S1_H, S9_QS = S9_QF + 32, S9_WAVE_FLOATS = 3 * S9_QS;
constexpr int S9_LDS_BYTES = S9_WAVES * S9_WAVE_FLOATS * 4;
static_assert(S9_LDS_BYTES >= (4 * 8 * 16 * 64 + S9_WAVES * 32) * 4, "the closing sums re-use the rings");
#ifdef S9_STAMP
__device__ long long s9_stamps[6 * 4096];          // diagnostic build only: (loop cycles, loop 100 MHz ticks, rows, t_entry, t_loop_end, t_exit) per wave
#endif
struct S9A { f32x4 p0, p1; f32x2 p2; };
__device__ __forceinline__ float s9_a(const S9A& a, int s) { return s < 4 ? a.p0[s] : (s < 8 ? a.p1[s - 4] : a.p2[s - 8]); }

// XT = uint8_t: a quad is 4 x 336 bytes (84 pieces: two loads per lane), the ring holds bytes and a lane reads its 44 bytes
// as eleven dwords; the bytes are converted (exactly) where the MFMA consumes them.
template <class XT>
__global__ __launch_bounds__(64 * S9_WAVES) void conv1_wgrad_ring_kernel(const XT* __restrict__ x, const int64_t* __restrict__ idx,
                                                                        const float* __restrict__ dy, float* __restrict__ slabs,
                                                                        long slab_stride, int B, float scale) {
    constexpr bool U8 = std::is_same<XT, uint8_t>::value;
    constexpr int NP = U8 ? 2 : 6;                               // 16-byte pieces per lane and quad (the last one partial)
    constexpr int PIECES = U8 ? 84 : 336, CH_PIECES = PIECES / 4, LAST = PIECES - 64 * (NP - 1);
    constexpr int SLOT = S9_QS * (int)sizeof(XT);                // bytes
    extern __shared__ __attribute__((aligned(16))) float s9_lds[];
#ifdef S9_STAMP
    const long long st_in = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, h = lane >> 5;
    char* wl = reinterpret_cast<char*>(s9_lds + wave * S9_WAVE_FLOATS);
    const long U = (long)B * 20;
    const int R = gridDim.x * S9_WAVES, r = blockIdx.x * S9_WAVES + wave;
    const int u0 = (int)(U * r / R), u1 = (int)(U * (r + 1) / R);

    int poff[NP];                                              // piece 64 i + lane of a quad: [c][4 rows x 84 values], in values
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int pi = lane + 64 * i, pc = pi < PIECES ? pi : PIECES - 1;
        poff[i] = (pc / CH_PIECES) * (S1_H * S1_H) + (pc % CH_PIECES) * (16 / (int)sizeof(XT));
    }
    const int rowoff = (((li >> 3) * 4 + (li & 3)) * S1_H + 40 * h) * (int)sizeof(XT);   // the lane's 44 values inside a slot (bytes)
    const bool upper = (li & 7) >= 4;                                  // frame rows ky >= 4 lie in the step's own quad
    const int aoff = li * S1_SP + 10 * h;

    f32x16 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    float bsum = 0.f;

    struct __attribute__((packed, aligned(8))) F4 { f32x4 v; };
    u32x4 st[NP];
    auto issue_quad = [&](int gs) {
        const int b = gs / 21, k = gs - b * 21;
        const long img = idx ? idx[b] : (long)b;
        const XT* src = x + img * (S1_C * S1_H * S1_H) + k * (4 * S1_H);
#pragma unroll
        for (int i = 0; i < NP - 1; ++i) st[i] = *reinterpret_cast<const u32x4*>(src + poff[i]);
        if (lane < LAST) st[NP - 1] = *reinterpret_cast<const u32x4*>(src + poff[NP - 1]);
    };
    auto write_quad = [&](int gs) {
        char* dst = wl + (gs % 3) * SLOT + 16 * lane;
#pragma unroll
        for (int i = 0; i < NP - 1; ++i) *reinterpret_cast<u32x4*>(dst + 1024 * i) = st[i];
        if (lane < LAST) *reinterpret_cast<u32x4*>(dst + 1024 * (NP - 1)) = st[NP - 1];
    };
    auto load_a = [&](int gs) {                                 // dY of the row step gs multiplies (k = 0: row 0, not used)
        const int b = gs / 21, k = gs - b * 21;
        const float* p = dy + (long)b * (S1_OC * S1_SP) + (k > 0 ? k - 1 : 0) * 20 + aoff;
        S9A a;
        a.p0 = reinterpret_cast<const F4*>(p)->v;
        a.p1 = reinterpret_cast<const F4*>(p + 4)->v;
        a.p2 = *reinterpret_cast<const f32x2*>(p + 8);
        return a;
    };
#ifdef S9_STAMP
    const long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (u0 < u1) {
        const int ul = u1 - 1;
        const int gs0 = (u0 / 20) * 21 + u0 % 20, gsl = (ul / 20) * 21 + ul % 20 + 1;
        // the first two quads and the first dY row are requested together (one memory latency, not two, before the
        // first MFMA; the second quad borrows registers the accumulators do not need yet)
        issue_quad(gs0);
        u32x4 st0[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) st0[i] = st[i];
        issue_quad(gs0 + 1);                                    // (gs0 + 1 <= gsl: the range holds at least one row)
        S9A nxt = load_a(gs0 + 1);
        {
            char* dst = wl + (gs0 % 3) * SLOT + 16 * lane;
#pragma unroll
            for (int i = 0; i < NP - 1; ++i) *reinterpret_cast<u32x4*>(dst + 1024 * i) = st0[i];
            if (lane < LAST) *reinterpret_cast<u32x4*>(dst + 1024 * (NP - 1)) = st0[NP - 1];
        }
        for (int gs = gs0 + 1; gs <= gsl; ++gs) {
            const int k = gs % 21;
            const S9A cur = nxt;
            // the quad of this step arrives from its register set; then the set and the dY registers are re-filled
            write_quad(gs);
            {
                const int gn = gs + 1 <= gsl ? gs + 1 : gsl;
                issue_quad(gn);
                nxt = load_a(gn);
            }
            if (k == 0) continue;                               // first quad of an image: nothing to multiply yet
            const char* row = wl + ((upper ? gs : gs - 1) % 3) * SLOT + rowoff;
            if constexpr (U8) {
                unsigned pw[11];
#pragma unroll
                for (int w = 0; w < 11; ++w) pw[w] = *reinterpret_cast<const unsigned*>(row + 4 * w);
#pragma unroll
                for (int s2 = 0; s2 < 10; ++s2) {
                    const float a = s9_a(cur, s2);
                    bsum += a;
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, (float)((pw[s2 + (q >> 2)] >> (8 * (q & 3))) & 0xffu), acc[q], 0, 0, 0);
                }
            } else {
                // the lane's eleven 16-byte pieces in two batches (six now, five once pieces 0 .. 3 are dead)
                f32x4 pc[11];
#pragma unroll
                for (int w = 0; w < 6; ++w) pc[w] = *reinterpret_cast<const f32x4*>(row + 16 * w);
#pragma unroll
                for (int s2 = 0; s2 < 10; ++s2) {
                    const float a = s9_a(cur, s2);
                    bsum += a;
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pc[s2 + (q >> 2)][q & 3], acc[q], 0, 0, 0);
                    if (s2 == 3) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int w = 6; w < 11; ++w) pc[w] = *reinterpret_cast<const f32x4*>(row + 16 * w);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }
#ifdef S9_STAMP
    if (lane == 0) {
        const int w = blockIdx.x * S9_WAVES + wave;
        s9_stamps[6 * w] = __builtin_amdgcn_s_memtime() - st_c0;
        s9_stamps[6 * w + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
        s9_stamps[6 * w + 2] = u1 - u0;
        s9_stamps[6 * w + 3] = st_in;
        s9_stamps[6 * w + 4] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // fixed-order sum of the eight waves' partials: ((w0+w4) + (w2+w6)) + ((w1+w5) + (w3+w7)), through the (dead) rings
    __syncthreads();
    auto dump = [&](int slot) {
        float* m = s9_lds + slot * (8 * 16 * 64);
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) m[(q * 16 + i) * 64 + lane] = acc[q][i];
    };
    auto add_in = [&](int slot) {
        const float* m = s9_lds + slot * (8 * 16 * 64);
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][i] += m[(q * 16 + i) * 64 + lane];
    };
    {
        const float t = bsum + __shfl_xor(bsum, 32);
        if (h == 0) s9_lds[4 * 8 * 16 * 64 + wave * 32 + li] = t;
    }
    if (wave >= 4) dump(wave - 4);
    __syncthreads();
    if (wave < 4) add_in(wave);
    __syncthreads();
    if (wave == 2 || wave == 3) dump(wave - 2);
    __syncthreads();
    if (wave < 2) add_in(wave);
    __syncthreads();
    if (wave == 1) dump(0);
    __syncthreads();
    float* slab = slabs + (long)blockIdx.x * slab_stride;
    if (wave == 0) {
        add_in(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int oc = (i & 3) + 8 * (i >> 2) + 4 * h;
            float* dst = slab + oc * S1_KC + li * 8;
            f32x4 v0, v1;
#pragma unroll
            for (int u = 0; u < 4; ++u) { v0[u] = acc[u][i] * scale; v1[u] = acc[4 + u][i] * scale; }
            *reinterpret_cast<f32x4*>(dst) = v0;
            *reinterpret_cast<f32x4*>(dst + 4) = v1;
        }
    } else if (wave == 1 && lane < 32) {
        const float* bs = s9_lds + 4 * 8 * 16 * 64;
        float t = bs[lane];
#pragma unroll
        for (int w = 1; w < S9_WAVES; ++w) t += bs[w * 32 + lane];
        slab[S1_OC * S1_KC + lane] = t;
    }
#ifdef S9_STAMP
    if (lane == 0) s9_stamps[6 * (blockIdx.x * S9_WAVES + wave) + 5] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Stationary-accumulator weight gradient of the 4x4 stride-2 layer on 20x20x32 activations (reference model.py:177, its
// autograd backward):  dW[oc, (c,ky,kx)] = scale * sum_{b,p} dY[b,oc,p] * X[b,c,2oy+ky,2ox+kx],  db[oc] = sum dY.
// Same construction as conv1_wgrad_sa_kernel; what differs:
//   * 64 oc = two MFMA row blocks, 512 taps = four channel groups of 8: a wave owns one group as four interleaved
//     column blocks (kx = the four floats of one 16-byte load, 8-byte aligned) x two row blocks = eight accumulators;
//   * 81 positions per image = ten blocks of 8 + one position (a single k-pair step whose upper k half is zeroed in A);
//     twelve virtual blocks walk a ring of three register slots, all offsets compile-time constants selected by h;
//     A is one 16-byte load (4-byte aligned: rows of 81 floats) per row block and slot, a single float in the last block;
//   * eight waves = 4 groups x 2 image sub-ranges; the second sub-range's partial sums go through LDS (128 KB), the first
//     adds them and writes the slab rows with 16-byte stores (a lane's four kx are adjacent).
constexpr int S2_C = 32, S2_HW = 400, S2_OC = 64, S2_SP = 81, S2_KC = 512, S2_WAVES = 4, S2_IMG = S2_C * S2_HW;
constexpr int s2_pos_off(int p) { return (p / 9) * 40 + (p % 9) * 2; }

struct S2Slot { f32x4 b[4]; f32x4 a[2]; };

__global__ __launch_bounds__(64 * S2_WAVES) void conv2_wgrad_sa_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                       float* __restrict__ slabs, long slab_stride, int B,
                                                                       float scale) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, h = lane >> 5;
    const int g = wave;
    const int img0 = (int)((long)B * blockIdx.x / gridDim.x), img1 = (int)((long)B * (blockIdx.x + 1) / gridDim.x);
    const int lbase = (8 * g + (li >> 2)) * S2_HW + (li & 3) * 20;     // channel, ky of this lane
    const int abase = li * S2_SP;                                      // dY row of this lane (row block 0; +32 rows for 1)
    f32x16 acc[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][q][i] = 0.f;
    float bsum[2] = {0.f, 0.f};
    S2Slot ring[3];
    auto issue = [&](auto VBc, S2Slot& sl, int img) {
        constexpr int vb = decltype(VBc)::value;
        const float* xi = x + (long)img * S2_IMG;
        const float* di = dy + (long)img * (S2_OC * S2_SP);
        if constexpr (vb < 10) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int po = h ? s2_pos_off(8 * vb + 4 + s) : s2_pos_off(8 * vb + s);
                sl.b[s] = *reinterpret_cast<const f32x4*>(xi + lbase + po);
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                struct __attribute__((packed, aligned(4))) U4 { f32x4 v; };
                sl.a[mb] = reinterpret_cast<const U4*>(di + abase + mb * (32 * S2_SP) + 8 * vb + 4 * h)->v;
            }
        } else if constexpr (vb == 10) {
            sl.b[0] = *reinterpret_cast<const f32x4*>(xi + lbase + s2_pos_off(80));
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) sl.a[mb][0] = di[abase + mb * (32 * S2_SP) + 80];
        }
    };
    auto consume = [&](auto VBc, S2Slot& sl) {
        constexpr int vb = decltype(VBc)::value;
        if constexpr (vb <= 10) {
#pragma unroll
            for (int s = 0; s < (vb == 10 ? 1 : 4); ++s) {
                float a0 = sl.a[0][s], a1 = sl.a[1][s];
                if constexpr (vb == 10) { a0 = h ? 0.f : a0; a1 = h ? 0.f : a1; }     // position 81 does not exist
                bsum[0] += a0;
                bsum[1] += a1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, sl.b[s][q], acc[0][q], 0, 0, 0);
                    acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, sl.b[s][q], acc[1][q], 0, 0, 0);
                }
            }
        }
    };
    auto step = [&](auto VBc, int img, int imgn) {
        constexpr int vb = decltype(VBc)::value;
        S2Slot& sl = ring[vb % 3];
        consume(VBc, sl);
        if constexpr (vb + 3 < 12) issue(std::integral_constant<int, vb + 3>{}, sl, img);
        else issue(std::integral_constant<int, vb + 3 - 12>{}, sl, imgn);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (img0 < img1) {
        issue(std::integral_constant<int, 0>{}, ring[0], img0);
        issue(std::integral_constant<int, 1>{}, ring[1], img0);
        issue(std::integral_constant<int, 2>{}, ring[2], img0);
        __builtin_amdgcn_sched_barrier(0);
        for (int img = img0; img < img1; ++img) {
            const int imgn = img + 1 < img1 ? img + 1 : img1 - 1;     // past the end: the last image again (never multiplied)
            step(std::integral_constant<int, 0>{}, img, imgn);
            step(std::integral_constant<int, 1>{}, img, imgn);
            step(std::integral_constant<int, 2>{}, img, imgn);
            step(std::integral_constant<int, 3>{}, img, imgn);
            step(std::integral_constant<int, 4>{}, img, imgn);
            step(std::integral_constant<int, 5>{}, img, imgn);
            step(std::integral_constant<int, 6>{}, img, imgn);
            step(std::integral_constant<int, 7>{}, img, imgn);
            step(std::integral_constant<int, 8>{}, img, imgn);
            step(std::integral_constant<int, 9>{}, img, imgn);
            step(std::integral_constant<int, 10>{}, img, imgn);
            step(std::integral_constant<int, 11>{}, img, imgn);
        }
    }
    // each wave writes its group's 128 columns of the slab: a lane's four kx are adjacent -> 16-byte stores
    float* slab = slabs + (long)blockIdx.x * slab_stride;
    const int col = ((8 * g + (li >> 2)) * 4 + (li & 3)) * 4;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int oc = 32 * mb + 4 * h + (i & 3) + 8 * (i >> 2);
            f32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = acc[mb][q][i] * scale;
            *reinterpret_cast<f32x4*>(slab + oc * S2_KC + col) = v;
        }
    if (g == 0) {                                    // every group saw all of dY: group 0 reports the bias sums
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            float v = bsum[mb];
            v += __shfl_xor(v, 32);
            if (h == 0) slab[S2_OC * S2_KC + 32 * mb + li] = v;
        }
    }
}

}  // namespace acktr
