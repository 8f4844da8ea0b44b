// Input gradient of NatureCNN's third conv layer (reference model.py:179: Conv2d(64, 32, 3, stride=1) on 9x9 maps;
// autograd of F.conv2d) in OVERLAP-ADD ("scatter") form, gfx950.
//
//   dX[b, ci, y+ky, x+kx] += sum_co W[co, ci, ky, kx] * dY[b, co, y, x]          dY: [B,32,7,7]   dX: [B,64,9,9]
//   dX *= (x_act > 0)                                                            (ReLU gate of the layer's input)
//
// The gather form (conv_sw.hpp, one dX pixel per lane over the zero-padded dY) multiplies the zero border: 81 pixels x 9
// taps = 729 tap products per image and channel pair where only 49 x 9 = 441 are non-zero -- 40 % of its MFMAs.  Here
// the GEMM is G[(ci, tap), p] = sum_co W^T[(ci, tap), co] dY[co, p] over the 49 REAL positions p of an image (M = 576,
// K = 32, N = 49 B: exactly the layer's FLOPs) and the nine 7x7 planes of a channel are added into the 9x9 plane at their
// tap's offset.
//
// Work split.  A workgroup (8 waves, two per SIMD) serves one quarter of the input channels (cb: 16 of ci): its 18 KB of
// W^T fragments stay in LDS.  A wave OWNS a contiguous range of images (two at minibatch 1024) and walks their 49 n
// positions in tiles of 16 (lane = (position j, k quarter kq)); per tile 8 loads of dY (channels co = 4 s + kq), 9
// accumulators (one per tap) x 8 k-steps = 72 v_mfma_f32_16x16x4_f32, and 9 read - add - write rounds that add the
// accumulators into the wave's private LDS image of dX[b, cb] (16 channels x 81 pixels at pitch 84; a ring of two images,
// since a tile straddles at most two).  The rounds of tile t - 1 sit BETWEEN the MFMAs of tile t (eight MFMAs cover an
// LDS round trip); the 32x32x2 form with one wave per SIMD and the rounds behind the MFMAs took 49 us, this one 31.
// Only the owning wave ever adds to an image, in program order: sums are in a fixed order, run to run.  When a tile
// completes an image the wave gates it with x_act (requested a pass earlier), stores dX -- and, optionally, the copy
// in the zero-padded plane layout the conv2 dgrad reads -- and zeroes the slot.
// Measured at minibatch 1024 (rocprofv3, same box): 30.9 us against 43-46 for the gather kernel; 3.0 us are the launch
// and the LDS fill, 20 us the seven multiplying passes (98 positions = 6.1 tiles; the MFMA pipe's own time for them is
// 13.4 us at 2.4 GHz, 16.9 at the 1.9 GHz the box runs this kernel at), 1.2 us the closing pass, 6.6 us the two flush
// bursts: every wave completes its images at the same two moments, and 32 MB (gate in, dX out twice) move per burst.
#pragma once
#include <type_traits>

#include "conv_sw.hpp"

namespace acktr {

typedef float oa_f4 __attribute__((ext_vector_type(4)));

constexpr int OA_CO = 32, OA_CI = 64, OA_CB = 16, OA_NCB = OA_CI / OA_CB, OA_P = 49, OA_OW = 7, OA_Q = 81, OA_IW = 9, OA_TAPS = 9;
constexpr int OA_TN = 16;                                  // positions per tile (v_mfma_f32_16x16x4_f32)
constexpr int OA_WIMG_CB = OA_TAPS * 2 * 64 * 4;          // floats of one channel quarter's fragments
constexpr int OA_IMG_FLOATS = OA_NCB * OA_WIMG_CB;        // 18432 = 32 * 64 * 9: a permutation of the weights
constexpr int OA_QP = 84;                                 // LDS pitch of a channel's 81 pixels: the four channel groups of a
                                                          // scatter instruction (4 kq + r) land 16 banks apart, not 4
constexpr int OA_NEL = OA_CB * OA_Q;                      // 1296 values: dX[b, cb] of one image
constexpr int OA_SLOT = OA_CB * OA_QP;                    // ... and its LDS slot
constexpr int OA_NT = 512, OA_WAVES = OA_NT / 64;
constexpr int OA_WAVE_FLOATS = 2 * OA_SLOT + 352;         // a wave's ring of two images + the dummy target of idle lanes
constexpr int OA_LDS_FLOATS = OA_WIMG_CB + OA_WAVES * OA_WAVE_FLOATS;
constexpr int OA_FL = (OA_NEL + 63) / 64;                // 21 elements per lane and image in the flush

// weight image: float index ((((cb*9 + tap)*2 + g)*64 + lane)*4 + u), lane = 16 kq + i, holds
//   W[co = 4 (4 g + u) + kq][ci = 16 cb + i][tap]                                  w: [32][64][3][3]
__device__ __forceinline__ void oa_pack_body(const float* __restrict__ w, float* __restrict__ img) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < OA_IMG_FLOATS; i += gridDim.x * 256) {
        const int u = i & 3, lane = (i >> 2) & 63, g = (i >> 8) & 1;
        const int tap = (i >> 9) % OA_TAPS, cb = (i >> 9) / OA_TAPS;
        const int co = 4 * (4 * g + u) + (lane >> 4), ci = OA_CB * cb + (lane & 15);
        img[i] = w[(co * OA_CI + ci) * OA_TAPS + tap];
    }
}

template <bool GATED>
__global__ __launch_bounds__(OA_NT) void conv3_dgrad_oa_kernel(const float* __restrict__ dy, const float* __restrict__ wimg,
                                                               const float* __restrict__ xact, float* __restrict__ dx,
                                                               float* __restrict__ dxp, int pw, int psz, int poff0, int B,
                                                               int gmain, const SwCarry carry) {
    if ((int)blockIdx.x >= gmain) {                       // a pending slab sum of the previous launch rides here (conv_sw.hpp)
        constexpr int VPB = OA_NT / 256;
        const int ncw = gridDim.x - gmain;
        for (int vb = ((int)blockIdx.x - gmain) * VPB + (threadIdx.x >> 8); vb < carry.nvb; vb += ncw * VPB)
            (void)slab_finalize_block(carry.fin, vb);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) float oa_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = blockIdx.x % OA_NCB, grp = blockIdx.x / OA_NCB, ngrp = gmain / OA_NCB;
    // image ranges: nr of them per channel quarter, dealt wave-major (wave 0 of every workgroup first), so a small batch
    // spreads over the CUs instead of filling a few of them
    const int nr = B < OA_WAVES * ngrp ? B : OA_WAVES * ngrp;
    const int s = wave * ngrp + grp;
    const bool active = s < nr;
    const int b0 = active ? (int)((long)s * B / nr) : 0, b1 = active ? (int)((long)(s + 1) * B / nr) : 1;
    const int nimg = b1 - b0, P = OA_P * nimg, ntile = (P + OA_TN - 1) / OA_TN;
    const int j = lane & 15, kq = lane >> 4;
    float* slots = oa_smem + OA_WIMG_CB + wave * OA_WAVE_FLOATS;
    const int dummy = 2 * OA_SLOT + lane;                  // where the lanes behind the range's last position add their zeros
    const oa_f4* afrag = reinterpret_cast<const oa_f4*>(oa_smem) + lane;
    const float* dyb = dy + (long)b0 * (OA_CO * OA_P) + kq * OA_P;
    const long xoff0 = ((long)b0 * OA_CI + cb * OA_CB) * OA_Q;

    auto tile_geom = [&](int t, bool& valid, int& goff, int& lbase) {
        const int pos = OA_TN * t + j;
        valid = pos < P;
        const int pc = valid ? pos : P - 1;
        const int im = pc / OA_P, p = pc - im * OA_P;
        const int y = p / OA_OW, x = p - y * OA_OW;
        goff = im * (OA_CO * OA_P) + p;
        lbase = valid ? (im & 1) * OA_SLOT + 4 * kq * OA_QP + y * OA_IW + x : dummy;
    };
    auto load_b = [&](float (&bv)[8], int goff) {
#pragma unroll
        for (int k = 0; k < 8; ++k) bv[k] = dyb[goff + 4 * k * OA_P];
        __builtin_amdgcn_sched_barrier(0);
    };

    float bv[2][8];
    bool valid; int goff, lbase;
    tile_geom(0, valid, goff, lbase);
    load_b(bv[0], goff);                                   // (flies over the LDS fill below)
    {   // this channel quarter's fragments -> LDS; all image slots of the workgroup start at zero
        const oa_f4* src = reinterpret_cast<const oa_f4*>(wimg + cb * OA_WIMG_CB);
        oa_f4* dst = reinterpret_cast<oa_f4*>(oa_smem);
        for (int i = tid; i < OA_WIMG_CB / 4; i += OA_NT) dst[i] = src[i];
        oa_f4* z = reinterpret_cast<oa_f4*>(oa_smem + OA_WIMG_CB);
        for (int i = tid; i < OA_WAVES * OA_WAVE_FLOATS / 4; i += OA_NT) z[i] = oa_f4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (!active) return;

    // One pass = the MFMAs of tile t and, between them, the overlap-add of tile t - 1 (nine read - add - write rounds, one
    // per tap, each LDS round trip under eight MFMAs), then the flush of the image tile t - 1 completed.
    // Overlap-add: tap (ky, kx) of position (y, x) lands on pixel (y + ky, x + kx); accumulator register r is channel
    // 4 kq + r.  Read - add - write, not ds_add_f32: the lanes of one instruction hit distinct addresses (distinct positions
    // of a tap are distinct pixels) and the LDS atomic runs at a fraction of the plain rate (measured: 180 us for the
    // kernel against 49).  Lanes behind the range's last position add zeros to a private dummy address, so that the
    // sequence is free of branches.
    oa_f4 acc[2][OA_TAPS];
    int lbase_p = dummy;
    int next_flush = 0;                                    // first image of the range not yet written out
    // pass t, accumulators acc[PAR] for tile t (MFMA = false in the range's last pass, t == ntile: nothing left to multiply)
    auto pass = [&](auto mfma_c, auto par_c, int t) {
        constexpr bool MFMA = decltype(mfma_c)::value;
        constexpr int PAR = decltype(par_c)::value;
        // operands of the next tile and the gate of the image tile t - 1 completed: both fly over this pass's MFMAs
        bool valid_n = false; int goff_n = goff, lbase_n = dummy;
        if constexpr (MFMA) {
            if (t + 1 < ntile) tile_geom(t + 1, valid_n, goff_n, lbase_n);
            load_b(bv[PAR ^ 1], goff_n);
        }
        const int done = (OA_TN * t) / OA_P;                // images complete after tile t - 1 (of the range, capped below)
        const int ndone = done < nimg ? done : nimg;
        const bool fl = next_flush < ndone;                 // at most one image completes per tile (49 > 16)
        float gate[GATED ? OA_FL : 1];
        if constexpr (GATED) {
            if (fl) {
                const float* gp = xact + xoff0 + (long)next_flush * (OA_CI * OA_Q) + lane;
#pragma unroll
                for (int i = 0; i < OA_FL; ++i) gate[i] = (i * 64 + lane < OA_NEL) ? gp[i * 64] : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float bsel[8];
        oa_f4 af[2][OA_TAPS];
        if constexpr (MFMA) {
#pragma unroll
            for (int k = 0; k < 8; ++k) bsel[k] = valid ? bv[PAR][k] : 0.f;
#pragma unroll
            for (int tp = 0; tp < OA_TAPS; ++tp) af[0][tp] = afrag[(tp * 2 + 0) * 64];
#pragma unroll
            for (int tp = 0; tp < OA_TAPS; ++tp) acc[PAR][tp] = oa_f4{0.f, 0.f, 0.f, 0.f};
        }
        float* lp = slots + lbase_p;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < OA_TAPS; ++c) {
            float* q = lp + (c / 3) * OA_IW + c % 3;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = q[r * OA_QP];
            if constexpr (MFMA) {
                if (c == 2) {                               // the second half of the K fragments: first used in round 4
#pragma unroll
                    for (int tp = 0; tp < OA_TAPS; ++tp) af[1][tp] = afrag[(tp * 2 + 1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 8 * c; m < 8 * c + 8; ++m) {
                    const int g = m / 36, u = (m / 9) % 4, tp = m % 9;
                    acc[PAR][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][tp][u], bsel[4 * g + u], acc[PAR][tp], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) q[r * OA_QP] = v[r] + acc[PAR ^ 1][c][r];
            // the next tap's reads hit pixels other LANES have just written: keep the program order (the LDS serves
            // one wave's requests in order; the compiler only sees per-lane addresses)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (fl) {
            // 1296 = 20 x 64 + 16 elements: the 21st round is for lanes 0..15 (the others read inside the wave's own
            // LDS region and store nothing).  Element e = 64 i + lane = ci * 81 + q sits at ci * 84 + q in the slot.
            float* ls = slots + (next_flush & 1) * OA_SLOT;
            const long xo = xoff0 + (long)next_flush * (OA_CI * OA_Q) + lane;
            const bool tail = lane < OA_NEL - 64 * (OA_FL - 1);
            int la[OA_FL], pa[OA_FL];                       // LDS index; offset inside the padded planes
            {
                int ci = lane / OA_Q, q = lane - ci * OA_Q;
#pragma unroll
                for (int i = 0; i < OA_FL; ++i) {
                    const int yy = (q * 57) >> 9;           // q / 9 for q < 81
                    la[i] = ci * OA_QP + q;
                    pa[i] = ci * psz + q + yy * (pw - OA_IW);
                    q += 64;
                    if (q >= OA_Q) { q -= OA_Q; ++ci; }
                }
            }
            float v[OA_FL];
#pragma unroll
            for (int i = 0; i < OA_FL; ++i) v[i] = ls[la[i]];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < OA_FL - 1; ++i) ls[la[i]] = 0.f;
            if (tail) ls[la[OA_FL - 1]] = 0.f;
            if constexpr (GATED) {
#pragma unroll
                for (int i = 0; i < OA_FL; ++i) v[i] = gate[i] > 0.f ? v[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < OA_FL - 1; ++i) dx[xo + i * 64] = v[i];
            if (tail) dx[xo + (OA_FL - 1) * 64] = v[OA_FL - 1];
            if (dxp) {                                      // the same values at [ci][yy + pad][xx + pad] of the padded planes
                float* dpl = dxp + ((long)(b0 + next_flush) * OA_CI + cb * OA_CB) * psz + poff0;
#pragma unroll
                for (int i = 0; i < OA_FL - 1; ++i) dpl[pa[i]] = v[i];
                if (tail) dpl[pa[OA_FL - 1]] = v[OA_FL - 1];
            }
            ++next_flush;
            __builtin_amdgcn_wave_barrier();
        }
        valid = valid_n; goff = goff_n; lbase_p = lbase; lbase = lbase_n;
    };
    using T_ = std::true_type; using F_ = std::false_type;
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    int t = 0;
#pragma unroll 1
    for (; t + 1 < ntile; t += 2) {
        pass(T_{}, P0{}, t);
        pass(T_{}, P1{}, t + 1);
    }
    if (t < ntile) {                                       // odd tile count: one more multiplying pass, then the closing one
        pass(T_{}, P0{}, t);
        pass(F_{}, P1{}, t + 1);
    } else {
        pass(F_{}, P0{}, t);
    }
}

template <bool GATED>
int launch_conv3_dgrad_oa(const float* dy, const float* wimg, const float* xact, float* dx, float* dxp, int pw, int psz,
                          int poff0, int B, hipStream_t stream, const acktr_slab_fin* carry) {
    constexpr int smem_bytes = OA_LDS_FLOATS * (int)sizeof(float);
    auto kern = conv3_dgrad_oa_kernel<GATED>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           smem_bytes);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(%d B LDS) failed: %s", smem_bytes, hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    // one workgroup per CU, four channel quarters: a multiple of four, and no more workgroups than image ranges need
    int grid = sw_num_cus() / OA_NCB * OA_NCB;
    if (grid < OA_NCB) grid = OA_NCB;
    if (grid > OA_NCB * B) grid = OA_NCB * B;
    SwCarry cy;
    cy.fin.dst = nullptr; cy.fin.slabs = nullptr; cy.fin.nslabs = 0; cy.fin.n4 = 0; cy.fin.stride = 0; cy.fin.scale = 1.f;
    cy.nvb = 0;
    int extra = 0;
    if (carry && carry->dst) {
        cy.fin = *carry;
        cy.nvb = (int)((carry->n4 + 15) / 16);
        extra = cdiv(cy.nvb, OA_NT / 256);
        if (extra > 128) extra = 128;
    }
    hipLaunchKernelGGL(kern, dim3(grid + extra), dim3(OA_NT), smem_bytes, stream, dy, wimg, xact, dx, dxp, pw, psz, poff0, B,
                       grid, cy);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

}  // namespace acktr
