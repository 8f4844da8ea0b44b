// Stationary-weight convolution kernel (gfx950): forward and dgrad of the policy's conv layers.
//
//   out(m, r) = sum_{c, ky, kx} Wm(m; c, ky, kx) * X[img(r), c, S*oy(r) + ky, S*ox(r) + kx]
//
// The GEMM is tall and skinny the "wrong" way round: M (output channels, 32..128) is tiny, N (positions,
// ~10^5) is huge, K (256..576) moderate.  So the WEIGHTS (<= 128 KB) stay in LDS for the lifetime of a
// workgroup, pre-permuted into MFMA A-fragment order, and the position operand never touches LDS: each lane
// loads the 2..4 contiguous kernel-row taps of ITS position straight from global memory into the register
// that the MFMA consumes (the B operand of v_mfma_f32_32x32x2_f32 wants lane -> column anyway).  Consequences:
//   * no barrier in the main loop: the waves of a workgroup stream independent 32-position tiles
//   * one 8/12/16-byte global load + TM LDS reads per E*TM MFMAs, and almost no address arithmetic
//     (lane base pointer per tile; every chunk offset is a compile-time constant)
//   * every activation is fetched exactly once per tap row by exactly one lane
// K order: the reduction index is split into "chunks" (c, ky, 4-wide piece of kx); the MFMA's two k-lanes
// (lane halves h = 0, 1) take the two chunks of a chunk PAIR, which differ by a constant address offset (see
// the HM_* modes), so both halves use the same constants and differ only in their base pointer.  Sums are fp32
// FMA chains in a fixed order.
//
// dgrad runs through the same kernel: dX of a stride-S layer is S*S stride-1 correlations of the zero-padded
// dY with (KH/S x KW/S) sub-kernels; the S*S phases share the position operand and are stacked along M, so a
// lane ends up holding horizontally adjacent dX pixels and stores them as one 8-byte write.
#pragma once
#include <utility>

#include "gemm_core.hpp"
#include "head_finalize.hpp"

namespace acktr {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
template <int E> struct TapVec;
template <> struct TapVec<2> { typedef f32x2 type; typedef f32x2 utype __attribute__((aligned(4))); };
template <> struct TapVec<3> { typedef f32x3 type; typedef f32x3 utype __attribute__((aligned(4))); };
template <> struct TapVec<4> { typedef f32x4 type; typedef f32x4 utype __attribute__((aligned(4))); };

// How the MFMA's two k-lanes (lane halves h = 0, 1) split a chunk pair:
//   HM_CH: h picks the channel half (c = cp + h*C/2)         any kernel
//   HM_KX: h picks the 4-wide piece of an 8-wide kernel row   (both halves of a load hit the same cache lines)
//   HM_KY: h picks the parity of the kernel row ky = 2*kyp+h  (adjacent rows: mostly the same cache lines)
// Chunk pairs are ordered CHANNEL-FASTEST: j = g*CP + cp (g = kernel row group, cp = channel).  Consecutive loads
// of a lane then touch different channel planes, and the next kernel row of the same plane -- which lives in the
// same 128-byte lines -- is requested CP loads later, when the line has arrived.  (With the rows of one channel
// requested back to back the L1 sat in hit-on-miss stalls for 2/3 of the kernel: TCP_PENDING_STALL_CYCLES.)
enum { HM_CH = 0, HM_KX = 1, HM_KY = 2 };

// NT_ threads per workgroup (one workgroup per CU): 512 -> 2 waves per SIMD with a 256-register budget.
// CH_ input channels are compile time too: the chunk loop is fully unrolled, which makes every chunk offset a
// constant and lets the compiler count outstanding loads exactly (a run-time stage loop made it drain the
// prefetch with s_waitcnt vmcnt(0) at every loop header).
// TM_ 32-row M blocks are resident in LDS; one work item (a wave's unit of work) is a 32-position tile times
// TMW_ of them.  TMW_ < TM_ re-reads the activations TM_/TMW_ times (from L1/L2) but makes the items small
// enough to balance: at minibatch 1024 there are only 1.3 - 1.6 full tiles per wave.
// KSPLIT_ (0 or 4): a workgroup whose share of the last, partial round is ONE item more than a multiple of four -- a second
// tail item for one of its four SIMDs while three waves of the workgroup have none -- cuts that item's reduction into four
// parts, one per SIMD, and sums them through LDS in a fixed order (see conv_sw_kernel).
template <int KH_, int KW_, int S_, int H_, int W_, int CH_, int TM_, int TMW_, int NL_, int HM_, int NT_ = 512,
          bool BY_WG_ = true, int KSPLIT_ = 0>
struct SwGeom {
    static constexpr int KSPLIT = KSPLIT_;
    // BY_WG: the waves of a workgroup take CONSECUTIVE work items (neighbouring tiles of the same images and both M
    // groups of a tile share the CU's L1 and the XCD's L2) instead of items a whole grid apart
    static constexpr bool BY_WG = BY_WG_;
    static constexpr int KH = KH_, KW = KW_, S = S_, H = H_, W = W_, CH = CH_, TM = TM_, NL = NL_, HM = HM_;
    static constexpr int TMW = TMW_, MG = TM_ / TMW_;
    static_assert(TM_ % TMW_ == 0, "M blocks per work item");
    static constexpr int NT = NT_, NWAVES = NT_ / 64;
    static_assert(KW <= 4 || KW % 4 == 0, "kernel rows are cut into 4-wide chunks");
    static constexpr int NKXC = (KW + 3) / 4;           // chunks per kernel row
    static constexpr int E = KW >= 4 ? 4 : KW;          // taps per chunk (= MFMA k-steps per chunk)
    static constexpr int EP = E == 3 ? 4 : E;           // floats per LDS fragment slot
    static constexpr int OH = (H - KH) / S + 1, OW = (W - KW) / S + 1, SP = OH * OW, HW = H * W;
    static_assert(HM != HM_CH || CH % 2 == 0, "channel halves");
    static_assert(HM != HM_KX || NKXC == 2, "kx halves need an 8-wide kernel row");
    static_assert(HM != HM_KY || KH % 2 == 0, "ky parity needs an even kernel height");
    static constexpr int CP = HM == HM_CH ? CH / 2 : CH;                                   // channels per row group
    static constexpr int NG = HM == HM_CH ? KH * NKXC : (HM == HM_KX ? KH : (KH / 2) * NKXC);   // row groups
    static constexpr int J = NG * CP;                                                      // chunk pairs
    static constexpr int HOFF = HM == HM_CH ? (CH / 2) * HW : (HM == HM_KX ? 4 : W);       // h = 1 lanes' extra offset
    // element offset (floats, h = 0) of chunk pair j relative to the lane's window origin
    __host__ __device__ static constexpr int off0(int j) {
        const int g = j / CP, cp = j % CP;
        return cp * HW + (HM == HM_CH ? (g / NKXC) * W + (g % NKXC) * 4
                                      : (HM == HM_KX ? g * W : 2 * (g / NKXC) * W + (g % NKXC) * 4));
    }
    // (c, ky, kx0) of chunk pair j for half h  (used by the weight packer)
    __host__ __device__ static constexpr int dec_c(int j, int h) { return j % CP + (HM == HM_CH ? h * (CH / 2) : 0); }
    __host__ __device__ static constexpr int dec_ky(int j, int h) {
        const int g = j / CP;
        return HM == HM_CH ? g / NKXC : (HM == HM_KX ? g : 2 * (g / NKXC) + h);
    }
    __host__ __device__ static constexpr int dec_kx0(int j, int h) {
        const int g = j / CP;
        return HM == HM_KX ? 4 * h : (g % NKXC) * 4;
    }
    static_assert(J % (2 * NL) == 0, "even number of stages (the two activation buffers alternate across tiles)");
    static_assert(J % 4 == 0, "A-fragment ring slots must line up across tiles");
    static constexpr int NS = J / NL;
    static constexpr long IMAGE_FLOATS = (long)J * TM * 64 * EP;
    static_assert(IMAGE_FLOATS * 4 <= 160 * 1024, "weights must fit in LDS");
    static bool ok(int C, int M) { return C == CH && M <= TM * 32; }
    __host__ __device__ static constexpr long image_floats() { return IMAGE_FLOATS; }
    // K split: the parts' chunk pairs [q J/4, (q+1) J/4) must be the first part's, shifted by a constant offset
    __host__ __device__ static constexpr bool ksplit_shift_ok() {
        if (KSPLIT == 0) return true;
        for (int q = 0; q < KSPLIT; ++q)
            for (int i = 0; i < J / KSPLIT; ++i)
                if (off0(q * (J / KSPLIT) + i) != off0(q * (J / KSPLIT)) + off0(i)) return false;
        return true;
    }
    static_assert(KSPLIT == 0 || (KSPLIT == 4 && NT_ == 512 && J % (4 * KSPLIT) == 0 && NS % (2 * KSPLIT) == 0),
                  "K split: four parts over the waves of a SIMD row, each an even number of stages and whole ring turns");
    static constexpr long KS_SCRATCH_FLOATS = KSPLIT ? (long)(KSPLIT - 1) * TMW_ * 16 * 64 : 0;   // the helpers' partial tiles
    static_assert((IMAGE_FLOATS + KS_SCRATCH_FLOATS) * 4 <= 160 * 1024, "weights + K-split scratch must fit in LDS");
    static constexpr long lds_bytes() { return (IMAGE_FLOATS + KS_SCRATCH_FLOATS) * 4; }
};

// ---------------------------------------------------------------------------------------------------
// weight image: float index (((j*TM + a)*2 + h)*32 + li)*EP + t  holds  Wm(m = 32a+li; chunk pair j, half h, tap t)
//   mode 0 (forward): Wm(m; c,ky,kx) = w[m][c][ky][kx]                                   w: [OC][C][KH][KW]
//   mode 1 (dgrad)  : m = (py*SD+px)*CX + cx,  c = oc,
//                     Wm = w[oc][cx][py + SD*(KH-1-ky)][px + SD*(KW-1-kx)]               w: [OC][CX][KH*SD][KW*SD]
// ---------------------------------------------------------------------------------------------------
template <class G>
__device__ __forceinline__ void sw_pack_body(const float* __restrict__ w, float* __restrict__ img, int M, int mode, int CX,
                                             int SD) {
    constexpr long n = G::image_floats();
    constexpr int C = G::CH;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int t = (int)(i % G::EP);
        long q = i / G::EP;
        const int li = (int)(q % 32); q /= 32;
        const int h = (int)(q % 2); q /= 2;
        const int a = (int)(q % G::TM);
        const int j = (int)(q / G::TM);
        const int m = 32 * a + li;
        const int c = G::dec_c(j, h), ky = G::dec_ky(j, h), kx = G::dec_kx0(j, h) + t;
        float v = 0.f;
        if (m < M && t < G::E && kx < G::KW) {
            if (mode == 0) {
                v = w[(((long)m * C + c) * G::KH + ky) * G::KW + kx];
            } else {
                const int ph = m / CX, cx = m % CX;
                const int py = ph / SD, px = ph % SD;
                const int wy = py + SD * (G::KH - 1 - ky), wx = px + SD * (G::KW - 1 - kx);
                v = w[(((long)c * CX + cx) * (G::KH * SD) + wy) * (G::KW * SD) + wx];
            }
        }
        img[i] = v;
    }
}

template <class G>
__global__ __launch_bounds__(256) void sw_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ img,
                                                              int M, int mode, int CX, int SD) {
    sw_pack_body<G>(w, img, M, mode, CX, SD);
}

// zero-padded copy dyp[pl][y+P][x+P] = dy[pl][y][x]  (planes OH x OW -> HP x WP); 8 planes per workgroup,
// compile-time plane geometry so the index split is multiply-shift
template <int OH, int OW, int P>
__global__ __launch_bounds__(256) void sw_pad_kernel(const float* __restrict__ dy, float* __restrict__ dyp, long planes) {
    constexpr int HP = OH + 2 * P, WP = OW + 2 * P, PP = HP * WP, PPB = 8;
    const long pl0 = (long)blockIdx.x * PPB;
    for (int i = threadIdx.x; i < PPB * PP; i += 256) {
        const int pl = i / PP, e = i % PP;
        if (pl0 + pl >= planes) break;
        const int yy = e / WP - P, xx = e % WP - P;
        float v = 0.f;
        if (xx >= 0 && xx < OW && yy >= 0 && yy < OH) v = dy[((pl0 + pl) * OH + yy) * OW + xx];
        dyp[(pl0 + pl) * PP + e] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// epilogues: lane = position r (valid or not), half h, acc[a][reg] -> row m = 32a + (reg&3) + 8(reg>>2) + 4h
// ---------------------------------------------------------------------------------------------------
// lane LANE of `old` := the wave-uniform value `val` (v_writelane_b32: scalar source, immediate lane select)
template <int LANE>
__device__ __forceinline__ int sw_writelane(int val, int old) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(val), "n"(LANE));
    return old;
}
template <class F, int... I>
__device__ __forceinline__ void sw_static_for(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

template <int SP, bool BIASED>
struct SwFwdEpi {       // Y[b, m, sp] = relu?(acc*scale + bias[m])
    float* y; const float* bias; int M; float scale; int relu;
    // optional: the ReLU gate of Y as bits -- word [r / 32][m] holds (Y > 0) of the 32 positions r of a tile (r = b*SP + sp),
    // which is all the layer's input gradient needs of Y (52 MB as floats, 1.6 MB as bits for conv1 at 1024 frames)
    unsigned* gmask = nullptr;
    template <int TM> struct Pre { float bv[BIASED ? TM : 1][16]; };
    template <int TM>
    __device__ __forceinline__ void prefetch(int, bool, int h, int mg, Pre<TM>& p) const {   // (see SwGateEpi)
        if constexpr (BIASED) {
            const int m0 = 32 * TM * mg;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int m = m0 + 32 * a + (g & 3) + 8 * (g >> 2) + 4 * h;
                    p.bv[a][g] = bias[m < M ? m : M - 1];
                }
        }
    }
    template <int TM>
    __device__ __forceinline__ void store(int r, bool valid, int h, int mg, const f32x16 (&acc)[TM], const Pre<TM>& p) const {
        const long off = (long)(r / SP) * M * SP + (r % SP);
        const bool full = (M % 32 == 0);                  // wave-uniform: no per-row bound checks
        const int m0 = 32 * TM * mg;
        if (m0 >= M) return;
        // GM: also the gate bits.  Every lane of the wave is here and r = 32*tile + li, so r >> 5 is the tile.  The ballot of
        // row g (a scalar pair: low word = the h = 0 lanes' row, high word = the h = 1 lanes') goes into lane g of each
        // half with v_writelane, so that after the sixteen rows lanes 0..15 / 32..47 hold the 32 words of the tile: ONE
        // store into one 128-byte line.  (First version: sixteen 4-byte stores of two lanes each, +5 us on the conv1 forward;
        // second: values recomputed in a loop of their own and a compare + select per row, +2.2 us.)
        auto body = [&](auto gm_c) {
            constexpr bool GM = decltype(gm_c)::value;
            const unsigned long long vmask = GM ? __ballot(valid) : 0ull;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                int word = 0;
                sw_static_for([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    const int m = m0 + 32 * a + (g & 3) + 8 * (g >> 2) + 4 * h;
                    float v = acc[a][g] * scale;
                    if constexpr (BIASED) v += p.bv[a][g];
                    if (relu) v = v > 0.f ? v : 0.f;
                    if (valid && (full || m < M)) y[off + (long)m * SP] = v;
                    if constexpr (GM) {
                        const unsigned long long bal = __ballot(v > 0.f) & vmask;
                        word = sw_writelane<g>((int)(unsigned)bal, word);
                        word = sw_writelane<32 + g>((int)(unsigned)(bal >> 32), word);
                    }
                }, std::make_integer_sequence<int, 16>{});
                if constexpr (GM) {
                    const int li = threadIdx.x & 31;
                    const int m = m0 + 32 * a + (li & 3) + 8 * (li >> 2) + 4 * h;
                    if (li < 16 && (full || m < M)) gmask[(long)(r >> 5) * M + m] = (unsigned)word;
                }
            }
        };
        if (gmask) body(std::true_type{});
        else body(std::false_type{});
    }
};

// Epilogues with a gate read it in prefetch(), which the kernel calls one stage of MFMAs before the tile is finished
// (issued in store() the 16 gate loads of a lane were latency nothing overlapped: 8 us of the conv3 dgrad's 52).  The
// loads are unconditional (indices clamped) and GATED is a compile-time switch: behind a run-time branch the compiler
// can no longer count the loads in flight.
template <int SP, bool GATED>
struct SwGateEpi {      // stride-1 dgrad: dX[b, m, sp] = acc * (xact > 0)
    float* dx; const float* xact; int M;
    // optional second copy of dX in the zero-padded plane layout the NEXT layer's dgrad reads ([b][m][psz] planes of
    // rows pw long, interior origin at poff0, the layer's own rows ow long): saves that layer's padding pass
    float* dxp; int ow, pw, psz, poff0;
    template <int TM> struct Pre { float gv[GATED ? TM : 1][16]; };
    template <int TM>
    __device__ __forceinline__ void prefetch(int r, bool valid, int h, int mg, Pre<TM>& p) const {
        if constexpr (GATED) {
            const int rr = valid ? r : 0;
            const long off = (long)(rr / SP) * M * SP + (rr % SP);
            const int m0 = 32 * TM * mg;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int m = m0 + 32 * a + (g & 3) + 8 * (g >> 2) + 4 * h;
                    p.gv[a][g] = xact[off + (long)(m < M ? m : M - 1) * SP];
                }
        }
    }
    template <int TM>
    __device__ __forceinline__ void store(int r, bool valid, int h, int mg, const f32x16 (&acc)[TM], const Pre<TM>& p) const {
        const int m0 = 32 * TM * mg;
        if (!valid || m0 >= M) return;
        const long off = (long)(r / SP) * M * SP + (r % SP);
        const long offp = dxp ? (long)(r / SP) * M * psz + ((r % SP) / ow) * pw + ((r % SP) % ow) + poff0 : 0;
        const bool full = (M % 32 == 0);
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int m = m0 + 32 * a + (g & 3) + 8 * (g >> 2) + 4 * h;
                float v = acc[a][g];
                if constexpr (GATED) v = p.gv[a][g] > 0.f ? v : 0.f;
                if (full || m < M) {
                    dx[off + (long)m * SP] = v;
                    if (dxp) dxp[offp + (long)m * psz] = v;
                }
            }
        }
    }
};

// stride-2 dgrad, CX == 32: M block = phase (py, px), one work item = the two px phases of row phase py = mg;
// position r = (b, jy, jx) of the JH x JW phase grid; dX[b, cx, 2jy+py, 2jx + {0,1}] leaves as one 8-byte store
// ROWCLS: the kernel deals the positions of the phase grid by ROW CLASS (see conv_sw_kernel): offset() below is unchanged,
// r is still b*JP + jy*JW + jx.
// GATED: 0 = no gate, 1 = xact holds the layer's input (floats, gate = x > 0), 2 = xact holds the gate as bits in the layout
// SwFwdEpi::gmask writes (word [R1 / 32][cx], R1 = b*IH*IW + pixel): the 16 channel words of a lane are four 16-byte loads
// out of ONE 128-byte line instead of sixteen 8-byte loads out of sixteen planes.
template <int JH, int JW, int IH, int IW, int GATED, bool ROWCLS_ = false>
struct SwPhase2Epi {
    float* dx; const float* xact;
    static constexpr bool ROWCLS = ROWCLS_;
    static constexpr int RC_JH = JH, RC_JW = JW;
    static constexpr int JP = JH * JW;
    __device__ __forceinline__ static long offset(int r, int mg) {
        const int b = r / JP, j = r % JP;
        const int jy = j / JW, jx = j % JW;
        return (long)b * 32 * (IH * IW) + (2 * jy + mg) * IW + 2 * jx;
    }
    typedef unsigned int gw4 __attribute__((ext_vector_type(4)));
    template <int TM> struct Pre { f32x2 gt[GATED == 1 ? 16 : 1]; gw4 gw[GATED == 2 ? 4 : 1]; };
    __device__ __forceinline__ static int pixel_index(int r, int mg) {      // R1 of the lane's left pixel (even)
        const int b = r / JP, j = r % JP;
        return b * (IH * IW) + (2 * (j / JW) + mg) * IW + 2 * (j % JW);
    }
    template <int TM>
    __device__ __forceinline__ void prefetch(int r, bool valid, int h, int mg, Pre<TM>& p) const {
        if constexpr (GATED == 2) {
            const int R1 = pixel_index(valid ? r : 0, mg);
            const gw4* gm = reinterpret_cast<const gw4*>(reinterpret_cast<const unsigned*>(xact) + (long)(R1 >> 5) * 32 + 4 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) p.gw[q] = gm[2 * q];          // channels 8q + 4h .. + 3
        }
        if constexpr (GATED == 1) {
            const long off = offset(valid ? r : 0, mg);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int cx = (g & 3) + 8 * (g >> 2) + 4 * h;
                p.gt[g] = *reinterpret_cast<const f32x2*>(xact + off + (long)cx * (IH * IW));
            }
        }
    }
    template <int TM>
    __device__ __forceinline__ void store(int r, bool valid, int h, int mg, const f32x16 (&acc)[TM], const Pre<TM>& p) const {
        static_assert(TM == 2, "one work item = the two horizontal phases of one row phase");
        static_assert(IH % 2 == 0 && IW % 2 == 0 && 2 * JH == IH && 2 * JW == IW, "even phase grid");
        if (!valid) return;
        const long off = offset(r, mg);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int cx = (g & 3) + 8 * (g >> 2) + 4 * h;
            f32x2 v;
            v[0] = acc[0][g];
            v[1] = acc[1][g];
            if constexpr (GATED == 1) {
                v[0] = p.gt[g][0] > 0.f ? v[0] : 0.f;
                v[1] = p.gt[g][1] > 0.f ? v[1] : 0.f;
            }
            if constexpr (GATED == 2) {
                const unsigned wbits = p.gw[g >> 2][g & 3] >> (pixel_index(r, mg) & 31);
                v[0] = (wbits & 1u) ? v[0] : 0.f;
                v[1] = (wbits & 2u) ? v[1] : 0.f;
            }
            *reinterpret_cast<f32x2*>(dx + off + (long)cx * (IH * IW)) = v;
        }
    }
};

// The position operand's element type XT: float, or uint8_t for frames kept as bytes on the device (rollout storage
// with obs_dtype=torch.uint8).  A lane's four kx taps are then ONE aligned 4-byte load, and the
// bytes become floats (v_cvt_f32_ubyte0..3, exact) where the MFMA consumes them -- not at load time, which would make
// the wave wait for the load on the spot.  Same products in the same order: bit-identical to the float frames.
template <class XT, int E> struct SwRaw { typedef typename TapVec<E>::type type; typedef typename TapVec<E>::utype utype; };
template <> struct SwRaw<uint8_t, 4> { typedef uint32_t type; typedef uint32_t utype; };
__device__ __forceinline__ float sw_tap(const f32x2& v, int t) { return v[t]; }
__device__ __forceinline__ float sw_tap(const f32x3& v, int t) { return v[t]; }
__device__ __forceinline__ float sw_tap(const f32x4& v, int t) { return v[t]; }
__device__ __forceinline__ float sw_tap(uint32_t v, int t) { return (float)((v >> (8 * t)) & 0xffu); }

// ---------------------------------------------------------------------------------------------------
// main kernel
// ---------------------------------------------------------------------------------------------------
// A pending slab sum of the PREVIOUS launch (a weight gradient whose workgroup slabs are written but not yet added up,
// acktr_slab_fin) can ride in this launch: workgroups [gmain, gridDim.x) do nothing but that sum.  They are numbered
// behind the one-per-CU main workgroups, so they start on the CUs whose main workgroup has run out of items -- the
// last, partial round of work items leaves most CUs idle for an item's time -- and the sum's own launch (6 us as a
// kernel of its own: a dependent launch, a ramp, 19-34 MB of slabs) disappears from the step.
struct SwCarry { acktr_slab_fin fin; int nvb; };      // nvb: virtual 256-thread blocks of slab_finalize_block

// Row classes (stride-2 dgrad on the zero-padded dY, KH = 2): in the top row of the phase grid (jy = 0) the kernel row
// ky = 0 reads nothing but the padding, in the bottom row (jy = JH - 1) the row ky = 1 does -- 10 % of the launch's MFMAs
// multiply zeros there (and as many again in the first / last column, which stay).  With ROWCLS the tiles are built per
// class: middle rows (full reduction), top rows (only the ky = 1 half of the chunk pairs: the same offsets one row further
// down, the second half of the weight image), bottom rows (only the ky = 0 half).  The half items are numbered LAST, so
// they also make the last rounds lighter: conv2 dgrad 74.1 -> 69.0 us inside the PPO step at 1024 rows.
// Tried and dropped: (a) dealing the last full round wave-major as well, which evens the tail out to 6.0 item-times on the
// busiest SIMD for 5.63 on average, lost more in L1 / L2 sharing than it gained (87.0 against 81.8 us stand-alone);
// (b) column classes too (first / last column of the middle rows with one of a chunk's two taps): a tile of 32 positions
// out of ONE column is 32 different rows of 4 images, every load and store of it touches 32 cache lines: 100 us;
// (c) a mixed round (every workgroup's lower four waves a full item, its upper four a half item, so that all SIMDs end
// within half an item of each other): 71.7 against 69.0 us -- runs of consecutive items per workgroup are worth more.
template <class Epi, class = void>
struct EpiRowCls : std::false_type {};
template <class Epi>
struct EpiRowCls<Epi, std::enable_if_t<Epi::ROWCLS>> : std::true_type {};
template <class G, class Epi>
__host__ __device__ inline int sw_num_tiles(int R) {
    if constexpr (EpiRowCls<Epi>::value) {
        const int B = R / (Epi::RC_JH * Epi::RC_JW);
        return (B * (Epi::RC_JH - 2) * Epi::RC_JW + 31) / 32 + 2 * ((B * Epi::RC_JW + 31) / 32);
    } else {
        return (R + 31) / 32;
    }
}

template <class G, class Epi, class XT = float>
__global__ __launch_bounds__(G::NT) void conv_sw_kernel(const XT* __restrict__ x,
                                                             const int64_t* __restrict__ idx,
                                                             const float* __restrict__ wimg, const Epi epi, int R,
                                                             int ntiles, int gmain, const SwCarry carry) {
    if ((int)blockIdx.x >= gmain) {
        constexpr int VPB = G::NT / 256;                     // virtual blocks per workgroup pass
        const int ncw = gridDim.x - gmain;
        for (int vb = ((int)blockIdx.x - gmain) * VPB + (threadIdx.x >> 8); vb < carry.nvb; vb += ncw * VPB)
            (void)slab_finalize_block(carry.fin, vb);
        return;
    }
    using VT = typename SwRaw<XT, G::E>::type;
    using VU = typename SwRaw<XT, G::E>::utype;
    using AT = typename TapVec<G::EP>::type;
    constexpr int TM = G::TM, TMW = G::TMW, MG = G::MG, NL = G::NL, NS = G::NS, C = G::CH;
    extern __shared__ __attribute__((aligned(16))) float sw_smem[];

    const int tid = threadIdx.x;
    {   // resident weights: global image -> LDS (16-byte copies, 8 loads in flight per thread)
        constexpr int n4 = (int)(G::image_floats() / 4);
        const f32x4* src = reinterpret_cast<const f32x4*>(wimg);
        f32x4* dst = reinterpret_cast<f32x4*>(sw_smem);
        constexpr int UN = 8;
        for (int i0 = tid; i0 < n4; i0 += UN * G::NT) {
            f32x4 t[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) { const int i = i0 + u * G::NT; t[u] = i < n4 ? src[i] : zero4(); }
#pragma unroll
            for (int u = 0; u < UN; ++u) { const int i = i0 + u * G::NT; if (i < n4) dst[i] = t[u]; }
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const float* lds_lane = sw_smem + (h * 32 + li) * G::EP;
    constexpr long img_stride = (long)C * G::HW;
    const int half_off = h * G::HOFF;

    constexpr bool RC = EpiRowCls<Epi>::value;
    static_assert(!RC || (G::KH == 2 && G::NKXC == 1 && G::HM == HM_CH && NS % 2 == 0 && (G::J / 2) % 4 == 0),
                  "row classes: two kernel rows, each one half of the chunk pairs");
    int rc_mid = 0, rc_edge = 0, rc_B = 0;                   // tiles of the middle rows / of one edge row; images
    if constexpr (RC) {
        rc_B = R / (Epi::RC_JH * Epi::RC_JW);
        rc_mid = (rc_B * (Epi::RC_JH - 2) * Epi::RC_JW + 31) / 32;
        rc_edge = (rc_B * Epi::RC_JW + 31) / 32;
    }
    // tile -> the lane's position r, its window origin, and (row classes) which half of the reduction the item runs:
    // nst stages starting at activation offset xoff / weight-image offset woff
    auto lane_base = [&](int tile, int& r, bool& valid, int& nst, int& woff) -> const XT* {
        int xoff = 0;
        nst = NS; woff = 0;
        if constexpr (RC) {
            constexpr int JH = Epi::RC_JH, JW = Epi::RC_JW, MIDP = (JH - 2) * JW;
            if (tile < rc_mid) {
                const int e = tile * 32 + li;
                valid = e < rc_B * MIDP;
                const int ee = valid ? e : 0, b = ee / MIDP, q = ee % MIDP;
                r = b * (JH * JW) + JW + q;
            } else {
                const bool top = tile < rc_mid + rc_edge;
                const int e = (tile - rc_mid - (top ? 0 : rc_edge)) * 32 + li;
                valid = e < rc_B * JW;
                const int ee = valid ? e : 0, b = ee / JW, jx = ee % JW;
                r = b * (JH * JW) + (top ? 0 : (JH - 1) * JW) + jx;
                nst = NS / 2;
                if (top) { xoff = G::W; woff = (G::J / 2) * TM * 64 * G::EP; }     // ky = 1: one row down, second half of the image
            }
        } else {
            r = tile * 32 + li;
            valid = r < R;
        }
        const int rr = valid ? r : R - 1;
        const int b = rr / G::SP, sp = rr % G::SP;
        const int oy = sp / G::OW, ox = sp % G::OW;
        const long img = idx ? (long)idx[b] : (long)b;
        return x + img * img_stride + (oy * G::S) * G::W + ox * G::S + half_off + xoff;
    };
    auto stage_load = [&](VT (&buf)[NL], const XT* base, int s) {
#pragma unroll
        for (int i = 0; i < NL; ++i) buf[i] = *reinterpret_cast<const VU*>(base + G::off0(s * NL + i));
        __builtin_amdgcn_sched_barrier(0);     // the compiler otherwise sinks these loads below the MFMAs they overlap
    };
    // A fragments run through a 4-slot register ring, fetched from LDS two chunks ahead of their MFMAs (left to
    // itself the compiler issues each ds_read right in front of its consumer and exposes the LDS latency)
    constexpr int J = G::J;
    AT ring[4][TMW];
    auto load_a = [&](int q, const float* la) {       // q: chunk pair index within the item (compile time after unrolling)
#pragma unroll
        for (int a = 0; a < TMW; ++a)
            ring[q & 3][a] = *reinterpret_cast<const AT*>(la + (q * TM + a) * 64 * G::EP);
    };

    // work items (tile, M group) are dealt round-robin over all waves of the grid: every CU gets the same number of
    // items +-1 however few there are.  Which wave takes which item in a FULL round: SwGeom::BY_WG (measured per layer:
    // +9 % on the conv2 forward, +4-8 % on conv3 and the dgrads, -3 % on conv1 whose windows barely overlap).  The last,
    // partial round is dealt wave-major (wave 0 of every workgroup first, then wave 1, ...): dealt like a full round it
    // would land on all waves of the first few CUs, whose SIMDs then run two items more than everybody else's (conv2
    // dgrad at minibatch 1024: 6400 items on 2048 waves -> 8 item-times on 32 CUs against 6 elsewhere; now 7 everywhere).
    const int nitems = ntiles * MG;
    const int item_stride = gmain * G::NWAVES;
    // (not BY_WG) workgroups are dispatched round-robin over the 8 XCDs: renumber them so that consecutive items run
    // on the same XCD and share its L2 (conv1 forward: 72.5 -> 71.1 us)
    const int bx = (gmain % 8 == 0) ? (blockIdx.x % 8) * (gmain / 8) + blockIdx.x / 8 : blockIdx.x;
    const int slot = G::BY_WG ? blockIdx.x * G::NWAVES + wave : bx + gmain * wave;
    const int full = nitems / item_stride, rem = nitems - full * item_stride;
    const int tail_slot = wave * gmain + blockIdx.x;
    int my_items = full + (tail_slot < rem ? 1 : 0);
    // K split (SwGeom::KSPLIT): this workgroup's tail items sit on its waves 0 .. tb-1.  With tb = 4 m + 1 the last of them
    // would be a SECOND tail item for its SIMD while waves tb .. 4 m + 3 have none: those four waves (one per SIMD) take a
    // quarter of its reduction each; the helpers leave their partial tiles in LDS and the item's own wave adds them in
    // the order 1, 2, 3 and runs the epilogue.  (conv2 forward at minibatch 1024: 5184 items on 1024 SIMDs, five
    // per SIMD and a sixth on 64 of them -- 65.2 us stand-alone against 56.7 at 1011 images.)
    constexpr int KS = G::KSPLIT;
    static_assert(!KS || (!RC && G::ksplit_shift_ok()), "K split: constant shift between the parts; not with row classes");
    bool ks_wg = false;                                     // workgroup-uniform
    int ks_part = -1, ks_item = 0;                          // this wave's part of the split item, or -1
    if constexpr (KS != 0) {
        const int tb = rem > (int)blockIdx.x ? (rem - (int)blockIdx.x + gmain - 1) / gmain : 0;
        ks_wg = (tb & 3) == 1;
        if (ks_wg && wave >= tb - 1 && wave < tb + 3) {
            ks_part = wave - (tb - 1);
            ks_item = full * item_stride + (tb - 1) * gmain + blockIdx.x;
            my_items = full + 1;
        }
    }
    if (my_items == 0) {
        if (ks_wg) __syncthreads();                         // (the workgroup meets once behind the split item)
        return;
    }
    auto item_of = [&](int k) {
        return k < full ? k * item_stride + slot : (ks_part >= 0 ? ks_item : full * item_stride + tail_slot);
    };
    // the split item is the wave's last: a quarter of the stages, from the part's chunk pairs on
    auto ks_adjust = [&](int k, const XT*& bs, int& nst_, int& woff_) {
        if constexpr (KS != 0) {
            if (ks_part >= 0 && k == my_items - 1) {
                bs += G::off0(ks_part * (G::J / KS));
                nst_ = NS / KS;
                woff_ = ks_part * (G::J / KS) * TM * 64 * G::EP;
            }
        }
    };
    int kround = 0;
    int item = item_of(0);

    VT buf[2][NL];
    int r; bool valid;
    int mg = item % MG, nst, woff;
    const XT* base = lane_base(item / MG, r, valid, nst, woff);
    ks_adjust(0, base, nst, woff);
    const float* la = lds_lane + mg * (TMW * 64 * G::EP) + woff;
    stage_load(buf[0], base, 0);
    load_a(0, la);
    load_a(1, la);

    f32x16 acc[TMW];
    typename Epi::template Pre<TMW> pre;
    while (true) {
#pragma unroll
        for (int a = 0; a < TMW; ++a)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][g] = 0.f;

        const bool has_next = kround + 1 < my_items;
        const int next_item = has_next ? item_of(kround + 1) : item;
        int r_next = 0; bool valid_next = false;
        const XT* base_next = base;
        const int mg_next = has_next ? next_item % MG : mg;
        int nst_next = nst, woff_next = woff;
        if (has_next) {
            base_next = lane_base(next_item / MG, r_next, valid_next, nst_next, woff_next);
            ks_adjust(kround + 1, base_next, nst_next, woff_next);
        }
        const float* la_next = lds_lane + mg_next * (TMW * 64 * G::EP) + woff_next;

        // one item: NST stages of NL chunk pairs (row classes: NS / 2 for the edge rows)
        auto run = [&](auto nst_c) {
            constexpr int NST = decltype(nst_c)::value, JN = NST * NL;
#pragma unroll
            for (int s = 0; s < NST; ++s) {
                if (s == NST - 1) {                // the epilogue's own reads fly over the last stage of MFMAs
                    epi.template prefetch<TMW>(r, valid, h, mg, pre);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the activations of stage s+1 (or of the next item's stage 0: it flies over the epilogue) are requested
                // a whole stage of MFMAs before they are needed
                // (unconditional: behind a branch the compiler can no longer count the loads in flight and drains them;
                //  the last item simply re-requests its own first stage)
                if (s + 1 < NST) stage_load(buf[(s + 1) & 1], base, s + 1);
                else stage_load(buf[0], base_next, 0);
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const int q = s * NL + i;
                    if (q + 2 < JN) load_a(q + 2, la); else load_a(q + 2 - JN, la_next);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < G::E; ++t)
#pragma unroll
                        for (int a = 0; a < TMW; ++a)
                            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[q & 3][a][t], sw_tap(buf[s & 1][i], t), acc[a], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if constexpr (RC) {
            if (nst == NS) run(std::integral_constant<int, NS>{});
            else run(std::integral_constant<int, NS / 2>{});
        } else if constexpr (KS != 0) {
            if (nst == NS) run(std::integral_constant<int, NS>{});
            else run(std::integral_constant<int, NS / (KS ? KS : 1)>{});
        } else {
            run(std::integral_constant<int, NS>{});
        }
        if (!has_next && ks_part >= 0) break;               // a part of the split item: finished below
        epi.template store<TMW>(r, valid, h, mg, acc, pre);
        if (!has_next) break;
        item = next_item; base = base_next; r = r_next; valid = valid_next; mg = mg_next; nst = nst_next; woff = woff_next;
        la = la_next; ++kround;
    }
    if constexpr (KS != 0) {
        if (ks_wg) {
            float* sc = sw_smem + G::image_floats();         // [part - 1][a][g][lane]
            if (ks_part > 0) {
#pragma unroll
                for (int a = 0; a < TMW; ++a)
#pragma unroll
                    for (int g = 0; g < 16; ++g) sc[(((ks_part - 1) * TMW + a) * 16 + g) * 64 + lane] = acc[a][g];
            }
            __syncthreads();
            if (ks_part == 0) {
#pragma unroll
                for (int q = 0; q < KS - 1; ++q)
#pragma unroll
                    for (int a = 0; a < TMW; ++a)
#pragma unroll
                        for (int g = 0; g < 16; ++g) acc[a][g] += sc[((q * TMW + a) * 16 + g) * 64 + lane];
                epi.template store<TMW>(r, valid, h, mg, acc, pre);
            }
        }
    }
}

int sw_num_cus();

template <class G, class Epi, class XT = float>
int launch_conv_sw(const XT* x, const int64_t* idx, const float* wimg, const Epi& epi, int R, hipStream_t stream,
                   const acktr_slab_fin* carry = nullptr) {
    static_assert(G::NT % 256 == 0, "carry workgroups run virtual 256-thread blocks");
    const int ntiles = sw_num_tiles<G, Epi>(R);
    constexpr int smem_bytes = (int)G::lds_bytes();
    auto kern = conv_sw_kernel<G, Epi, XT>;
    if (smem_bytes > 48 * 1024) {
        static bool attr_set = false;      // per instantiation
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
    // one workgroup per CU, and never fewer workgroups than items allow: the partial round is dealt wave-major, so with
    // fewer than 8 items per CU the items spread over all CUs (one or two waves each) instead of filling a few CUs with
    // eight waves -- at 80 rows (the A2C update) conv2 / conv3 forward ran 2 items per SIMD on 16-51 CUs: 22 us each
    long grid_l = (long)ntiles * G::MG;
    const int cus = sw_num_cus();
    const int grid = (int)(grid_l > cus ? cus : grid_l);
    SwCarry cy;
    cy.fin.dst = nullptr; cy.fin.slabs = nullptr; cy.fin.nslabs = 0; cy.fin.n4 = 0; cy.fin.stride = 0; cy.fin.scale = 1.f;
    cy.nvb = 0;
    int extra = 0;
    if (carry && carry->dst) {
        cy.fin = *carry;
        cy.nvb = (int)((carry->n4 + 15) / 16);
        extra = cdiv(cy.nvb, G::NT / 256);
        if (extra > 128) extra = 128;
    }
    hipLaunchKernelGGL(kern, dim3(grid + extra), dim3(G::NT), smem_bytes, stream, x, idx, wimg, epi, R, ntiles, grid, cy);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

template <class G>
int launch_sw_pack(const float* w, float* img, int M, int mode, int CX, int SD, hipStream_t stream) {
    constexpr long n = G::image_floats();
    hipLaunchKernelGGL((sw_pack_weights_kernel<G>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, img, M,
                       mode, CX, SD);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

}  // namespace acktr
