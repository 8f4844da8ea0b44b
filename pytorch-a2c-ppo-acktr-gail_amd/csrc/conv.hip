// NCHW convolution forward / dgrad / wgrad / bias-grad as implicit GEMMs on fp32 MFMA.
// Replaces the nn.Conv2d call sites of CNNBase (reference a2c_ppo_acktr/model.py:176-179,190) and
// their autograd backward.  Supported geometries are the policy's three layers:
//   8x8 stride 4 on 84x84, 4x4 stride 2 on 20x20, 3x3 stride 1 on 9x9 (any channel counts).
#include "conv_ops.hpp"
#include "conv_sw.hpp"
#include "conv3_dgrad_oa.hpp"
#include "conv_wgrad_sa.hpp"
#include "../../include/acktr_hip.h"

namespace acktr {
int pick_slices(long tiles, long K, long max_by_ws);

enum { GEOM_NONE = 0, GEOM_8S4 = 1, GEOM_4S2 = 2, GEOM_3S1 = 3 };
static int geom_of(int H, int W, int KH, int KW, int S) {
    if (H == 84 && W == 84 && KH == 8 && KW == 8 && S == 4) return GEOM_8S4;
    if (H == 20 && W == 20 && KH == 4 && KW == 4 && S == 2) return GEOM_4S2;
    if (H == 9 && W == 9 && KH == 3 && KW == 3 && S == 1) return GEOM_3S1;
    return GEOM_NONE;
}

template <int KH, int KW, int S, int H, int W, int BM, int WM, int WN>
static int conv_fwd_t(const float* x, const int64_t* idx, const float* w, const float* bias, float* y, int B, int C,
                      int OC, float scale, int relu, hipStream_t s) {
    using Patch = Im2col<KH, KW, S, H, W, true>;
    const int Kc = C * KH * KW, R = B * Patch::SP;
    Patch pop{x, idx, C, (long)C * H * W, R, Kc};
    ConvFwdEpi<Patch::SP> epi{y, bias, OC, R, scale, relu};
    if (vec_ok(w, Kc)) {
        DenseKCT<true> wop{w, OC, Kc, (long)Kc};
        return launch_gemm<BM, 128, WM, WN>(wop, pop, epi, OC, R, Kc, 1, s);
    }
    DenseKCT<false> wop{w, OC, Kc, (long)Kc};
    return launch_gemm<BM, 128, WM, WN>(wop, pop, epi, OC, R, Kc, 1, s);
}

// stride-1 layers: positions are the contiguous axis of the im2col operand (one dwordx4 per 4 positions)
template <int KH, int KW, int H, int W, int OWP, int BM, int WM, int WN>
static int conv_fwd_pos_t(const float* x, const float* w, const float* bias, float* y, int B, int C, int OC,
                          float scale, int relu, hipStream_t s) {
    using Patch = Im2colPos<KH, KW, H, W, OWP, false>;
    const int Kc = C * KH * KW, RP = B * Patch::PP;
    Patch pop{x, x + (long)B * C * H * W, C, RP, Kc};
    ConvFwdEpiPos<Patch::OH, Patch::OW, OWP> epi{y, bias, OC, RP, scale, relu};
    if (vec_ok(w, Kc)) {
        DenseKCT<true> wop{w, OC, Kc, (long)Kc};
        return launch_gemm<BM, 128, WM, WN>(wop, pop, epi, OC, RP, Kc, 1, s);
    }
    DenseKCT<false> wop{w, OC, Kc, (long)Kc};
    return launch_gemm<BM, 128, WM, WN>(wop, pop, epi, OC, RP, Kc, 1, s);
}

// dW[oc, kk] = scale * sum_r G(oc, r) * P(kk, r) with split-K over the position axis (R positions, padded or not).
// db[oc] = sum_r G(oc, r) rides along as the GEMM's M-operand row sum (gemm_core.hpp): dY is read once.
// Slab layout per K slice: [OC*Kc weights | OC (padded to 4) bias]; when db sits right behind dw (it does in the
// policy's flat gradient buffer) one fixed-order reduction finishes both.
static long wgrad_slab_floats(int OC, long Kc) { return (long)OC * Kc + (OC + 3) / 4 * 4; }

template <int BM, int WM, int WN, class GOp, class POp>
static int conv_wgrad_core(const GOp& gop, const POp& pop, float* dw, float* db, int OC, int Kc, int R, float scale,
                           float* ws, long ws_floats, hipStream_t s) {
    const long wk = (long)OC * Kc, slab = wgrad_slab_floats(OC, Kc);
    const long tiles = (long)cdiv(OC, BM) * cdiv(Kc, 128);
    long max_by_ws = ws_floats / slab - SLAB_GROUP_EXTRA;
    int ns = pick_slices(tiles, R, max_by_ws < 1 ? 1 : max_by_ws);
    if (ns > 1 && max_by_ws < ns) ns = 1;
    if (ns == 1) {
        RowMajorEpi e = row_major_epi(dw, OC, Kc, Kc);
        e.scale = scale;
        e.rowsum_dst = db;
        return launch_gemm<BM, 128, WM, WN>(gop, pop, e, OC, Kc, R, 1, s);
    }
    RowMajorEpi e = row_major_epi(ws, OC, Kc, Kc);
    e.slab_stride = slab;
    e.scale = scale;                      // per-slab scaling keeps one reduction for weights and (unscaled) bias
    e.rowsum_dst = db ? ws + wk : nullptr;
    e.rowsum_stride = slab;
    int rc = launch_gemm<BM, 128, WM, WN>(gop, pop, e, OC, Kc, R, ns, s);
    if (rc) return rc;
    if (db == dw + wk && OC % 4 == 0) return launch_slab_reduce(dw, ws, ns, wk + OC, slab, 1.f, 0, s);
    rc = launch_slab_reduce(dw, ws, ns, wk, slab, 1.f, 0, s);
    if (rc || !db) return rc;
    return launch_slab_reduce(db, ws + wk, ns, OC, slab, 1.f, 0, s);
}

template <int KH, int KW, int S, int H, int W, int BM, int WM, int WN>
static int conv_wgrad_t(const float* x, const int64_t* idx, const float* dy, float* dw, float* db, int B, int C,
                        int OC, float scale, float* ws, long ws_floats, hipStream_t s) {
    using Patch = Im2col<KH, KW, S, H, W, false>;
    constexpr int SP = Patch::SP;
    const int Kc = C * KH * KW, R = B * Patch::SPP;      // reduction over positions, each image padded to SPP
    ConvOutKC<SP> gop{dy, OC, R};
    Patch pop{x, idx, C, (long)C * H * W, R, Kc};
    return conv_wgrad_core<BM, WM, WN>(gop, pop, dw, db, OC, Kc, R, scale, ws, ws_floats, s);
}

// stride-1 layers without a gather: both operands stream 4 padded positions per 16-byte load
template <int KH, int KW, int H, int W, int OWP, int BM, int WM, int WN>
static int conv_wgrad_pos_t(const float* x, const float* dy, float* dw, float* db, int B, int C, int OC, float scale,
                            float* ws, long ws_floats, hipStream_t s) {
    using Patch = Im2colPos<KH, KW, H, W, OWP, true>;
    const int Kc = C * KH * KW, RP = B * Patch::PP;
    ConvOutPos<Patch::OH, Patch::OW, OWP> gop{dy, dy + (long)B * OC * Patch::OH * Patch::OW, OC, RP};
    Patch pop{x, x + (long)B * C * H * W, C, RP, Kc};
    return conv_wgrad_core<BM, WM, WN>(gop, pop, dw, db, OC, Kc, RP, scale, ws, ws_floats, s);
}

// 3x3 stride-1 layer with OC == 32, C == 64 and no gather: stationary accumulators (conv_wgrad_sa.hpp)
static bool sa_wgrad_ok(int geom, int C, int OC, const int64_t* idx) { return geom == GEOM_3S1 && C == 64 && OC == SA_OC && !idx; }
static int sa_wgrad_slabs(int B) {
    const int cus = sw_num_cus();
    const int per = cdiv(B, cus);                      // images per workgroup: one workgroup per CU when B allows
    return cdiv(B, per);
}
static int conv3_wgrad_sa(const float* x, const float* dy, float* dw, float* db, int B, float scale, float* ws,
                          long ws_floats, hipStream_t s, acktr_slab_fin* defer = nullptr) {
    constexpr int C = 64, KC = C * SA_KK;
    const long wk = (long)SA_OC * KC, slab = wk + SA_OC;
    const int per = cdiv(B, sw_num_cus()), ns = cdiv(B, per);
    if (ws_floats < (long)(ns + SLAB_GROUP_EXTRA) * slab) return -2;     // caller falls back to the tile GEMM
    hipLaunchKernelGGL((conv3_wgrad_sa_kernel<C>), dim3(ns), dim3(64 * SA3_WAVES), 0, s, x, dy, ws, slab, B, per, scale);
    ACKTR_LAUNCH_CHECK();
    if (defer && db == dw + wk && ns >= 16 && (((uintptr_t)dw | (uintptr_t)ws) & 15) == 0) {
        defer->dst = dw; defer->slabs = ws; defer->nslabs = ns; defer->n4 = (wk + SA_OC) / 4; defer->stride = slab;
        defer->scale = 1.f;
        return 0;
    }
    if (db == dw + wk) return launch_slab_reduce(dw, ws, ns, wk + SA_OC, slab, 1.f, 0, s);
    int rc = launch_slab_reduce(dw, ws, ns, wk, slab, 1.f, 0, s);
    if (rc || !db) return rc;
    return launch_slab_reduce(db, ws + wk, ns, SA_OC, slab, 1.f, 0, s);
}

// 8x8 stride-4 layer on 84x84x4 frames with OC == 32: stationary accumulators, one workgroup per CU (conv_wgrad_sa.hpp)
static bool sa1_wgrad_ok(int geom, int C, int OC, int B) {
    static const bool off = getenv("ACKTR_CONV1_WGRAD_TILE") != nullptr;     // tuning / comparison switch
    return !off && geom == GEOM_8S4 && C == S1_C && OC == S1_OC && B >= 256;     // (B = 80: the tile GEMM's ~400 slices win)
}
static int sa1_wgrad_slabs(int B) {
    const int cus = sw_num_cus();
    const int by_work = B * S1_GROUPS / 20;            // at least five row-pair groups per wave
    return by_work < cus ? (by_work < 1 ? 1 : by_work) : cus;
}
template <class XT>
static int conv1_wgrad_sa(const XT* x, const int64_t* idx, const float* dy, float* dw, float* db, int B, float scale,
                          float* ws, long ws_floats, hipStream_t s, acktr_slab_fin* defer = nullptr) {
    const long wk = (long)S1_OC * S1_KC, slab = wk + S1_OC;
    const int ns = sa1_wgrad_slabs(B);
    if (ws_floats < (long)(ns + SLAB_GROUP_EXTRA) * slab) return -2;     // caller falls back to the tile GEMM
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_wgrad_sa_kernel<XT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, S1_LDS_BYTES);
        if (e != hipSuccess) { set_error("conv1_wgrad_sa: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    // ACKTR_C1W=0 (tuning / comparison switch): the register-direct kernel.  Default: the LDS-ring kernel, 67.0 against 78.1 us
    // inside the PPO step on float frames.  Byte frames take it too although the register-direct form is a little faster
    // there (61.8 against 64.4 us: a byte frame is a quarter of the lines): one summation order for both storage types keeps
    // the gradients of a byte-frame run bit-identical to the float-frame run's (tests/test_u8_gpu.py).
    static const bool ring = [] { const char* v = getenv("ACKTR_C1W"); return !(v && v[0] == '0'); }();
    if (ring && (((uintptr_t)x) & 15) == 0) {
        static bool attr9_set = false;
        if (!attr9_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_wgrad_ring_kernel<XT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, S9_LDS_BYTES);
            if (e != hipSuccess) { set_error("conv1_wgrad_ring: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
            attr9_set = true;
        }
        hipLaunchKernelGGL(conv1_wgrad_ring_kernel<XT>, dim3(ns), dim3(64 * S9_WAVES), S9_LDS_BYTES, s, x, idx, dy, ws, slab, B, scale);
    } else {
        hipLaunchKernelGGL(conv1_wgrad_sa_kernel<XT>, dim3(ns), dim3(64 * S1_WAVES), S1_LDS_BYTES, s, x, idx, dy, ws, slab, B, scale);
    }
    ACKTR_LAUNCH_CHECK();
    if (defer && db == dw + wk && ns >= 16 && (((uintptr_t)dw | (uintptr_t)ws) & 15) == 0) {   // (what slab_reduce_wave_kernel serves)
        defer->dst = dw; defer->slabs = ws; defer->nslabs = ns; defer->n4 = (wk + S1_OC) / 4; defer->stride = slab;
        defer->scale = 1.f;
        return 0;
    }
    if (db == dw + wk) return launch_slab_reduce(dw, ws, ns, wk + S1_OC, slab, 1.f, 0, s);
    int rc = launch_slab_reduce(dw, ws, ns, wk, slab, 1.f, 0, s);
    if (rc || !db) return rc;
    return launch_slab_reduce(db, ws + wk, ns, S1_OC, slab, 1.f, 0, s);
}

// 4x4 stride-2 layer on 20x20x32 activations with OC == 64: stationary accumulators (conv_wgrad_sa.hpp)
static bool sa2_wgrad_ok(int geom, int C, int OC, int B, const int64_t* idx) {
    static const bool off = getenv("ACKTR_CONV2_WGRAD_TILE") != nullptr;     // tuning / comparison switch
    return !off && geom == GEOM_4S2 && C == S2_C && OC == S2_OC && B >= 256 && !idx && (long)B * S2_IMG < (1L << 31);
}
static int sa2_wgrad_slabs(int B) {
    const int cus = sw_num_cus();
    const int by_work = B / 2;                         // at least two images per workgroup
    return by_work < cus ? (by_work < 1 ? 1 : by_work) : cus;
}
static int conv2_wgrad_sa(const float* x, const float* dy, float* dw, float* db, int B, float scale, float* ws, long ws_floats,
                          hipStream_t s, acktr_slab_fin* defer = nullptr) {
    const long wk = (long)S2_OC * S2_KC, slab = wk + S2_OC;
    const int ns = sa2_wgrad_slabs(B);
    if (ws_floats < (long)(ns + SLAB_GROUP_EXTRA) * slab) return -2;     // caller falls back to the tile GEMM
    hipLaunchKernelGGL(conv2_wgrad_sa_kernel, dim3(ns), dim3(64 * S2_WAVES), 0, s, x, dy, ws, slab, B, scale);
    ACKTR_LAUNCH_CHECK();
    if (defer && db == dw + wk && ns >= 16 && (((uintptr_t)dw | (uintptr_t)ws) & 15) == 0) {
        defer->dst = dw; defer->slabs = ws; defer->nslabs = ns; defer->n4 = (wk + S2_OC) / 4; defer->stride = slab;
        defer->scale = 1.f;
        return 0;
    }
    if (db == dw + wk) return launch_slab_reduce(dw, ws, ns, wk + S2_OC, slab, 1.f, 0, s);
    int rc = launch_slab_reduce(dw, ws, ns, wk, slab, 1.f, 0, s);
    if (rc || !db) return rc;
    return launch_slab_reduce(db, ws + wk, ns, S2_OC, slab, 1.f, 0, s);
}

// dgrad weights re-packed once per call into dense per-phase matrices Wp[phase][c][k], k = (oc, a, a'), rows padded
// to a multiple of 4: the GEMM's M operand then streams 16-byte rows instead of gathering 4-byte taps that sit
// KH*KW*C floats apart (one cache line per lane).
template <int KH, int KW, int S>
__global__ __launch_bounds__(256) void dgrad_pack_w_kernel(const float* w, float* wp, int C, int OC, int Kredp) {
    constexpr int TB = (KW + S - 1) / S, TA = (KH + S - 1) / S, TAPS = TA * TB;
    const long n = (long)S * S * C * Kredp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int k = (int)(i % Kredp);
        const int c = (int)((i / Kredp) % C);
        const int ph = (int)(i / ((long)Kredp * C));
        const int oc = k / TAPS, t = k % TAPS;
        const int ky = ph / S + S * (t / TB), kx = ph % S + S * (t % TB);
        float v = 0.f;
        if (oc < OC && ky < KH && kx < KW) v = w[((long)oc * C + c) * (KH * KW) + ky * KW + kx];
        wp[i] = v;
    }
}

struct DgradWPacked {   // M-operand: rows = input channel c of phase blockIdx.z, k contiguous
    static constexpr bool INNER_IS_K = true;
    const float* wp; int C; int Kredp;
    struct Outer { const float* rp; int ok; };
    __device__ __forceinline__ Outer prep(int c) const {
        Outer o; o.ok = c < C; o.rp = wp + ((long)blockIdx.z * C + (o.ok ? c : 0)) * Kredp; return o;
    }
    __device__ static __forceinline__ f32x4 post(const f32x4& v, int meta) { return sel4(meta & 1, v); }
    struct Inner { int k0; };
    __device__ __forceinline__ Inner prep_inner(int k0) const { return Inner{k0}; }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, const Inner& in, int& meta) const {
        const int k0 = in.k0;
        const bool ok = o.ok && k0 < Kredp;
        meta = ok;
        return *reinterpret_cast<const f32x4*>(o.rp + (ok ? k0 : 0));
    }
};

template <int KH, int KW, int S, int OH, int OW, int BM, int WM, int WN>
static int conv_dgrad_t(const float* dy, const float* w, const float* xact, float* dx, int B, int C, int OC,
                        float* wp, hipStream_t s) {
    using G = DgradGeom<KH, KW, S, OH, OW>;
    const int Kred = OC * G::TAPS, Kredp = (Kred + 3) / 4 * 4;
    const long npack = (long)S * S * C * Kredp;
    hipLaunchKernelGGL((dgrad_pack_w_kernel<KH, KW, S>), dim3((unsigned)((npack + 255) / 256)), dim3(256), 0, s, w, wp,
                       C, OC, Kredp);
    ACKTR_LAUNCH_CHECK();
    DgradWPacked wop{wp, C, Kredp};
    // all S*S stride phases in ONE launch: blockIdx.z selects the phase (the K range is never split here)
    // (a position-contiguous 16-byte dY loader was measured slower here: the padded phase rows cost more MFMA
    //  work than the wider loads save)
    const int NP = B * G::JP;
    DgradDy<KH, KW, S, OH, OW> gop{dy, OC, NP, Kred};
    DgradEpi<KH, KW, S, OH, OW> epi{dx, xact, C, NP};
    return launch_gemm_z<BM, 128, WM, WN>(wop, gop, epi, C, NP, Kred, S * S, s);
}

int sw_num_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

// stationary-weight geometries (conv_sw.hpp): forward of the three layers, dgrad of layers 3 and 2
// (channel counts are those of the reference's CNNBase, model.py:176-179; other counts take the tile-GEMM path)
// (last parameter: threads per workgroup.  3-4 waves per SIMD were measured no faster than 2 except for the conv3
//  dgrad: the kernels are paced by the MFMA pipe and the L1, and the smaller prefetch stages that fit the tighter
//  register budget cost more than the extra waves hide)
// (stage depths NL: compile-time switches for same-box A/B builds, tools/ab_update.py)
#ifndef ACKTR_NL_F1
#define ACKTR_NL_F1 16     // (8: 68.8 us, 16: 66.9 us at minibatch 1024, same-box A/B; the other layers lose with deeper stages)
#endif
#ifndef ACKTR_NL_F2
#define ACKTR_NL_F2 8
#endif
#ifndef ACKTR_NL_F3
#define ACKTR_NL_F3 12
#endif
#ifndef ACKTR_NL_D2
#define ACKTR_NL_D2 8
#endif
using SwF1 = SwGeom<8, 8, 4, 84, 84, 4, 1, 1, ACKTR_NL_F1, HM_KX, 512, false>;
#ifndef ACKTR_KS_F2
#define ACKTR_KS_F2 4      // (0: no K split of the odd tail item; comparison builds)
#endif
using SwF2 = SwGeom<4, 4, 2, 20, 20, 32, 2, 1, ACKTR_NL_F2, HM_KY, 512, true, ACKTR_KS_F2>;
using SwF3 = SwGeom<3, 3, 1, 9, 9, 64, 1, 1, ACKTR_NL_F3, HM_CH, 512>;
using SwD3 = SwGeom<3, 3, 1, 11, 11, 32, 2, 1, 8, HM_CH, 768>;     // padded 7x7 dY, flipped 3x3 kernel (stages of 8: room for the prefetched gate in 168 registers)
using SwD2 = SwGeom<2, 2, 1, 11, 11, 64, 4, 2, ACKTR_NL_D2, HM_CH, 512>;     // padded 9x9 dY, four 2x2 phase kernels stacked along M

static long sw_fwd_image_bytes(int geom, int C, int OC) {
    if (geom == GEOM_8S4 && SwF1::ok(C, OC)) return SwF1::image_floats() * 4;
    if (geom == GEOM_4S2 && SwF2::ok(C, OC)) return SwF2::image_floats() * 4;
    if (geom == GEOM_3S1 && SwF3::ok(C, OC)) return SwF3::image_floats() * 4;
    return 0;
}
// dgrad through the stationary kernel: C = channels of dX, OC = channels of dY (the reduction)
static bool sw_dgrad_ok(int geom, int C, int OC) {
    if (geom == GEOM_4S2) return C == 32 && SwD2::ok(OC, 128);
    if (geom == GEOM_3S1) return SwD3::ok(OC, C);
    return false;
}
static long sw_dgrad_image_floats(int geom, int OC) {
    return geom == GEOM_4S2 ? SwD2::image_floats() : SwD3::image_floats();
}
static long sw_dgrad_pad_floats(int geom, int B, int OC) { return (long)B * OC * 121 + 4; }
// conv3's input gradient in overlap-add form (conv3_dgrad_oa.hpp) instead of the gather over the zero-padded dY:
// NatureCNN's own channel counts only; ACKTR_D3_OA=0 keeps the gather kernel (same-box comparisons)
static bool d3_oa(int geom, int C, int OC) {
    static const bool on = []() { const char* ev = getenv("ACKTR_D3_OA"); return !ev || atoi(ev) != 0; }();
    return on && geom == GEOM_3S1 && C == OA_CI && OC == OA_CO;
}

// Several weight images in ONE launch (blockIdx.y = job): inside a captured graph every dependent launch costs
// ~4.5 us, more than any of these packs takes.
enum { PK_F1, PK_F2, PK_F3, PK_D3, PK_D2, PK_D3OA };
struct SwPackJob { const float* w; float* img; int kind, M, mode, CX, SD; };
constexpr int SW_PACK_MAX_JOBS = 8;
struct SwPackJobs { SwPackJob j[SW_PACK_MAX_JOBS]; };
__global__ __launch_bounds__(256) void sw_pack_multi_kernel(SwPackJobs jobs) {
    const SwPackJob jb = jobs.j[blockIdx.y];
    switch (jb.kind) {
        case PK_F1: sw_pack_body<SwF1>(jb.w, jb.img, jb.M, jb.mode, jb.CX, jb.SD); break;
        case PK_F2: sw_pack_body<SwF2>(jb.w, jb.img, jb.M, jb.mode, jb.CX, jb.SD); break;
        case PK_F3: sw_pack_body<SwF3>(jb.w, jb.img, jb.M, jb.mode, jb.CX, jb.SD); break;
        case PK_D3: sw_pack_body<SwD3>(jb.w, jb.img, jb.M, jb.mode, jb.CX, jb.SD); break;
        case PK_D3OA: oa_pack_body(jb.w, jb.img); break;
        default: sw_pack_body<SwD2>(jb.w, jb.img, jb.M, jb.mode, jb.CX, jb.SD); break;
    }
}
}  // namespace acktr

using namespace acktr;

#define CONV_COMMON_CHECKS(name)                                                                              \
    ACKTR_ARG(B > 0 && C > 0 && OC > 0, name ": bad sizes B=%d C=%d OC=%d", B, C, OC);                         \
    const int geom = geom_of(H, W, KH, KW, S);                                                                \
    ACKTR_ARG(geom != GEOM_NONE,                                                                              \
              name ": unsupported geometry %dx%d k%dx%d s%d (supported: 84x84 k8 s4, 20x20 k4 s2, 9x9 k3 s1)", \
              H, W, KH, KW, S);                                                                               \
    ACKTR_ARG((long)B * C * H * W < (1L << 31) || gather_ok, name ": tensor too large for 32-bit positions")

extern "C" long acktr_conv2d_fwd_workspace_bytes(int C, int H, int W, int OC, int KH, int KW, int S) {
    const int geom = geom_of(H, W, KH, KW, S);
    if (geom == GEOM_NONE || C <= 0 || OC <= 0) return 0;
    return sw_fwd_image_bytes(geom, C, OC);
}

static int conv2d_fwd_impl(const float* x, const int64_t* gather_idx, const float* w, const float* bias,
                           float* y, int B, int C, int H, int W, int OC, int KH, int KW, int S,
                           float in_scale, int relu, void* workspace, long workspace_bytes, int prepacked,
                           unsigned* gate_mask, void* stream) {
    ACKTR_ARG(x && w && y, "conv2d_fwd: null pointer");
    const bool gather_ok = true;
    CONV_COMMON_CHECKS("conv2d_fwd");
    ACKTR_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "conv2d_fwd: x and w must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // stationary-weight path (needs the weight-image workspace); the tile-GEMM path below covers everything else
    const long img_bytes = sw_fwd_image_bytes(geom, C, OC);
    if (img_bytes > 0 && workspace && workspace_bytes >= img_bytes && ((uintptr_t)workspace & 15) == 0 &&
        (long)B * OC * 400 < (1L << 31)) {
        float* img = (float*)workspace;
        int rc = 0;
        if (geom == GEOM_8S4) {
            if (!prepacked) rc = launch_sw_pack<SwF1>(w, img, OC, 0, 0, 0, s);
            if (rc) return rc;
            if (bias) {
                SwFwdEpi<SwF1::SP, true> epi{y, bias, OC, in_scale, relu, gate_mask};
                return launch_conv_sw<SwF1>(x, gather_idx, img, epi, B * SwF1::SP, s);
            }
            SwFwdEpi<SwF1::SP, false> epi{y, nullptr, OC, in_scale, relu, gate_mask};
            return launch_conv_sw<SwF1>(x, gather_idx, img, epi, B * SwF1::SP, s);
        }
        ACKTR_ARG(!gate_mask, "conv2d_fwd_gate: the gate mask is written by the 8x8 stride-4 stationary-weight path only");
        if (geom == GEOM_4S2) {
            if (!prepacked) rc = launch_sw_pack<SwF2>(w, img, OC, 0, 0, 0, s);
            if (rc) return rc;
            if (bias) {
                SwFwdEpi<SwF2::SP, true> epi{y, bias, OC, in_scale, relu};
                return launch_conv_sw<SwF2>(x, gather_idx, img, epi, B * SwF2::SP, s);
            }
            SwFwdEpi<SwF2::SP, false> epi{y, nullptr, OC, in_scale, relu};
            return launch_conv_sw<SwF2>(x, gather_idx, img, epi, B * SwF2::SP, s);
        }
        if (!prepacked) rc = launch_sw_pack<SwF3>(w, img, OC, 0, 0, 0, s);
        if (rc) return rc;
        if (bias) {
            SwFwdEpi<SwF3::SP, true> epi{y, bias, OC, in_scale, relu};
            return launch_conv_sw<SwF3>(x, gather_idx, img, epi, B * SwF3::SP, s);
        }
        SwFwdEpi<SwF3::SP, false> epi{y, nullptr, OC, in_scale, relu};
        return launch_conv_sw<SwF3>(x, gather_idx, img, epi, B * SwF3::SP, s);
    }
    ACKTR_ARG(!gate_mask, "conv2d_fwd_gate: the gate mask is written by the 8x8 stride-4 stationary-weight path only "
                          "(needs the weight-image workspace)");
    if (geom == GEOM_8S4) {
        if (OC <= 32) return conv_fwd_t<8, 8, 4, 84, 84, 32, 1, 4>(x, gather_idx, w, bias, y, B, C, OC, in_scale, relu, s);
        return conv_fwd_t<8, 8, 4, 84, 84, 64, 2, 2>(x, gather_idx, w, bias, y, B, C, OC, in_scale, relu, s);
    }
    if (geom == GEOM_4S2) {
        if (OC <= 32) return conv_fwd_t<4, 4, 2, 20, 20, 32, 1, 4>(x, gather_idx, w, bias, y, B, C, OC, in_scale, relu, s);
        return conv_fwd_t<4, 4, 2, 20, 20, 64, 2, 2>(x, gather_idx, w, bias, y, B, C, OC, in_scale, relu, s);
    }
    if (gather_idx) {
        if (OC <= 32) return conv_fwd_t<3, 3, 1, 9, 9, 32, 1, 4>(x, gather_idx, w, bias, y, B, C, OC, in_scale, relu, s);
        return conv_fwd_t<3, 3, 1, 9, 9, 64, 2, 2>(x, gather_idx, w, bias, y, B, C, OC, in_scale, relu, s);
    }
    if (OC <= 32) return conv_fwd_pos_t<3, 3, 9, 9, 8, 32, 1, 4>(x, w, bias, y, B, C, OC, in_scale, relu, s);
    return conv_fwd_pos_t<3, 3, 9, 9, 8, 64, 2, 2>(x, w, bias, y, B, C, OC, in_scale, relu, s);
}

extern "C" int acktr_conv2d_fwd_f32(const float* x, const int64_t* gather_idx, const float* w, const float* bias,
                                    float* y, int B, int C, int H, int W, int OC, int KH, int KW, int S,
                                    float in_scale, int relu, void* workspace, long workspace_bytes, int prepacked,
                                    void* stream) {
    return conv2d_fwd_impl(x, gather_idx, w, bias, y, B, C, H, W, OC, KH, KW, S, in_scale, relu, workspace, workspace_bytes,
                           prepacked, nullptr, stream);
}
extern "C" long acktr_conv2d_gate_mask_bytes(int B, int OC, int OH, int OW) {
    if (B <= 0 || OC <= 0 || OH <= 0 || OW <= 0) return 0;
    return (((long)B * OH * OW + 31) / 32) * OC * (long)sizeof(unsigned);
}
extern "C" int acktr_conv2d_fwd_gate_f32(const float* x, const int64_t* gather_idx, const float* w, const float* bias,
                                         float* y, int B, int C, int H, int W, int OC, int KH, int KW, int S,
                                         float in_scale, int relu, void* workspace, long workspace_bytes, int prepacked,
                                         uint32_t* gate_mask, void* stream) {
    ACKTR_ARG(gate_mask && ((uintptr_t)gate_mask & 15) == 0, "conv2d_fwd_gate: gate_mask must be given, 16-byte aligned");
    return conv2d_fwd_impl(x, gather_idx, w, bias, y, B, C, H, W, OC, KH, KW, S, in_scale, relu, workspace, workspace_bytes,
                           prepacked, gate_mask, stream);
}

extern "C" long acktr_conv2d_wgrad_workspace_bytes(int B, int C, int H, int W, int OC, int KH, int KW, int S) {
    if (geom_of(H, W, KH, KW, S) == GEOM_NONE || B <= 0) return 0;
    const int OH = (H - KH) / S + 1, OW = (W - KW) / S + 1;
    const long Kc = (long)C * KH * KW;
    const int BM = OC <= 32 ? 32 : 64;
    const int OWP = (OW + 3) / 4 * 4;      // the stride-1 path reduces over padded positions
    int ns = pick_slices((long)cdiv(OC, BM) * cdiv(Kc, 128), (long)B * OH * OWP, 1024);
    if (sa_wgrad_ok(geom_of(H, W, KH, KW, S), C, OC, nullptr) && sa_wgrad_slabs(B) > ns) ns = sa_wgrad_slabs(B);
    if (sa1_wgrad_ok(geom_of(H, W, KH, KW, S), C, OC, B) && sa1_wgrad_slabs(B) > ns) ns = sa1_wgrad_slabs(B);
    if (sa2_wgrad_ok(geom_of(H, W, KH, KW, S), C, OC, B, nullptr) && sa2_wgrad_slabs(B) > ns) ns = sa2_wgrad_slabs(B);
    return (long)(ns + SLAB_GROUP_EXTRA) * wgrad_slab_floats(OC, Kc) * (long)sizeof(float);
}

static int conv2d_wgrad_impl(const float* x, const int64_t* gather_idx, const float* dy, float* dw,
                             float* db, int B, int C, int H, int W, int OC, int KH, int KW, int S,
                             float in_scale, void* workspace, long workspace_bytes, void* stream, acktr_slab_fin* defer) {
    ACKTR_ARG(x && dy && dw, "conv2d_wgrad: null pointer");
    const bool gather_ok = true;
    CONV_COMMON_CHECKS("conv2d_wgrad");
    ACKTR_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "conv2d_wgrad: x and dy must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const long wf = workspace ? workspace_bytes / (long)sizeof(float) : 0;
#define WG(KH_, KW_, S_, H_, W_)                                                                                  \
    (OC <= 32 ? conv_wgrad_t<KH_, KW_, S_, H_, W_, 32, 1, 4>(x, gather_idx, dy, dw, db, B, C, OC, in_scale, ws, wf, s) \
              : conv_wgrad_t<KH_, KW_, S_, H_, W_, 64, 2, 2>(x, gather_idx, dy, dw, db, B, C, OC, in_scale, ws, wf, s))
    if (sa1_wgrad_ok(geom, C, OC, B)) {
        const int rc = conv1_wgrad_sa(x, gather_idx, dy, dw, db, B, in_scale, ws, wf, s, defer);
        if (rc != -2) return rc;
    }
    if (sa2_wgrad_ok(geom, C, OC, B, gather_idx)) {
        const int rc = conv2_wgrad_sa(x, dy, dw, db, B, in_scale, ws, wf, s, defer);
        if (rc != -2) return rc;
    }
    if (geom == GEOM_8S4) return WG(8, 8, 4, 84, 84);
    if (geom == GEOM_4S2) return WG(4, 4, 2, 20, 20);
    if (gather_idx) return WG(3, 3, 1, 9, 9);
#undef WG
    if (sa_wgrad_ok(geom, C, OC, gather_idx) && (long)B * C * 81 * 4 < (1L << 31)) {   // 32-bit byte offsets
        const int rc = conv3_wgrad_sa(x, dy, dw, db, B, in_scale, ws, wf, s, defer);
        if (rc != -2) return rc;
    }
    if (OC <= 32) return conv_wgrad_pos_t<3, 3, 9, 9, 8, 32, 1, 4>(x, dy, dw, db, B, C, OC, in_scale, ws, wf, s);
    return conv_wgrad_pos_t<3, 3, 9, 9, 8, 64, 2, 2>(x, dy, dw, db, B, C, OC, in_scale, ws, wf, s);
}

static long dgrad_ws_floats(int B, int C, int H, int W, int OC, int KH, int KW, int S) {
    const int taps = ((KH + S - 1) / S) * ((KW + S - 1) / S);
    const long kredp = ((long)OC * taps + 3) / 4 * 4;
    long n = (long)S * S * C * kredp;                                   // tile-GEMM path: per-phase packed weights
    const int geom = geom_of(H, W, KH, KW, S);
    if (sw_dgrad_ok(geom, C, OC)) {                                    // stationary path: weight image + padded dY
        const long m = sw_dgrad_image_floats(geom, OC) + sw_dgrad_pad_floats(geom, B, OC);
        if (m > n) n = m;
    }
    return n;
}

// ---- frames stored as bytes (rollout storage with obs_dtype=torch.uint8): NatureCNN's first layer only ----
extern "C" long acktr_conv1_wgrad_u8_workspace_bytes(int B) {
    if (B <= 0) return 0;
    return (long)(sa1_wgrad_slabs(B) + SLAB_GROUP_EXTRA) * ((long)S1_OC * S1_KC + S1_OC) * (long)sizeof(float);
}

static int conv1_fwd_u8_impl(const uint8_t* x, const int64_t* gather_idx, const float* w, const float* bias, float* y,
                             int B, float in_scale, int relu, void* workspace, long workspace_bytes, int prepacked,
                             unsigned* gate_mask, void* stream) {
    ACKTR_ARG(x && w && y && workspace, "conv1_fwd_u8: null pointer");
    ACKTR_ARG(B > 0 && (long)B * 32 * 400 < (1L << 31), "conv1_fwd_u8: bad batch size %d", B);
    ACKTR_ARG(((uintptr_t)x & 3) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
              "conv1_fwd_u8: x must be 4-byte, w and workspace 16-byte aligned");
    ACKTR_ARG(workspace_bytes >= SwF1::image_floats() * 4, "conv1_fwd_u8: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* img = (float*)workspace;
    if (!prepacked) { const int rc = launch_sw_pack<SwF1>(w, img, 32, 0, 0, 0, s); if (rc) return rc; }
    if (bias) {
        SwFwdEpi<SwF1::SP, true> epi{y, bias, 32, in_scale, relu, gate_mask};
        return launch_conv_sw<SwF1, SwFwdEpi<SwF1::SP, true>, uint8_t>(x, gather_idx, img, epi, B * SwF1::SP, s);
    }
    SwFwdEpi<SwF1::SP, false> epi{y, nullptr, 32, in_scale, relu, gate_mask};
    return launch_conv_sw<SwF1, SwFwdEpi<SwF1::SP, false>, uint8_t>(x, gather_idx, img, epi, B * SwF1::SP, s);
}
extern "C" int acktr_conv1_fwd_u8(const uint8_t* x, const int64_t* gather_idx, const float* w, const float* bias, float* y,
                                  int B, float in_scale, int relu, void* workspace, long workspace_bytes, int prepacked,
                                  void* stream) {
    return conv1_fwd_u8_impl(x, gather_idx, w, bias, y, B, in_scale, relu, workspace, workspace_bytes, prepacked, nullptr, stream);
}
extern "C" int acktr_conv1_fwd_u8_gate(const uint8_t* x, const int64_t* gather_idx, const float* w, const float* bias, float* y,
                                       int B, float in_scale, int relu, void* workspace, long workspace_bytes, int prepacked,
                                       uint32_t* gate_mask, void* stream) {
    ACKTR_ARG(gate_mask && ((uintptr_t)gate_mask & 15) == 0, "conv1_fwd_u8_gate: gate_mask must be given, 16-byte aligned");
    return conv1_fwd_u8_impl(x, gather_idx, w, bias, y, B, in_scale, relu, workspace, workspace_bytes, prepacked, gate_mask, stream);
}

extern "C" int acktr_conv1_wgrad_u8(const uint8_t* x, const int64_t* gather_idx, const float* dy, float* dw, float* db,
                                    int B, float in_scale, void* workspace, long workspace_bytes, acktr_slab_fin* pending,
                                    void* stream) {
    ACKTR_ARG(x && dy && dw && workspace, "conv1_wgrad_u8: null pointer");
    ACKTR_ARG(B > 0, "conv1_wgrad_u8: bad batch size %d", B);
    ACKTR_ARG(((uintptr_t)x & 3) == 0 && ((uintptr_t)dy & 15) == 0, "conv1_wgrad_u8: x must be 4-byte, dy 16-byte aligned");
    if (pending) pending->dst = nullptr;
    const int rc = conv1_wgrad_sa<uint8_t>(x, gather_idx, dy, dw, db, B, in_scale, (float*)workspace,
                                           workspace_bytes / (long)sizeof(float), (hipStream_t)stream, pending);
    ACKTR_ARG(rc != -2, "conv1_wgrad_u8: workspace too small (acktr_conv2d_wgrad_workspace_bytes(B, 4, 84, 84, 32, 8, 8, 4))");
    return rc;
}

extern "C" int acktr_conv2d_wgrad_f32(const float* x, const int64_t* gather_idx, const float* dy, float* dw,
                                      float* db, int B, int C, int H, int W, int OC, int KH, int KW, int S,
                                      float in_scale, void* workspace, long workspace_bytes, void* stream) {
    return conv2d_wgrad_impl(x, gather_idx, dy, dw, db, B, C, H, W, OC, KH, KW, S, in_scale, workspace, workspace_bytes, stream,
                             nullptr);
}

extern "C" int acktr_conv2d_wgrad_deferred_f32(const float* x, const int64_t* gather_idx, const float* dy, float* dw,
                                               float* db, int B, int C, int H, int W, int OC, int KH, int KW, int S,
                                               float in_scale, void* workspace, long workspace_bytes,
                                               acktr_slab_fin* pending, void* stream) {
    ACKTR_ARG(pending, "conv2d_wgrad_deferred: null pending");
    pending->dst = nullptr;
    return conv2d_wgrad_impl(x, gather_idx, dy, dw, db, B, C, H, W, OC, KH, KW, S, in_scale, workspace, workspace_bytes, stream,
                             pending);
}

extern "C" long acktr_conv2d_dgrad_workspace_bytes(int B, int C, int H, int W, int OC, int KH, int KW, int S) {
    if (B <= 0 || C <= 0 || OC <= 0 || KH <= 0 || KW <= 0 || S <= 0) return 0;
    return dgrad_ws_floats(B, C, H, W, OC, KH, KW, S) * (long)sizeof(float);
}

extern "C" long acktr_conv2d_dgrad_padded_dy_offset_bytes(int C, int H, int W, int OC, int KH, int KW, int S) {
    const int geom = geom_of(H, W, KH, KW, S);
    if (geom == GEOM_NONE || !sw_dgrad_ok(geom, C, OC)) return -1;
    if (d3_oa(geom, C, OC)) return -1;                 // the overlap-add kernel reads dY as it is
    return sw_dgrad_image_floats(geom, OC) * (long)sizeof(float);
}

extern "C" int acktr_conv2d_dgrad_f32(const float* dy, const float* w, const float* x_act, float* dx, int B, int C,
                                      int H, int W, int OC, int KH, int KW, int S, void* workspace,
                                      long workspace_bytes, int prepacked, void* stream) {
    return acktr_conv2d_dgrad_ex_f32(dy, w, x_act, dx, B, C, H, W, OC, KH, KW, S, workspace, workspace_bytes, prepacked,
                                     nullptr, 0, stream);
}

extern "C" int acktr_conv2d_dgrad_ex_f32(const float* dy, const float* w, const float* x_act, float* dx, int B, int C,
                                         int H, int W, int OC, int KH, int KW, int S, void* workspace,
                                         long workspace_bytes, int flags, float* dx_padded, int pad, void* stream) {
    return acktr_conv2d_dgrad_carry_f32(dy, w, x_act, dx, B, C, H, W, OC, KH, KW, S, workspace, workspace_bytes, flags,
                                        dx_padded, pad, nullptr, stream);
}

static int run_carry_alone(const acktr_slab_fin* carry, hipStream_t s) {
    if (!carry || !carry->dst) return 0;
    return launch_slab_reduce(carry->dst, carry->slabs, carry->nslabs, 4 * carry->n4, carry->stride, carry->scale, 0, s);
}

extern "C" int acktr_conv2d_dgrad_carry_f32(const float* dy, const float* w, const float* x_act, float* dx, int B, int C,
                                            int H, int W, int OC, int KH, int KW, int S, void* workspace,
                                            long workspace_bytes, int flags, float* dx_padded, int pad,
                                            const acktr_slab_fin* carry, void* stream) {
    const int prepacked = flags & 1;
    const bool prepadded = (flags & 2) != 0;      // the zero-padded dY already sits in the workspace (a producer wrote it)
    const bool gate_bits = (flags & 4) != 0;      // x_act holds the gate as bits (acktr_conv2d_fwd_gate_f32's gate_mask)
    ACKTR_ARG((dy || prepadded) && w && dx, "conv2d_dgrad: null pointer");
    ACKTR_ARG(!gate_bits || (x_act && ((uintptr_t)x_act & 15) == 0), "conv2d_dgrad: flags bit 2 needs a 16-byte aligned gate mask in x_act");
    ACKTR_ARG(!dx_padded || (pad > 0 && pad <= 4), "conv2d_dgrad: dx_padded needs 1 <= pad <= 4");
    const bool gather_ok = false;
    CONV_COMMON_CHECKS("conv2d_dgrad");
    ACKTR_ARG(workspace && workspace_bytes >= dgrad_ws_floats(B, C, H, W, OC, KH, KW, S) * (long)sizeof(float),
              "conv2d_dgrad: workspace too small (%ld bytes given)", workspace_bytes);
    ACKTR_ARG(((uintptr_t)workspace & 15) == 0, "conv2d_dgrad: workspace must be 16-byte aligned");
    ACKTR_ARG(!gate_bits || (geom == GEOM_4S2 && sw_dgrad_ok(geom, C, OC) && (long)B * OC * 121 < (1L << 31)),
              "conv2d_dgrad: a gate mask (flags bit 2) is read by the 4x4 stride-2 stationary-weight path only");
    hipStream_t s = (hipStream_t)stream;
    float* wp = (float*)workspace;
    if (d3_oa(geom, C, OC) && sw_dgrad_ok(geom, C, OC) && (long)B * C * 121 < (1L << 31)) {
        ACKTR_ARG(dy && !prepadded, "conv2d_dgrad: this shape reads dY unpadded (acktr_conv2d_dgrad_padded_dy_offset_bytes says so)");
        static_assert(OA_IMG_FLOATS <= SwD3::IMAGE_FLOATS, "the overlap-add image lives in the gather kernel's slot");
        if (!prepacked) {
            hipLaunchKernelGGL(sw_pack_multi_kernel, dim3((OA_IMG_FLOATS + 255) / 256, 1), dim3(256), 0, s,
                               SwPackJobs{{{w, wp, PK_D3OA, 0, 0, 0, 0}}});
            ACKTR_LAUNCH_CHECK();
        }
        const int pw = W + 2 * pad;
        if (x_act) return launch_conv3_dgrad_oa<true>(dy, wp, x_act, dx, dx_padded, pw, (H + 2 * pad) * pw, pad * pw + pad, B, s, carry);
        return launch_conv3_dgrad_oa<false>(dy, wp, nullptr, dx, dx_padded, pw, (H + 2 * pad) * pw, pad * pw + pad, B, s, carry);
    }
    if (sw_dgrad_ok(geom, C, OC) && (long)B * OC * 121 < (1L << 31)) {
        float* img = wp;
        float* dyp = wp + sw_dgrad_image_floats(geom, OC);
        const long planes = (long)B * OC;
        const unsigned pgrid = (unsigned)((planes + 7) / 8);
        if (!prepadded) {
            if (geom == GEOM_4S2) hipLaunchKernelGGL((sw_pad_kernel<9, 9, 1>), dim3(pgrid), dim3(256), 0, s, dy, dyp, planes);
            else hipLaunchKernelGGL((sw_pad_kernel<7, 7, 2>), dim3(pgrid), dim3(256), 0, s, dy, dyp, planes);
            ACKTR_LAUNCH_CHECK();
        }
        int rc = 0;
        if (geom == GEOM_4S2) {
            ACKTR_ARG(!dx_padded, "conv2d_dgrad: dx_padded is only produced by the 3x3 stride-1 stationary path");
            if (!prepacked) rc = launch_sw_pack<SwD2>(w, img, 4 * C, 1, C, 2, s);
            if (rc) return rc;
            // row classes (conv_sw.hpp): edge rows of the phase grid skip the kernel row that only reads padding.
            // ACKTR_D2_ROWCLS=0 (tuning / comparison switch): every tile runs the full reduction
            static const bool rowcls = [] { const char* v = getenv("ACKTR_D2_ROWCLS"); return !(v && v[0] == '0'); }();
            if (gate_bits) {
                if (rowcls) {
                    SwPhase2Epi<10, 10, 20, 20, 2, true> epi{dx, x_act};
                    return launch_conv_sw<SwD2>(dyp, nullptr, img, epi, B * 100, s, carry);
                }
                SwPhase2Epi<10, 10, 20, 20, 2> epi{dx, x_act};
                return launch_conv_sw<SwD2>(dyp, nullptr, img, epi, B * 100, s, carry);
            }
            if (x_act) {
                if (rowcls) {
                    SwPhase2Epi<10, 10, 20, 20, true, true> epi{dx, x_act};
                    return launch_conv_sw<SwD2>(dyp, nullptr, img, epi, B * 100, s, carry);
                }
                SwPhase2Epi<10, 10, 20, 20, true> epi{dx, x_act};
                return launch_conv_sw<SwD2>(dyp, nullptr, img, epi, B * 100, s, carry);
            }
            if (rowcls) {
                SwPhase2Epi<10, 10, 20, 20, false, true> epi{dx, nullptr};
                return launch_conv_sw<SwD2>(dyp, nullptr, img, epi, B * 100, s, carry);
            }
            SwPhase2Epi<10, 10, 20, 20, false> epi{dx, nullptr};
            return launch_conv_sw<SwD2>(dyp, nullptr, img, epi, B * 100, s, carry);
        }
        if (!prepacked) rc = launch_sw_pack<SwD3>(w, img, C, 1, C, 1, s);
        if (rc) return rc;
        const int pw = W + 2 * pad;
        if (x_act) {
            SwGateEpi<81, true> epi{dx, x_act, C, dx_padded, W, pw, (H + 2 * pad) * pw, pad * pw + pad};
            return launch_conv_sw<SwD3>(dyp, nullptr, img, epi, B * 81, s, carry);
        }
        SwGateEpi<81, false> epi{dx, nullptr, C, dx_padded, W, pw, (H + 2 * pad) * pw, pad * pw + pad};
        return launch_conv_sw<SwD3>(dyp, nullptr, img, epi, B * 81, s, carry);
    }
    ACKTR_ARG(!prepadded && !dx_padded, "conv2d_dgrad: padded operands need the stationary-weight path (this shape runs the tile GEMM)");
    { const int rcc = run_carry_alone(carry, s); if (rcc) return rcc; }      // (no stationary launch to ride in)
#define DG(KH_, KW_, S_, OH_, OW_)                                                                 \
    (C <= 32 ? conv_dgrad_t<KH_, KW_, S_, OH_, OW_, 32, 1, 4>(dy, w, x_act, dx, B, C, OC, wp, s)   \
             : conv_dgrad_t<KH_, KW_, S_, OH_, OW_, 64, 2, 2>(dy, w, x_act, dx, B, C, OC, wp, s))
    if (geom == GEOM_8S4) return DG(8, 8, 4, 20, 20);
    if (geom == GEOM_4S2) return DG(4, 4, 2, 9, 9);
    return DG(3, 3, 1, 7, 7);
#undef DG
}

/* Weight images of several layers in one launch; see include/acktr_hip.h. */
extern "C" int acktr_conv2d_pack_weights_f32(const acktr_conv_pack_job* jobs, int count, void* stream) {
    ACKTR_ARG(jobs && count > 0 && count <= SW_PACK_MAX_JOBS, "conv2d_pack_weights: 1..%d jobs", SW_PACK_MAX_JOBS);
    SwPackJobs pj;
    int n = 0;
    long biggest = 0;
    for (int i = 0; i < count; ++i) {
        const acktr_conv_pack_job& jb = jobs[i];
        ACKTR_ARG(jb.w && jb.workspace && (((uintptr_t)jb.workspace | (uintptr_t)jb.w) & 15) == 0,
                  "conv2d_pack_weights: job %d: null or unaligned pointer", i);
        const int geom = geom_of(jb.H, jb.W, jb.KH, jb.KW, jb.S);
        SwPackJob o{jb.w, (float*)jb.workspace, 0, 0, 0, 0, 0};
        long img_floats = 0;
        if (!jb.dgrad) {
            const long bytes = geom == GEOM_NONE ? 0 : sw_fwd_image_bytes(geom, jb.C, jb.OC);
            ACKTR_ARG(bytes > 0 && jb.workspace_bytes >= bytes,
                      "conv2d_pack_weights: job %d is not a stationary-weight forward geometry (or workspace too small)", i);
            o.kind = geom == GEOM_8S4 ? PK_F1 : geom == GEOM_4S2 ? PK_F2 : PK_F3;
            o.M = jb.OC;
            img_floats = bytes / 4;
        } else {
            ACKTR_ARG(geom != GEOM_NONE && sw_dgrad_ok(geom, jb.C, jb.OC),
                      "conv2d_pack_weights: job %d is not a stationary-weight dgrad geometry", i);
            img_floats = sw_dgrad_image_floats(geom, jb.OC);
            ACKTR_ARG(jb.workspace_bytes >= img_floats * 4, "conv2d_pack_weights: job %d: workspace too small", i);
            o.mode = 1;
            o.CX = jb.C;
            if (geom == GEOM_4S2) { o.kind = PK_D2; o.M = 4 * jb.C; o.SD = 2; }
            else if (d3_oa(geom, jb.C, jb.OC)) { o.kind = PK_D3OA; img_floats = OA_IMG_FLOATS; }
            else { o.kind = PK_D3; o.M = jb.C; o.SD = 1; }
        }
        if (img_floats > biggest) biggest = img_floats;
        pj.j[n++] = o;
    }
    hipLaunchKernelGGL(sw_pack_multi_kernel, dim3((unsigned)((biggest + 255) / 256), n), dim3(256), 0, (hipStream_t)stream, pj);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

#ifdef S9_STAMP
extern "C" int acktr_debug_s9_stamps(long long* host_out) {      // diagnostic build only (tools/probe/s9_clock.py)
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(acktr::s9_stamps), sizeof(long long) * 6 * 4096);
}
#endif
