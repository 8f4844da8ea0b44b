// NCHW convolution forward / dgrad / wgrad / bias-grad as implicit GEMMs on fp32 MFMA.
// Replaces the nn.Conv2d call sites of CNNBase (reference a2c_ppo_acktr/model.py:176-179,190) and
// their autograd backward.  Supported geometries are the policy's three layers:
//   8x8 stride 4 on 84x84, 4x4 stride 2 on 20x20, 3x3 stride 1 on 9x9 (any channel counts).
#include "conv_ops.hpp"
#include "../../include/acktr_hip.h"

namespace acktr {
int pick_slices(long tiles, long K, long max_by_ws);

// per-(oc, image slice) partial sums of dY [B, OC, SP] -> partial[slice*OC + oc]
__global__ __launch_bounds__(256) void conv_bias_partial_kernel(const float* dy, int B, int OC, int SP,
                                                                int imgs_per_slice, float* partial) {
    __shared__ float sm[16];
    const int oc = blockIdx.x;
    const int b0 = blockIdx.y * imgs_per_slice;
    int b1 = b0 + imgs_per_slice;
    if (b1 > B) b1 = B;
    const long n = (long)(b1 - b0) * SP;
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) {
        const int b = b0 + (int)(i / SP), sp = (int)(i % SP);
        s += dy[((long)b * OC + oc) * SP + sp];
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) partial[(long)blockIdx.y * OC + oc] = s;
}

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
    DenseKC wop{w, OC, Kc, (long)Kc, vec_ok(w, Kc)};
    Patch pop{x, idx, C, (long)C * H * W, R, Kc};
    ConvFwdEpi<Patch::SP> epi{y, bias, OC, R, scale, relu};
    return launch_gemm<BM, 128, WM, WN>(wop, pop, epi, OC, R, Kc, 1, s);
}

template <int KH, int KW, int S, int H, int W, int BM, int WM, int WN>
static int conv_wgrad_t(const float* x, const int64_t* idx, const float* dy, float* dw, float* db, int B, int C,
                        int OC, float scale, float* ws, long ws_floats, hipStream_t s) {
    using Patch = Im2col<KH, KW, S, H, W, false>;
    constexpr int SP = Patch::SP;
    const int Kc = C * KH * KW, R = B * SP;
    const long bias_floats = db ? (long)64 * OC : 0;
    const long tiles = (long)cdiv(OC, BM) * cdiv(Kc, 128);
    long max_by_ws = (ws_floats - bias_floats) / ((long)OC * Kc) - SLAB_GROUP_EXTRA;
    int ns = pick_slices(tiles, R, max_by_ws < 1 ? 1 : max_by_ws);
    if (ns > 1 && max_by_ws < ns) ns = 1;
    ConvOutKC<SP> gop{dy, OC, R};
    Patch pop{x, idx, C, (long)C * H * W, R, Kc};
    int rc;
    if (ns == 1) {
        RowMajorEpi e = row_major_epi(dw, OC, Kc, Kc);
        e.scale = scale;
        rc = launch_gemm<BM, 128, WM, WN>(gop, pop, e, OC, Kc, R, 1, s);
        if (rc) return rc;
    } else {
        RowMajorEpi e = row_major_epi(ws, OC, Kc, Kc);
        e.slab_stride = (long)OC * Kc;
        rc = launch_gemm<BM, 128, WM, WN>(gop, pop, e, OC, Kc, R, ns, s);
        if (rc) return rc;
        rc = launch_slab_reduce(dw, ws, ns, (long)OC * Kc, (long)OC * Kc, scale, 0, s);
        if (rc) return rc;
    }
    if (db) {
        if (ws_floats < bias_floats + (ns > 1 ? (long)(ns + SLAB_GROUP_EXTRA) * OC * Kc : 0)) {
            set_error("conv2d_wgrad: workspace too small for bias partials");
            return -1;
        }
        float* part = ws + (ns > 1 ? (long)(ns + SLAB_GROUP_EXTRA) * OC * Kc : 0);
        int nsl = B < 64 ? B : 64;
        const int ips = cdiv(B, nsl);
        nsl = cdiv(B, ips);
        hipLaunchKernelGGL(conv_bias_partial_kernel, dim3(OC, nsl), dim3(256), 0, s, dy, B, OC, SP, ips, part);
        ACKTR_LAUNCH_CHECK();
        rc = launch_slab_reduce(db, part, nsl, OC, OC, 1.f, 0, s);
        if (rc) return rc;
    }
    return 0;
}

template <int KH, int KW, int S, int OH, int OW, int BM, int WM, int WN>
static int conv_dgrad_t(const float* dy, const float* w, const float* xact, float* dx, int B, int C, int OC,
                        hipStream_t s) {
    using G = DgradGeom<KH, KW, S, OH, OW>;
    const int Kred = OC * G::TAPS, NP = B * G::JP;
    for (int py = 0; py < S; ++py)
        for (int px = 0; px < S; ++px) {
            DgradW<KH, KW, S, OH, OW> wop{w, C, Kred, py, px};
            DgradDy<KH, KW, S, OH, OW> gop{dy, OC, NP, Kred};
            DgradEpi<KH, KW, S, OH, OW> epi{dx, xact, C, NP, py, px};
            int rc = launch_gemm<BM, 128, WM, WN>(wop, gop, epi, C, NP, Kred, 1, s);
            if (rc) return rc;
        }
    return 0;
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

extern "C" int acktr_conv2d_fwd_f32(const float* x, const int64_t* gather_idx, const float* w, const float* bias,
                                    float* y, int B, int C, int H, int W, int OC, int KH, int KW, int S,
                                    float in_scale, int relu, void* stream) {
    ACKTR_ARG(x && w && y, "conv2d_fwd: null pointer");
    const bool gather_ok = true;
    CONV_COMMON_CHECKS("conv2d_fwd");
    ACKTR_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "conv2d_fwd: x and w must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (geom == GEOM_8S4) {
        if (OC <= 32) return conv_fwd_t<8, 8, 4, 84, 84, 32, 1, 4>(x, gather_idx, w, bias, y, B, C, OC, in_scale, relu, s);
        return conv_fwd_t<8, 8, 4, 84, 84, 64, 2, 2>(x, gather_idx, w, bias, y, B, C, OC, in_scale, relu, s);
    }
    if (geom == GEOM_4S2) {
        if (OC <= 32) return conv_fwd_t<4, 4, 2, 20, 20, 32, 1, 4>(x, gather_idx, w, bias, y, B, C, OC, in_scale, relu, s);
        return conv_fwd_t<4, 4, 2, 20, 20, 64, 2, 2>(x, gather_idx, w, bias, y, B, C, OC, in_scale, relu, s);
    }
    if (OC <= 32) return conv_fwd_t<3, 3, 1, 9, 9, 32, 1, 4>(x, gather_idx, w, bias, y, B, C, OC, in_scale, relu, s);
    return conv_fwd_t<3, 3, 1, 9, 9, 64, 2, 2>(x, gather_idx, w, bias, y, B, C, OC, in_scale, relu, s);
}

extern "C" long acktr_conv2d_wgrad_workspace_bytes(int B, int C, int H, int W, int OC, int KH, int KW, int S) {
    if (geom_of(H, W, KH, KW, S) == GEOM_NONE || B <= 0) return 0;
    const int OH = (H - KH) / S + 1, OW = (W - KW) / S + 1;
    const long Kc = (long)C * KH * KW;
    const int BM = OC <= 32 ? 32 : 64;
    const int ns = pick_slices((long)cdiv(OC, BM) * cdiv(Kc, 128), (long)B * OH * OW, 1024);
    return ((long)(ns + SLAB_GROUP_EXTRA) * OC * Kc + (long)64 * OC) * (long)sizeof(float);
}

extern "C" int acktr_conv2d_wgrad_f32(const float* x, const int64_t* gather_idx, const float* dy, float* dw,
                                      float* db, int B, int C, int H, int W, int OC, int KH, int KW, int S,
                                      float in_scale, void* workspace, long workspace_bytes, void* stream) {
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
    if (geom == GEOM_8S4) return WG(8, 8, 4, 84, 84);
    if (geom == GEOM_4S2) return WG(4, 4, 2, 20, 20);
    return WG(3, 3, 1, 9, 9);
#undef WG
}

extern "C" int acktr_conv2d_dgrad_f32(const float* dy, const float* w, const float* x_act, float* dx, int B, int C,
                                      int H, int W, int OC, int KH, int KW, int S, void* stream) {
    ACKTR_ARG(dy && w && dx, "conv2d_dgrad: null pointer");
    const bool gather_ok = false;
    CONV_COMMON_CHECKS("conv2d_dgrad");
    hipStream_t s = (hipStream_t)stream;
#define DG(KH_, KW_, S_, OH_, OW_)                                                              \
    (C <= 32 ? conv_dgrad_t<KH_, KW_, S_, OH_, OW_, 32, 1, 4>(dy, w, x_act, dx, B, C, OC, s)    \
             : conv_dgrad_t<KH_, KW_, S_, OH_, OW_, 64, 2, 2>(dy, w, x_act, dx, B, C, OC, s))
    if (geom == GEOM_8S4) return DG(8, 8, 4, 20, 20);
    if (geom == GEOM_4S2) return DG(4, 4, 2, 9, 9);
    return DG(3, 3, 1, 7, 7);
#undef DG
}
