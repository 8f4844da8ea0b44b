// Implicit-GEMM operand functors for NCHW convolutions (forward, dgrad, wgrad) -- see gemm_core.hpp.
// Geometry is compile time (the policy's three conv shapes); batch and channel counts are runtime.
#pragma once
#include "gemm_core.hpp"

namespace acktr {

// ------------------------------------------------------------------------------------------
// im2col patches of X [B(,gathered), C, H, W]:  element (r = (b,oy,ox), kk = (c,ky,kx))
//   INNER_K = true : forward N-operand (rows = positions, k = patch index)
//   INNER_K = false: wgrad   N-operand (k = positions (reduction), rows = patch index)
// `idx` (optional) redirects image b to stored image idx[b]: the minibatch gather of
// storage.py:127 folded into the conv1 loader (no gathered copy of the frames is ever made).
// ------------------------------------------------------------------------------------------
template <int KH, int KW, int S, int H, int W, bool INNER_K>
struct Im2col {
    static constexpr bool INNER_IS_K = INNER_K;
    static constexpr int OH = (H - KH) / S + 1, OW = (W - KW) / S + 1, SP = OH * OW, KK = KH * KW, HW = H * W;
    const float* x; const int64_t* idx; int C; long img_stride; int R; int Kc;
    struct Outer { const float* p; int ok; };
    __device__ __forceinline__ Outer prep(int r) const {
        Outer o; o.ok = r < R;
        const int rr = o.ok ? r : 0;
        const int b = rr / SP, sp = rr % SP;
        const int oy = sp / OW, ox = sp % OW;
        const long img = idx ? (long)idx[b] : (long)b;
        o.p = x + img * img_stride + (oy * S) * W + ox * S;
        return o;
    }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, int kk0) const {
        f32x4 v = zero4();
        if (!o.ok || kk0 >= Kc) return v;
        if constexpr (KW % 4 == 0) {          // 4 taps share (c, ky): contiguous in x   (Kc % 4 == 0)
            const int c = kk0 / KK, rem = kk0 % KK;
            const float* q = o.p + c * HW + (rem / KW) * W + (rem % KW);
            if constexpr (S % 4 == 0 && W % 4 == 0) {
                return *reinterpret_cast<const f32x4*>(q);
            } else if constexpr (S % 2 == 0 && W % 2 == 0) {
                const float2 a = *reinterpret_cast<const float2*>(q);
                const float2 b = *reinterpret_cast<const float2*>(q + 2);
                v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
                return v;
            } else {
                v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
                return v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = kk0 + j;
                if (kk < Kc) {
                    const int c = kk / KK, rem = kk % KK;
                    v[j] = o.p[c * HW + (rem / KW) * W + (rem % KW)];
                }
            }
            return v;
        }
    }
};

// Y / dY [B, OC, SP] viewed as (oc, r = (b, sp));  k-contiguous M-operand of wgrad
template <int SP>
struct ConvOutKC {
    static constexpr bool INNER_IS_K = true;
    const float* p; int OC; int R;
    struct Outer { const float* cp; int ok; };
    __device__ __forceinline__ Outer prep(int oc) const {
        Outer o; o.ok = oc < OC; o.cp = p + (long)(o.ok ? oc : 0) * SP; return o;
    }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, int r0) const {
        f32x4 v = zero4();
        if (!o.ok || r0 >= R) return v;
        if constexpr (SP % 4 == 0) {
            const int b = r0 / SP, sp = r0 % SP;
            return *reinterpret_cast<const f32x4*>(o.cp + (long)b * OC * SP + sp);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = r0 + j;
                if (r < R) { const int b = r / SP, sp = r % SP; v[j] = o.cp[(long)b * OC * SP + sp]; }
            }
            return v;
        }
    }
};

// forward epilogue: (m = oc, n = r) -> Y[b, oc, sp] = relu?(acc*scale + bias[oc])
template <int SP>
struct ConvFwdEpi {
    float* y; const float* bias; int OC; int R; float scale; int relu;
    struct Col { long off; int ok; };
    __device__ __forceinline__ Col col(int n) const {
        Col c; c.ok = n < R; const int nn = c.ok ? n : 0;
        c.off = (long)(nn / SP) * OC * SP + (nn % SP); return c;
    }
    __device__ __forceinline__ void st(const Col& c, int m, float acc, int) const {
        if (!c.ok || m >= OC) return;
        float v = acc * scale + (bias ? bias[m] : 0.f);
        if (relu) v = v > 0.f ? v : 0.f;
        y[c.off + (long)m * SP] = v;
    }
};

// ------------------------------------------------------------------------------------------
// dgrad as a gather-GEMM per stride phase (py, px):
//   dX[b, c, S*jy+py, S*jx+px] = sum_{oc,a,a'} dY[b, oc, jy-a, jx-a'] * W[oc, c, py+S*a, px+S*a']
// only the KH/S x KW/S taps that can reach an input pixel of that phase are visited, so no MFMA
// work is spent on structurally-zero taps of the strided convolution.
// ------------------------------------------------------------------------------------------
template <int KH, int KW, int S, int OH, int OW>
struct DgradGeom {
    static constexpr int IH = (OH - 1) * S + KH, IW = (OW - 1) * S + KW;
    static constexpr int TA = KH / S, TB = KW / S, TAPS = TA * TB;
    static constexpr int JH = (IH + S - 1) / S, JW = (IW + S - 1) / S, JP = JH * JW;
    static constexpr int SP = OH * OW;
    static_assert(KH % S == 0 && KW % S == 0, "kernel must be a multiple of the stride");
};

template <int KH, int KW, int S, int OH, int OW>
struct DgradDy {   // N-operand: rows = input positions of one phase, k = (oc, a, a')
    using G = DgradGeom<KH, KW, S, OH, OW>;
    static constexpr bool INNER_IS_K = true;
    const float* dy; int OC; int NP; int Kred;
    // per position: pointer to dY[b, 0, jy, jx] and a TAPS-bit mask of the taps (a, a') whose source pixel
    // (jy-a, jx-a') lies inside the output map -- all bounds tests are done once per row, not per element
    struct Outer { const float* p; unsigned mask; };
    __device__ __forceinline__ Outer prep(int pos) const {
        Outer o; o.mask = 0u;
        const bool ok = pos < NP; const int pp = ok ? pos : 0;
        const int b = pp / G::JP, j = pp % G::JP;
        const int jy = j / G::JW, jx = j % G::JW;
        o.p = dy + (long)b * OC * G::SP + jy * OW + jx;
        if (ok) {
#pragma unroll
            for (int t = 0; t < G::TAPS; ++t) {
                const int oy = jy - t / G::TB, ox = jx - t % G::TB;
                if (oy >= 0 && oy < OH && ox >= 0 && ox < OW) o.mask |= (1u << t);
            }
        }
        return o;
    }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, int k0) const {
        f32x4 v = zero4();
        if (o.mask == 0u || k0 >= Kred) return v;
        if constexpr (G::TAPS % 4 == 0) {          // 4 consecutive k share one oc (Kred % 4 == 0)
            const int oc = k0 / G::TAPS, t0 = k0 % G::TAPS;
            const float* q = o.p + (long)oc * G::SP;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + j;              // t0 is a multiple of 4: compile-time tap geometry per j
                if ((o.mask >> t) & 1u) v[j] = q[-((t / G::TB) * OW + (t % G::TB))];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j;
                if (k < Kred) {
                    const int oc = k / G::TAPS, t = k % G::TAPS;
                    if ((o.mask >> t) & 1u) v[j] = o.p[(long)oc * G::SP - ((t / G::TB) * OW + (t % G::TB))];
                }
            }
        }
        return v;
    }
};

template <int KH, int KW, int S, int OH, int OW>
struct DgradW {    // M-operand: rows = input channel c, k = (oc, a, a') -> W[oc, c, py+S*a, px+S*a']
    using G = DgradGeom<KH, KW, S, OH, OW>;
    static constexpr bool INNER_IS_K = true;
    const float* w; int C; int Kred; int py, px;
    struct Outer { const float* p; int ok; };
    __device__ __forceinline__ Outer prep(int c) const {
        Outer o; o.ok = c < C; o.p = w + (long)(o.ok ? c : 0) * (KH * KW); return o;
    }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, int k0) const {
        f32x4 v = zero4();
        if (!o.ok) return v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + j;
            if (k < Kred) {
                const int oc = k / G::TAPS, tap = k % G::TAPS;
                v[j] = o.p[(long)oc * C * (KH * KW) + (py + S * (tap / G::TB)) * KW + (px + S * (tap % G::TB))];
            }
        }
        return v;
    }
};

template <int KH, int KW, int S, int OH, int OW>
struct DgradEpi {  // (m = c, n = pos) -> dX[b, c, iy, ix] = acc * (X[b,c,iy,ix] > 0 if relu_gate)
    using G = DgradGeom<KH, KW, S, OH, OW>;
    float* dx; const float* xact; int C; int NP; int py, px;
    struct Col { long off; int ok; };
    __device__ __forceinline__ Col col(int n) const {
        Col c; c.ok = n < NP; const int nn = c.ok ? n : 0;
        const int b = nn / G::JP, j = nn % G::JP;
        const int iy = S * (j / G::JW) + py, ix = S * (j % G::JW) + px;
        if (iy >= G::IH || ix >= G::IW) c.ok = 0;
        c.off = (long)b * C * (G::IH * G::IW) + iy * G::IW + ix;
        return c;
    }
    __device__ __forceinline__ void st(const Col& c, int m, float acc, int) const {
        if (!c.ok || m >= C) return;
        const long o = c.off + (long)m * (G::IH * G::IW);
        float v = acc;
        if (xact) v = xact[o] > 0.f ? v : 0.f;
        dx[o] = v;
    }
};

}  // namespace acktr
