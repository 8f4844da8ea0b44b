// Implicit-GEMM operand functors for NCHW convolutions (forward, dgrad, wgrad) -- see gemm_core.hpp.
// Geometry is compile time (the policy's three conv shapes); batch and channel counts are runtime.
#pragma once
#include "gemm_core.hpp"

namespace acktr {

// 16-byte vector load from a 4-byte-aligned address: ONE global_load_dwordx4 on gfx950
typedef f32x4 f32x4_u __attribute__((aligned(4)));

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
            } else {                          // the 4 taps lie inside the image row: one unaligned 16-byte load
                return *reinterpret_cast<const f32x4_u*>(q);
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
    static constexpr bool HAS_ROWSUM = false;
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
struct DgradEpi {  // (m = c, n = pos) -> dX[b, c, iy, ix] = acc * (X[b,c,iy,ix] > 0 if relu_gate)
    static constexpr bool HAS_ROWSUM = false;
    using G = DgradGeom<KH, KW, S, OH, OW>;
    float* dx; const float* xact; int C; int NP;
    struct Col { long off; int ok; };
    __device__ __forceinline__ Col col(int n) const {
        const int py = (int)blockIdx.z / S, px = (int)blockIdx.z % S;              // stride phase = grid z
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


// ==========================================================================================
// Position-contiguous ("Pos") operands for stride-1 access patterns.
// A vector-memory instruction costs the same issue slots whether it moves 4 or 16 bytes per lane, and the
// scalar gathers of the generic im2col / dgrad loaders are what bounds those kernels.  For a stride-1 window,
// consecutive output positions read consecutive pixels, so the loaders below fetch FOUR positions with one
// 4-byte-aligned global_load_dwordx4.  Rows of the position space are padded to a multiple of 4 (OWP / JWP);
// padding positions are dropped by the epilogue (forward, dgrad) or multiplied by an exact zero from the
// other operand (wgrad).
// ==========================================================================================
__device__ __forceinline__ f32x4 load4_unaligned(const float* q, const float* lo, const float* hi) {
    if (q >= lo && q + 4 <= hi) return *reinterpret_cast<const f32x4_u*>(q);
    f32x4 v = zero4();                                   // tensor edge: guarded scalar loads
#pragma unroll
    for (int j = 0; j < 4; ++j) if (q + j >= lo && q + j < hi) v[j] = q[j];
    return v;
}

// stride-1 im2col: element (kk = (c,ky,kx), padded position pp = (b, oy, oxp))
template <int KH, int KW, int H, int W, int OWP, bool INNER_K>
struct Im2colPos {
    static constexpr bool INNER_IS_K = INNER_K;          // true: positions are the GEMM K (wgrad); false: rows (fwd)
    static constexpr int OH = H - KH + 1, OW = W - KW + 1, PP = OH * OWP, KK = KH * KW, HW = H * W;
    static_assert(OWP % 4 == 0 && OWP >= OW, "padded row length");
    const float* x; const float* x_end; int C; int RP; int Kc;
    struct Outer { int off; int ok; };
    __device__ __forceinline__ Outer prep(int kk) const {
        Outer o; o.ok = kk < Kc; const int k = o.ok ? kk : 0;
        const int c = k / KK, rem = k % KK;
        o.off = c * HW + (rem / KW) * W + (rem % KW);
        return o;
    }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, int p0) const {
        if (!o.ok || p0 >= RP) return zero4();
        const int b = p0 / PP, r = p0 % PP;
        const float* q = x + (long)b * C * HW + o.off + (r / OWP) * W + (r % OWP);
        return load4_unaligned(q, x, x_end);
    }
};

// Y / dY [B, OC, OH*OW] viewed as (oc, padded position); exact zeros at padding positions
template <int OH, int OW, int OWP>
struct ConvOutPos {
    static constexpr bool INNER_IS_K = true;
    static constexpr int SP = OH * OW, PP = OH * OWP;
    const float* p; const float* p_end; int OC; int RP;
    struct Outer { const float* cp; int ok; };
    __device__ __forceinline__ Outer prep(int oc) const {
        Outer o; o.ok = oc < OC; o.cp = p + (long)(o.ok ? oc : 0) * SP; return o;
    }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, int p0) const {
        if (!o.ok || p0 >= RP) return zero4();
        const int b = p0 / PP, r = p0 % PP;
        const int oy = r / OWP, ox0 = r % OWP;
        if (ox0 >= OW) return zero4();
        f32x4 v = load4_unaligned(o.cp + (long)b * OC * SP + oy * OW + ox0, p, p_end);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (ox0 + j >= OW) v[j] = 0.f;
        return v;
    }
};

template <int OH, int OW, int OWP>
struct ConvFwdEpiPos {   // (m = oc, n = padded position) -> Y[b, oc, oy, ox]
    static constexpr bool HAS_ROWSUM = false;
    static constexpr int SP = OH * OW, PP = OH * OWP;
    float* y; const float* bias; int OC; int RP; float scale; int relu;
    struct Col { long off; int ok; };
    __device__ __forceinline__ Col col(int n) const {
        Col c; c.ok = n < RP; const int nn = c.ok ? n : 0;
        const int b = nn / PP, r = nn % PP;
        const int oy = r / OWP, ox = r % OWP;
        if (ox >= OW) c.ok = 0;
        c.off = (long)b * OC * SP + oy * OW + ox; return c;
    }
    __device__ __forceinline__ void st(const Col& c, int m, float acc, int) const {
        if (!c.ok || m >= OC) return;
        float v = acc * scale + (bias ? bias[m] : 0.f);
        if (relu) v = v > 0.f ? v : 0.f;
        y[c.off + (long)m * SP] = v;
    }
};

}  // namespace acktr
