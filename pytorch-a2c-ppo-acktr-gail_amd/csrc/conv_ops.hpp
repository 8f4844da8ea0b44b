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
    // wgrad (INNER_K = false): the reduction runs over positions with each image padded to SPP = roundup(SP, 4)
    // (positions sp >= SP contribute zeros), so that the dY operand can fetch 4 positions of one image per load
    static constexpr int SPP = INNER_K ? SP : (SP + 3) / 4 * 4;
    const float* x; const int64_t* idx; int C; long img_stride; int R; int Kc;     // R counts (padded) positions
    struct Outer { const float* p; int ok; };
    __device__ __forceinline__ Outer prep(int r) const {
        Outer o;
        const bool in = r < R;
        const int rr = in ? r : 0;
        const int b = rr / SPP, sp = rr % SPP;
        o.ok = in && sp < SP;
        const int spc = sp < SP ? sp : 0;
        const int oy = spc / OW, ox = spc % OW;
        const long img = idx ? (long)idx[b] : (long)b;
        o.p = x + img * img_stride + (oy * S) * W + ox * S;
        return o;
    }
    __device__ static __forceinline__ f32x4 post(const f32x4& v, int meta) { return sel4(meta & 1, v); }
    struct Inner { int kk0; int off; int ok; };     // patch index -> offset inside the window (K-invariant for wgrad)
    __device__ __forceinline__ Inner prep_inner(int kk0) const {
        Inner in; in.kk0 = kk0; in.ok = kk0 < Kc;
        const int kc = in.ok ? kk0 : 0;
        const int c = kc / KK, rem = kc % KK;
        in.off = c * HW + (rem / KW) * W + (rem % KW);
        return in;
    }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, const Inner& in, int& meta) const {
        const int kk0 = in.kk0;
        if constexpr (KW % 4 == 0) {          // 4 taps share (c, ky): contiguous in x   (Kc % 4 == 0); branch-free
            meta = o.ok && in.ok;
            const float* q = o.p + in.off;
            if constexpr (S % 4 == 0 && W % 4 == 0) return *reinterpret_cast<const f32x4*>(q);
            else return *reinterpret_cast<const f32x4_u*>(q);          // inside the image row: unaligned 16 bytes
        } else {
            meta = 1;
            f32x4 v = zero4();
            if (!o.ok || kk0 >= Kc) return v;
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

// Y / dY [B, OC, SP] viewed as (oc, k = (b, sp')) with sp' running to SPP = roundup(SP, 4): k-contiguous M-operand
// of wgrad.  A group of 4 k never crosses an image; the last group of an image holds SP % 4 real positions and is
// fetched from the image's last 4 floats and shifted down, so no load ever leaves the tensor.  Branch-free.
template <int SP>
struct ConvOutKC {
    static constexpr bool INNER_IS_K = true;
    static constexpr int SPP = (SP + 3) / 4 * 4, TAIL = SP % 4;      // TAIL real positions in a partial last group
    const float* p; int OC; int R;                                   // R = B * SPP
    struct Outer { const float* cp; int ok; };
    __device__ __forceinline__ Outer prep(int oc) const {
        Outer o; o.ok = oc < OC; o.cp = p + (long)(o.ok ? oc : 0) * SP; return o;
    }
    struct Inner { long off; int m; };               // position group -> offset of its (clamped) first float, flags
    __device__ __forceinline__ Inner prep_inner(int r0) const {
        const bool in = r0 < R;
        const int rc = in ? r0 : 0;
        const int b = rc / SPP, sp = rc % SPP;
        const bool tail = TAIL != 0 && sp + 4 > SP;                  // sp == SPP - 4
        Inner i; i.off = (long)b * OC * SP + (tail ? SP - 4 : sp); i.m = (int)in | ((int)tail << 1);
        return i;
    }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, const Inner& in, int& meta) const {
        meta = o.ok ? in.m : 0;
        if constexpr (TAIL == 0) return *reinterpret_cast<const f32x4*>(o.cp + in.off);
        else return *reinterpret_cast<const f32x4_u*>(o.cp + in.off);
    }
    __device__ static __forceinline__ f32x4 post(const f32x4& v, int meta) {
        if constexpr (TAIL == 0) {
            return sel4(meta & 1, v);
        } else {
            const bool tail = meta & 2;
            f32x4 r;                                                  // tail: real positions are v[4-TAIL .. 3]
            r[0] = tail ? v[4 - TAIL] : v[0];
            r[1] = tail ? (TAIL > 1 ? v[(5 - TAIL) & 3] : 0.f) : v[1];
            r[2] = tail ? (TAIL > 2 ? v[(6 - TAIL) & 3] : 0.f) : v[2];
            r[3] = tail ? 0.f : v[3];
            return sel4(meta & 1, r);
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
    __device__ static __forceinline__ f32x4 post(const f32x4& v, int) { return v; }
    struct Inner { int k0; };
    __device__ __forceinline__ Inner prep_inner(int k0) const { return Inner{k0}; }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, const Inner& in, int& meta) const {
        const int k0 = in.k0;
        meta = 1;
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
// 4 floats from q where only q[0..2] are guaranteed inside the tensor [.., hi): the padded 4th position of a row
// may fall one float behind the tensor's end.  Branch-free: fetch from min(q, hi-4); `clamped` tells post() to
// shift the vector down by one.
__device__ __forceinline__ f32x4 load4_tail_safe(const float* q, const float* hi, bool& clamped) {
    clamped = q + 4 > hi;                                // can only be by one float
    return *reinterpret_cast<const f32x4_u*>(clamped ? hi - 4 : q);
}
__device__ __forceinline__ f32x4 shift_down1(bool cl, const f32x4& v) {
    f32x4 r;
    r[0] = cl ? v[1] : v[0]; r[1] = cl ? v[2] : v[1]; r[2] = cl ? v[3] : v[2]; r[3] = cl ? 0.f : v[3];
    return r;
}

// stride-1 im2col: element (kk = (c,ky,kx), padded position pp = (b, oy, oxp))
// ZERO_PAD: the padded position of a row (oxp = OW) reads as an all-zero patch.  Not needed when the other GEMM operand
// is zero there (wgrad: dY) or the column is never stored (forward); needed when both operands are patches (KFAC's
// A factor), where the over-read window would otherwise be squared into the result.
template <int KH, int KW, int H, int W, int OWP, bool INNER_K, bool ZERO_PAD = false>
struct Im2colPos {
    static constexpr bool INNER_IS_K = INNER_K;          // true: positions are the GEMM K (wgrad); false: rows (fwd)
    static constexpr int OH = H - KH + 1, OW = W - KW + 1, PP = OH * OWP, KK = KH * KW, HW = H * W;
    static_assert(OWP % 4 == 0 && OWP >= OW && OWP - OW <= 1, "padded row length (at most one float of over-read)");
    const float* x; const float* x_end; int C; int RP; int Kc;
    struct Outer { int off; int ok; };
    __device__ __forceinline__ Outer prep(int kk) const {
        Outer o; o.ok = kk < Kc; const int k = o.ok ? kk : 0;
        const int c = k / KK, rem = k % KK;
        o.off = c * HW + (rem / KW) * W + (rem % KW);
        return o;
    }
    struct Inner { long off; int ok; };              // padded position group -> offset of its window origin
    __device__ __forceinline__ Inner prep_inner(int p0) const {
        Inner in; in.ok = p0 < RP;
        const int pc = in.ok ? p0 : 0;
        const int b = pc / PP, r = pc % PP;
        in.off = (long)b * C * HW + (r / OWP) * W + (r % OWP);
        if (ZERO_PAD && OWP > OW && (r % OWP) + 4 > OW) in.ok |= 2;      // this group of four ends on the padded position
        return in;
    }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, const Inner& in, int& meta) const {
        bool cl;
        const f32x4 v = load4_tail_safe(x + in.off + o.off, x_end, cl);
        meta = (int)(o.ok && (in.ok & 1)) | ((int)cl << 1) | ((in.ok & 2) << 1);
        return v;
    }
    __device__ static __forceinline__ f32x4 post(const f32x4& v, int meta) {
        f32x4 r = sel4(meta & 1, shift_down1(meta & 2, v));
        if (ZERO_PAD) r[3] = (meta & 4) ? 0.f : r[3];
        return r;
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
    struct Inner { long off; int m; };
    __device__ __forceinline__ Inner prep_inner(int p0) const {
        static_assert(OWP - OW <= 1, "at most one padded position per row (load4_tail_safe shifts by one)");
        const bool in = p0 < RP;
        const int pc = in ? p0 : 0;
        const int b = pc / PP, r = pc % PP;
        const int oy = r / OWP, ox0 = r % OWP;
        Inner i; i.off = (long)b * OC * SP + oy * OW + ox0; i.m = (int)in | (ox0 << 2);
        return i;
    }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, const Inner& in, int& meta) const {
        bool cl;
        const f32x4 v = load4_tail_safe(o.cp + in.off, p_end, cl);
        meta = (o.ok ? in.m : (in.m & ~1)) | ((int)cl << 1);
        return v;
    }
    __device__ static __forceinline__ f32x4 post(const f32x4& raw, int meta) {
        f32x4 v = shift_down1(meta & 2, raw);
        const int ox0 = meta >> 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (ox0 + j < OW) ? v[j] : 0.f;
        return sel4(meta & 1, v);
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
