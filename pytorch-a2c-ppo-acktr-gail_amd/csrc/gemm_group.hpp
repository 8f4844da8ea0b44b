// Grouped dense GEMMs on the tile-GEMM skeleton: blockIdx.z selects one of up to GROUP_MAX independent problems
// C_z(m, n) = sum_k A_z(m, k) B_z(n, k) whose descriptors travel in the kernel arguments.  Used by the KFAC
// preconditioner (kfac.py:216-227): the four products of v = Qg [(Qg^T G Qa) / (dg da^T + lambda)] Qa^T have the same
// operand layouts for every layer, so stage s of ALL layers is ONE launch -- 4 launches per optimizer step instead of 48
// latency-bound ones.  The grid is sized for the largest
// problem; workgroups outside a problem's tile range leave before touching memory (EpiSym::skip), and each problem runs
// its own K range (OpZK).
#pragma once
#include "gemm_core.hpp"

namespace acktr {

constexpr int GROUP_MAX = 16;
// vec: this operand of this problem allows 16-byte loads (contiguous extent % 4 == 0, ld % 4 == 0, 16-byte aligned base);
// decided per operand AND per problem at run time, so that the 1-column bias modules and the 1- / 6-row heads (whose
// gradient-side operand needs scalar loads) share a launch with the big layers and still read their 512 x 512 eigenbases
// with vector loads (as a separate scalar-load group they cost as much as the group with the 512 x 1568 layer in it).
struct GroupOpJob { const float* p; int rows; int K; long ld; int vec; };
__host__ inline GroupOpJob group_op(const float* p, int rows, int K, long ld, bool kc) {
    const bool vec = (ld % 4 == 0) && ((kc ? K : rows) % 4 == 0) && (((uintptr_t)p) & 15) == 0;
    return GroupOpJob{p, rows, K, ld, vec ? 1 : 0};
}

// KC: element (row, k) at p[row*ld + k];  !KC: element (row, k) at p[k*ld + row]   (cf. DenseKCT / DenseRCT)
template <bool KC>
struct GroupDense {
    static constexpr bool INNER_IS_K = KC;
    GroupOpJob j[GROUP_MAX];
    __device__ __forceinline__ int zK() const { return j[blockIdx.z].K; }
    struct Outer { const float* bp; int ok; };
    __device__ __forceinline__ Outer prep(int outer) const {        // KC: outer = row; RC: outer = k
        const GroupOpJob& J = j[blockIdx.z];
        Outer o; o.ok = outer < (KC ? J.rows : J.K); o.bp = J.p + (long)(o.ok ? outer : 0) * J.ld; return o;
    }
    struct Inner { int i0; };
    __device__ __forceinline__ Inner prep_inner(int i0) const { return Inner{i0}; }
    __device__ __forceinline__ f32x4 ld4(const Outer& o, const Inner& in, int& meta) const {
        const GroupOpJob& J = j[blockIdx.z];
        const int lim = KC ? J.K : J.rows, i0 = in.i0;
        if (J.vec) {                                                 // (uniform per workgroup)
            const bool ok = o.ok && i0 < lim;
            meta = ok;
            return *reinterpret_cast<const f32x4*>(o.bp + (ok ? i0 : 0));
        }
        meta = 1;
        f32x4 v = zero4();
        if (!o.ok || i0 >= lim) return v;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (i0 + q < lim) v[q] = o.bp[i0 + q];
        return v;
    }
    __device__ static __forceinline__ f32x4 post(const f32x4& v, int meta) { return sel4(meta & 1, v); }
};

struct GroupEpiJob { float* c; int M; int N; long ldc; const float* div_m; const float* div_n; };
// C_z[m*ldc + n] = acc, optionally / (div_m[m]*div_n[n] + div_add)
struct GroupEpi {
    GroupEpiJob j[GROUP_MAX];
    float div_add; int use_div;
    static constexpr bool HAS_ROWSUM = false;
    struct Col { int n; int ok; float dn; };
    __device__ __forceinline__ Col col(int n) const {
        const GroupEpiJob& J = j[blockIdx.z];
        Col cc; cc.n = n; cc.ok = n < J.N; cc.dn = (use_div && cc.ok) ? J.div_n[n] : 0.f; return cc;
    }
    __device__ __forceinline__ void st(const Col& cc, int m, float acc, int) const {
        const GroupEpiJob& J = j[blockIdx.z];
        if (!cc.ok || m >= J.M) return;
        float v = acc;
        if (use_div) v = v / (J.div_m[m] * cc.dn + div_add);
        J.c[(long)m * J.ldc + cc.n] = v;
    }
};
template <>
struct EpiSym<GroupEpi> {
    __device__ static __forceinline__ bool skip(const GroupEpi& e, int m0, int n0, int) {
        return m0 >= e.j[blockIdx.z].M || n0 >= e.j[blockIdx.z].N;
    }
};

}  // namespace acktr
