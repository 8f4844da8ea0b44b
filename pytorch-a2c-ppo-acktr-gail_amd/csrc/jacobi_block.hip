// Blocked one-sided Jacobi eigensolver for the KFAC factors (reference a2c_ppo_acktr/algo/kfac.py:205-214:
// torch.symeig of every running-average factor, eigenvalues ascending, `d *= (d > 1e-6)`), for 128 < n <= 2048.
// Hand-written for gfx950; replaces rocSOLVER ssyevd for NatureCNN's 256 / 512 / 576 / 1568 factors.
//
// Why Jacobi and why one-sided.  The factors are covariance matrices dominated by one eigenvalue (mean activation,
// ~7e2) above a bulk of 1e-3..1 and a large numerical null space; the KFAC update divides by d_g d_a + 1e-2, so it is
// the SMALL eigenpairs that must be right.  A tridiagonalisation-based solver has a backward error of eps*||A|| in
// absolute terms (measured: the update then sits 1.3e-3 (LAPACK) / 3.3e-3 (rocSOLVER) from the fp64 result on the
// 1568 factor); Jacobi with a relative rotation criterion keeps column-scaled accuracy (measured 1e-5..8e-5 on the
// same factors, tools/jacobi_proto.py).
//
// Algorithm (Hestenes, blocked).  W = A (symmetric: its columns), V = I, both stored TRANSPOSED (Wt[c][r] = column c),
// so a block of 32 columns is 32 contiguous rows.  The nb column blocks meet pairwise in nb-1 round-robin rounds per
// sweep; for a pair (I < J), P = [W_I W_J] (n x 64):
//   K1 gram   : G = P^T P, 64 x 64, fp32 MFMA over row chunks -> partial tiles (II, IJ, JJ) per chunk (fixed order)
//   K2 rotate : sum the partials, diagonalise G by two-sided cyclic Jacobi in LDS (relative criterion
//               |g_pq| > rtol sqrt(g_pp g_qq) + nu2), U = product of the rotations, columns by descending diagonal
//   K3 apply  : P <- P U and V_IJ <- V_IJ U as MFMA products, in place
// At convergence the columns of W are orthogonal: w_j = lambda_j v_j, lambda_j = (v_j . w_j) / (v_j . v_j).
// All matrices of a call are batched into the same three launches per round; the host reads the per-matrix rotation
// counters once per global sweep and retires converged matrices (one stream sync per sweep).
#include <stdlib.h>

#include <vector>

#include "common.hpp"
#include "../../include/acktr_hip.h"

namespace acktr {

typedef float jb_f4 __attribute__((ext_vector_type(4)));
typedef float jb_f16 __attribute__((ext_vector_type(16)));

constexpr int JB = 32;            // block width (columns)
constexpr int JP = 2 * JB;        // columns of a pair panel
constexpr int JB_RCH = 256;       // rows (r) per workgroup chunk in gram / apply; ldr is a multiple of it
constexpr int JB_MAX_MATS = 64;
constexpr int JB_MAX_N = 2048;
constexpr int JB_INNER_SWEEPS = 2;

struct JbMat {
    const float* a;      // n x n input (row major, symmetric; upper triangle authoritative)
    float* d;            // n eigenvalues out (ascending, thresholded)
    float* qt;           // n x n out, row j = eigenvector j
    float* wt;           // [npad][ldr]
    float* vt;           // [npad][ldr]
    float* gpart;        // [npairs][nch][3][JB*JB]
    float* ubuf;         // [npairs][JP][JP]
    float* lam;          // [npad] eigenvalue per column   (finalize)
    float* inv;          // [npad] 1/||v|| per column, 0 for padding columns
    int* rank;           // [npad]
    int n, nb, npad, ldr, nch, npairs;
    int pair_base;       // first global pair slot of this matrix
};

constexpr int JB_MAX_SLOTS = JB_MAX_MATS * (JB_MAX_N / JB / 2);
struct JbTable {
    JbMat m[JB_MAX_MATS];
    int count;
};
// pair slots of the matrices still iterating: blockIdx.x -> (matrix, pair); rebuilt by the host when a matrix retires
struct JbSlots {
    short mat[JB_MAX_SLOTS];
    short pair[JB_MAX_SLOTS];
};

// device-resident per-call state: table + control words
struct JbCtl {
    int active[JB_MAX_MATS];
    int rot[JB_MAX_MATS];        // rotations applied since the host last cleared it
    int colmax2_bits[JB_MAX_MATS];   // bits of the largest squared column norm; nu2 = 1e-14 * that: absolute floor of the rotation criterion
};


// round-robin ("circle") pairing of `nb` players (even) in round rr of nb-1; pair k of nb/2
__device__ __forceinline__ void jb_pair(int nb, int rr, int k, int& I, int& J) {
    int p, q;
    if (k == 0) { p = nb - 1; q = rr; }
    else { p = (rr + k) % (nb - 1); q = (rr - k + (nb - 1)) % (nb - 1); }
    I = p < q ? p : q;
    J = p < q ? q : p;
}

// ------------------------------------------------------------------------------------------ init
// Wt[c][r] = A[min(c,r)][max(c,r)] (upper triangle, like symeig's default), Vt = I, zero padding
__global__ __launch_bounds__(256) void jb_init_kernel(const JbTable* tab, int mi) {
    const JbMat& M = tab->m[mi];
    const long total = (long)M.npad * M.ldr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i / M.ldr), r = (int)(i % M.ldr);
        float w = 0.f, v = 0.f;
        if (c < M.n && r < M.n) {
            w = c <= r ? M.a[(long)c * M.n + r] : M.a[(long)r * M.n + c];
            v = c == r ? 1.f : 0.f;
        }
        M.wt[i] = w;
        M.vt[i] = v;
    }
}
// colmax2[mi] = largest squared column norm (after init): one wave per column, integer atomic max on the bits of the
// non-negative float (order preserving); the rotation criterion's absolute floor is nu2 = 1e-14 * colmax2
__global__ __launch_bounds__(256) void jb_scale_kernel(const JbTable* tab, JbCtl* ctl) {
    const int mi = blockIdx.y;
    const JbMat& M = tab->m[mi];
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= M.n) return;
    float s = 0.f;
    const float* row = M.wt + (long)c * M.ldr;
    for (int r = lane; r < M.n; r += 64) s += row[r] * row[r];
    s = wave_sum(s);
    if (lane == 0) atomicMax(&ctl->colmax2_bits[mi], __float_as_int(s));
}

// ------------------------------------------------------------------------------------------ K1: Gram partials
// grid (total pair slots, max chunks); 256 threads.  Wave w takes r in [chunk*256 + 64 w, +64): 4 k-steps of 16 rows;
// lane (i = l%32, h = l/32) holds rows r0 + 8h .. 8h+7 of column i of each block (two 16-byte loads per block), the
// same k permutation on both operands, so products pair up; three accumulators (II, IJ, JJ) per wave, summed over the
// four waves through LDS in a fixed order.
__global__ __launch_bounds__(256) void jb_gram_kernel(const JbTable* tab, const JbSlots* slots, int g) {
    __shared__ float red[4][3][JB * JB / 4 + 0];   // reduced in four passes of 256 floats per tile: see below
    const int mi = slots->mat[blockIdx.x];
    const JbMat& M = tab->m[mi];
    if ((int)blockIdx.y >= M.nch) return;
    const int k = slots->pair[blockIdx.x];
    int I, J;
    jb_pair(M.nb, g % (M.nb - 1), k, I, J);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 31, lh = lane >> 5;
    const float* pi = M.wt + (long)(I * JB + li) * M.ldr + blockIdx.y * JB_RCH + wid * 64 + 8 * lh;
    const float* pj = M.wt + (long)(J * JB + li) * M.ldr + blockIdx.y * JB_RCH + wid * 64 + 8 * lh;
    jb_f4 a[4][2], b[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a[s][0] = *reinterpret_cast<const jb_f4*>(pi + 16 * s);
        a[s][1] = *reinterpret_cast<const jb_f4*>(pi + 16 * s + 4);
        b[s][0] = *reinterpret_cast<const jb_f4*>(pj + 16 * s);
        b[s][1] = *reinterpret_cast<const jb_f4*>(pj + 16 * s + 4);
    }
    jb_f16 cii = {0}, cij = {0}, cjj = {0};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                cii = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][u][t], a[s][u][t], cii, 0, 0, 0);
                cij = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][u][t], b[s][u][t], cij, 0, 0, 0);
                cjj = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s][u][t], b[s][u][t], cjj, 0, 0, 0);
            }
    // cross-wave sum, four register groups at a time (LDS: 4 waves x 3 tiles x 256 floats = 12 KB)
    float* out = M.gpart + ((long)(k * M.nch + blockIdx.y) * 3) * (JB * JB);
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            red[wid][0][t * 64 + lane] = cii[grp * 4 + t];
            red[wid][1][t * 64 + lane] = cij[grp * 4 + t];
            red[wid][2][t * 64 + lane] = cjj[grp * 4 + t];
        }
        __syncthreads();
        // 3 tiles x 256 values; thread tid sums value tid of each tile over the four waves
#pragma unroll
        for (int tile = 0; tile < 3; ++tile) {
            const float v = (red[0][tile][tid] + red[1][tile][tid]) + (red[2][tile][tid] + red[3][tile][tid]);
            const int t = tid >> 6, l = tid & 63;                     // register t of group grp, lane l
            const int row = 4 * (l >> 5) + t + 8 * grp, col = l & 31; // accumulator layout of v_mfma_f32_32x32x2
            out[tile * (JB * JB) + row * JB + col] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ K2: 64 x 64 Jacobi
// One workgroup per pair.  G and U in LDS (pitch 65).  Per rotation round (63 per sweep, 32 disjoint pairs):
//   A: lanes 0..31 compute (c, s) of their pair (or the identity below the criterion);
//   B: every 2x2 block (k1, k2) of G is replaced by J_k1^T G_block J_k2 in ONE pass (each item touches only its own
//      four elements, so no barrier between the column and the row half of the similarity), and U <- U J.
// Systolic (Brent-Luk) form: the 64 indices sit in POSITIONS, pair k is always positions (2k, 2k+1); after every round
// the data moves by one fixed permutation (top row shifts right, bottom row shifts left, position 0 stays), written
// straight to its new place in the other LDS buffer.  Every address is a per-thread constant computed before the loop,
// so a round is: rotation parameters (32 lanes) | barrier | one pass over G (2 x 2 blocks, J_k1^T G_blk J_k2) and over
// U (U J) | barrier -- ~190 instructions per thread instead of ~450 with index arithmetic per element.
constexpr int JG = JP + 2;        // even pitch: the (p, q) element pairs are read as 8-byte LDS accesses
// CROSS schedule (every outer round but the first of a sweep): the top row holds block I's 32 columns, the bottom row
// block J's; only the 32 x 32 cross pairs are rotated -- the bottom row shifts by one position per round, 32 rounds, and
// ends where it started.  The pairs inside a block meet once per sweep, in the sweep's first outer round (FULL schedule,
// 63 rounds over all 2016 pairs).  Together: every column pair of the matrix is rotated exactly once per sweep, which is
// plain cyclic Jacobi in a block-friendly order -- half the rotation rounds of diagonalising every 64 x 64 Gram fully.
__device__ __forceinline__ int jb_next_pos_cross(int s) {
    if ((s & 1) == 0) return s;             // top row stays
    return s == 1 ? JP - 1 : s - 2;         // bottom k -> bottom k-1 (cyclic)
}
__device__ __forceinline__ int jb_next_pos(int s) {
    // position s = 2k (top k) or 2k+1 (bottom k), h = 32 pairs
    if (s == 0) return 0;                   // top 0 stays
    if (s == 1) return 2;                   // bottom 0 -> top 1
    if ((s & 1) == 0) return s == JP - 2 ? JP - 1 : s + 2;   // top k -> top k+1 ; last top -> last bottom
    return s - 2;                           // bottom k -> bottom k-1
}
__global__ __launch_bounds__(256) void jb_rotate_kernel(const JbTable* tab, const JbSlots* slots, JbCtl* ctl, int g, float rtol,
                                                        int inner_sweeps) {
    // initial positions: index i of block I (0..31) -> top i (2 i), index 32 + j of block J -> bottom j (2 j + 1)
    __shared__ __attribute__((aligned(16))) float Gs[2][JP * JG];
    __shared__ __attribute__((aligned(16))) float Us[2][JP * JG];
    __shared__ float2 cs[2][JB];
    __shared__ int cnt_round[2];
    __shared__ int flag;
    __shared__ int order[JP];
    const int mi = slots->mat[blockIdx.x];
    const JbMat& M = tab->m[mi];
    const int k = slots->pair[blockIdx.x];
    const int tid = threadIdx.x;
    const float nu = 1e-7f * sqrtf(__int_as_float(ctl->colmax2_bits[mi]));
    const float nu2 = nu * nu, rtol2 = rtol * rtol;
    // deflation: a column whose norm has fallen below 2e-7 of the largest one is numerically in the null space (its
    // direction is rounding noise); rotating it against the others never converges and changes nothing that matters
    // (V stays orthogonal by construction, A v = w stays tiny), so such pairs are left alone
    const float dfl2 = 4.f * nu2;
    // G = sum of the chunk partials (fixed order); tiles II, IJ, JJ -> full symmetric 64 x 64; U = I
    {
        const jb_f4* gp = reinterpret_cast<const jb_f4*>(M.gpart + (long)k * M.nch * 3 * (JB * JB));
        jb_f4 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = jb_f4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < M.nch; ++c)                       // 3 independent 16-byte loads per chunk, chunks in order
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j] += gp[(long)c * 3 * (JB * JB / 4) + tid + 256 * j];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = 4 * (tid + 256 * j) + u;
                const float s = acc[j][u];
                const int tile = e / (JB * JB), row = (e / JB) % JB, col = e % JB;
                const int tr = 2 * row, tc = 2 * col, br = 2 * row + 1, bc = 2 * col + 1;   // top / bottom positions
                if (tile == 0) Gs[0][tr * JG + tc] = s;
                else if (tile == 2) Gs[0][br * JG + bc] = s;
                else { Gs[0][tr * JG + bc] = s; Gs[0][bc * JG + tr] = s; }
            }
        // U rows = original column index of the panel (0..63), U columns = positions
        for (int e = tid; e < JP * JP; e += 256) {
            const int r = e >> 6, c = e & 63, pos_r = r < JB ? 2 * r : 2 * (r - JB) + 1;
            Us[0][r * JG + c] = (pos_r == c) ? 1.f : 0.f;
        }
        if (tid == 0) flag = 0;
    }
    __syncthreads();
    // anything to do at all?  (late sweeps: most pairs of blocks are already orthogonal)
    {
        bool any = false;
        for (int e = tid; e < JP * JP; e += 256) {
            const int i = e >> 6, j = e & 63;
            if (i < j) {
                const float ex = fabsf(Gs[0][i * JG + j]) - nu2;
                const float gi = Gs[0][i * JG + i], gj = Gs[0][j * JG + j];
                any = any || (ex > 0.f && ex * ex > rtol2 * fabsf(gi * gj) && fminf(gi, gj) > dfl2);
            }
        }
        if (any) flag = 1;                       // benign race: every writer stores 1
    }
    __syncthreads();
    int cur = 0, total_rot = 0;                  // total_rot is only meaningful in thread 0
    const bool full = (g % (M.nb - 1)) == 0;      // first outer round of this matrix's sweep: pairs inside the blocks too
    const int nrounds = full ? JP - 1 : JB;
    if (flag) {
        // per-thread constant addresses (floats): G block items (k1 = tid/32 + 8 it, k2 = tid%32), U items (row i, pair kk)
        int g_rd[4], g_w[4][4], u_rd[8], u_w[8][2];
        const int k2 = tid & 31;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int k1 = (tid >> 5) + 8 * it;
            g_rd[it] = (2 * k1) * JG + 2 * k2;
            auto np = [&](int pos) { return full ? jb_next_pos(pos) : jb_next_pos_cross(pos); };
            const int rp = np(2 * k1) * JG, rq = np(2 * k1 + 1) * JG;
            const int cp = np(2 * k2), cq = np(2 * k2 + 1);
            g_w[it][0] = rp + cp; g_w[it][1] = rp + cq; g_w[it][2] = rq + cp; g_w[it][3] = rq + cq;
        }
        const int ui = tid & 63;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int kk = (tid >> 6) + 4 * it;
            u_rd[it] = ui * JG + 2 * kk;
            u_w[it][0] = ui * JG + (full ? jb_next_pos(2 * kk) : jb_next_pos_cross(2 * kk));
            u_w[it][1] = ui * JG + (full ? jb_next_pos(2 * kk + 1) : jb_next_pos_cross(2 * kk + 1));
        }
        // Fast reciprocal / rsqrt: an error of a few ulp in (c, s) makes J a rotation times (1 + 1e-7) -- a common scale
        // of the two columns, which neither the relative criterion nor the final normalisation sees.
        for (int sweep = 0; sweep < inner_sweeps; ++sweep) {
            int sweep_rot = 0;                   // thread 0
            for (int rr = 0; rr < nrounds; ++rr) {
                const int db = rr & 1;
                const float* Gc = Gs[cur];
                if (tid < 64) {                  // wave 0; lanes 32..63 only take part in the ballot
                    bool act = false;
                    if (tid < JB) {
                        const float2 top = *reinterpret_cast<const float2*>(Gc + (2 * tid) * JG + 2 * tid);   // g_pp, g_pq
                        const float aqq = Gc[(2 * tid + 1) * JG + 2 * tid + 1];
                        const float app = top.x, apq = top.y;
                        const float ex = fabsf(apq) - nu2;            // |g_pq| > rtol sqrt(g_pp g_qq) + nu^2
                        act = ex > 0.f && ex * ex > rtol2 * fabsf(app * aqq) && fminf(app, aqq) > dfl2;
                        float c = 1.f, sn = 0.f;
                        if (act) {
                            const float tau = (aqq - app) * __frcp_rn(2.f * apq);
                            const float w = 1.f + tau * tau;
                            const float t = (tau >= 0.f ? 1.f : -1.f) * __frcp_rn(fabsf(tau) + w * __frsqrt_rn(w));
                            c = __frsqrt_rn(1.f + t * t);
                            sn = t * c;
                        }
                        cs[db][tid] = make_float2(c, sn);
                    }
                    const unsigned long long bal = __ballot(act);
                    if (tid == 0) sweep_rot += __popcll(bal);
                }
                __syncthreads();
                float* Gn = Gs[cur ^ 1];
                const float* Uc = Us[cur];
                float* Un = Us[cur ^ 1];
                const float2 r2 = cs[db][k2];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int k1 = (tid >> 5) + 8 * it;
                    const float2 r1 = cs[db][k1];
                    const float2 rowp = *reinterpret_cast<const float2*>(Gc + g_rd[it]);        // [p1][p2], [p1][q2]
                    const float2 rowq = *reinterpret_cast<const float2*>(Gc + g_rd[it] + JG);   // [q1][p2], [q1][q2]
                    // columns: col p' = c col_p - s col_q ; col q' = s col_p + c col_q
                    const float tpp = r2.x * rowp.x - r2.y * rowp.y, tpq = r2.y * rowp.x + r2.x * rowp.y;
                    const float tqp = r2.x * rowq.x - r2.y * rowq.y, tqq = r2.y * rowq.x + r2.x * rowq.y;
                    // rows: row p' = c row_p - s row_q ; row q' = s row_p + c row_q
                    float npp = r1.x * tpp - r1.y * tqp, npq = r1.x * tpq - r1.y * tqq;
                    float nqp = r1.y * tpp + r1.x * tqp, nqq = r1.y * tpq + r1.x * tqq;
                    if (k1 == k2 && r1.y != 0.f) { npq = 0.f; nqp = 0.f; }      // the annihilated element
                    Gn[g_w[it][0]] = npp; Gn[g_w[it][1]] = npq; Gn[g_w[it][2]] = nqp; Gn[g_w[it][3]] = nqq;
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const float2 r = cs[db][(tid >> 6) + 4 * it];
                    const float2 u = *reinterpret_cast<const float2*>(Uc + u_rd[it]);
                    Un[u_w[it][0]] = r.x * u.x - r.y * u.y;
                    Un[u_w[it][1]] = r.y * u.x + r.x * u.y;
                }
                cur ^= 1;
                __syncthreads();
            }
            if (tid == 0) { total_rot += sweep_rot; cnt_round[0] = sweep_rot; }
            __syncthreads();
            const int sr = cnt_round[0];
            __syncthreads();
            if (sr == 0) break;
        }
    }
    // columns by descending diagonal (stable), so the larger columns collect in the lower-numbered block
    const float* Gf = Gs[cur];
    const float* Uf = Us[cur];
    if (tid < JP) {
        const float di = Gf[tid * JG + tid];
        int r = 0;
        for (int j = 0; j < JP; ++j) {
            const float dj = Gf[j * JG + j];
            r += (dj > di || (dj == di && j < tid)) ? 1 : 0;
        }
        order[tid] = r;
    }
    __syncthreads();
    float* uo = M.ubuf + (long)k * JP * JP;
    for (int e = tid; e < JP * JP; e += 256) {
        const int row = e >> 6, col = e & 63;
        uo[row * JP + order[col]] = Uf[row * JG + col];
    }
    if (tid == 0 && total_rot) atomicAdd(&ctl->rot[mi], total_rot);
}

// ------------------------------------------------------------------------------------------ K3: apply
// grid (pair slots, chunks, 2: W / V).  X'[c][r] = sum_k U[k][c] X[k][r] over the pair's 64 rows k of Xt: the M operand
// is U^T (lane (c, h) holds U[k0 + h][c]), the N operand a row segment of Xt (lane (j, h) holds X[k0 + h][r0 + j],
// coalesced), the result rows c are stored back coalesced.  Each wave owns two 32-wide r tiles of the chunk and reads
// all 64 rows of a tile before it writes any.
__global__ __launch_bounds__(256) void jb_apply_kernel(const JbTable* tab, const JbSlots* slots, int g) {
    const int mi = slots->mat[blockIdx.x];
    const JbMat& M = tab->m[mi];
    if ((int)blockIdx.y * 64 >= M.ldr) return;
    const int k = slots->pair[blockIdx.x];
    int I, J;
    jb_pair(M.nb, g % (M.nb - 1), k, I, J);
    float* X = blockIdx.z ? M.vt : M.wt;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int ct = wid & 1;                                            // output column tile: 0 -> block I, 1 -> block J
    const int r0 = blockIdx.y * 64 + (wid >> 1) * 32;
    const float* U = M.ubuf + (long)k * JP * JP + ct * 32 + li;
    const float* xi = X + (long)(I * JB + lh) * M.ldr + r0 + li;      // panel rows 0..31  (k = 2 s + lh, s < 16)
    const float* xj = X + (long)(J * JB + lh) * M.ldr + r0 + li;      // panel rows 32..63 (s >= 16)
    float uf[32], x[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) uf[s] = U[(2 * s + lh) * JP];        // U^T fragment: U[k = 2 s + lh][c]
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        x[s] = xi[(long)(2 * s) * M.ldr];
        x[16 + s] = xj[(long)(2 * s) * M.ldr];
    }
    jb_f16 acc = {0};
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[s], x[s], acc, 0, 0, 0);
    // every wave of the workgroup has read its operands before any of them overwrites the panel (the two column tiles of
    // one r tile read the same 64 rows)
    __syncthreads();
    float* out = X + (long)((ct ? J : I) * JB) * M.ldr + r0 + li;
#pragma unroll
    for (int v = 0; v < 16; ++v) out[(long)(4 * lh + (v & 3) + 8 * (v >> 2)) * M.ldr] = acc[v];
}

// ------------------------------------------------------------------------------------------ finalize
// lam[c] = (v_c . w_c) / (v_c . v_c), inv[c] = 1/||v_c|| (0 for the zero padding columns): one wave per column
__global__ __launch_bounds__(256) void jb_lambda_kernel(const JbTable* tab, int mi) {
    const JbMat& M = tab->m[mi];
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= M.npad) return;
    const float* w = M.wt + (long)c * M.ldr;
    const float* v = M.vt + (long)c * M.ldr;
    double vw = 0.0, vv = 0.0;
    for (int r = lane; r < M.n; r += 64) { vw += (double)v[r] * (double)w[r]; vv += (double)v[r] * (double)v[r]; }
    vw = wave_sum(vw);
    vv = wave_sum(vv);
    if (lane == 0) {
        const bool real = vv > 0.25;
        M.lam[c] = real ? (float)(vw / vv) : 0.f;
        M.inv[c] = real ? (float)(1.0 / sqrt(vv)) : 0.f;
    }
}
// ascending rank among the real columns (stable)
__global__ __launch_bounds__(256) void jb_rank_kernel(const JbTable* tab, int mi) {
    const JbMat& M = tab->m[mi];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= M.npad) return;
    if (M.inv[c] == 0.f) { M.rank[c] = -1; return; }
    const float di = M.lam[c];
    int r = 0;
    for (int j = 0; j < M.npad; ++j) {
        const float dj = M.lam[j];
        r += (M.inv[j] != 0.f && (dj < di || (dj == di && j < c))) ? 1 : 0;
    }
    M.rank[c] = r;
}
__global__ __launch_bounds__(256) void jb_write_kernel(const JbTable* tab, int mi, float thresh) {
    const JbMat& M = tab->m[mi];
    const int c = blockIdx.x;
    const int rk = M.rank[c];
    if (rk < 0 || rk >= M.n) return;
    const float inv = M.inv[c];
    const float* v = M.vt + (long)c * M.ldr;
    float* q = M.qt + (long)rk * M.n;
    for (int r = threadIdx.x; r < M.n; r += 256) q[r] = v[r] * inv;
    if (threadIdx.x == 0) { const float l = M.lam[c]; M.d[rk] = l > thresh ? l : 0.f; }
}

static inline long jb_align(long x) { return (x + 255) & ~255L; }
static void jb_layout(int n, int& nb, int& npad, int& ldr, int& nch, int& npairs) {
    nb = cdiv(n, JB);
    if (nb & 1) ++nb;
    npad = nb * JB;
    ldr = cdiv(n, JB_RCH) * JB_RCH;
    nch = ldr / JB_RCH;
    npairs = nb / 2;
}
static long jb_mat_bytes(int n) {
    int nb, npad, ldr, nch, npairs;
    jb_layout(n, nb, npad, ldr, nch, npairs);
    long b = 0;
    b += 2 * jb_align((long)npad * ldr * 4);                 // wt, vt
    b += jb_align((long)npairs * nch * 3 * JB * JB * 4);     // gram partials
    b += jb_align((long)npairs * JP * JP * 4);               // U
    b += 3 * jb_align((long)npad * 4);                       // lam, inv, rank
    return b;
}
}  // namespace acktr

using namespace acktr;

extern "C" int acktr_jacobi_block_max_n(void) { return JB_MAX_N; }

extern "C" long acktr_sym_eig_jacobi_block_workspace_bytes(const int* ns, int count) {
    if (!ns || count < 0 || count > JB_MAX_MATS) return -1;
    long b = jb_align(sizeof(JbTable)) + jb_align(sizeof(JbCtl)) + jb_align(sizeof(JbSlots));
    for (int i = 0; i < count; ++i) {
        if (ns[i] < 1 || ns[i] > JB_MAX_N) return -1;
        b += jb_mat_bytes(ns[i]);
    }
    return b;
}

extern "C" int acktr_sym_eig_jacobi_block_f32(const acktr_jacobi_job* jobs, int count, float thresh, void* workspace,
                                              long workspace_bytes, int max_sweeps, int* sweeps_out, void* stream) {
    ACKTR_ARG(jobs && count >= 0, "sym_eig_jacobi_block: bad arguments");
    if (count == 0) return 0;
    ACKTR_ARG(count <= JB_MAX_MATS, "sym_eig_jacobi_block: at most %d matrices per call (got %d)", JB_MAX_MATS, count);
    ACKTR_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "sym_eig_jacobi_block: workspace must be 256-byte aligned");
    if (max_sweeps <= 0) max_sweeps = 24;
    hipStream_t s = (hipStream_t)stream;
    std::vector<int> ns(count);
    for (int i = 0; i < count; ++i) {
        ACKTR_ARG(jobs[i].a && jobs[i].d && jobs[i].qt, "sym_eig_jacobi_block: null pointer in job %d", i);
        ACKTR_ARG(jobs[i].n > JB && jobs[i].n <= JB_MAX_N, "sym_eig_jacobi_block: n=%d outside (%d, %d]", jobs[i].n, JB, JB_MAX_N);
        ns[i] = jobs[i].n;
    }
    const long need = acktr_sym_eig_jacobi_block_workspace_bytes(ns.data(), count);
    ACKTR_ARG(workspace_bytes >= need, "sym_eig_jacobi_block: workspace of %ld bytes, need %ld", workspace_bytes, need);

    // carve the workspace
    static thread_local JbTable host_tab;
    char* base = (char*)workspace;
    JbTable* dtab = (JbTable*)base;
    base += jb_align(sizeof(JbTable));
    JbCtl* dctl = (JbCtl*)base;
    base += jb_align(sizeof(JbCtl));
    JbSlots* dslots = (JbSlots*)base;
    base += jb_align(sizeof(JbSlots));
    static thread_local JbSlots host_slots;
    host_tab.count = count;
    int slots = 0, max_ch = 0, max_rounds = 0;
    for (int i = 0; i < count; ++i) {
        JbMat& M = host_tab.m[i];
        M.a = jobs[i].a; M.d = jobs[i].d; M.qt = jobs[i].qt; M.n = jobs[i].n;
        jb_layout(M.n, M.nb, M.npad, M.ldr, M.nch, M.npairs);
        M.wt = (float*)base; base += jb_align((long)M.npad * M.ldr * 4);
        M.vt = (float*)base; base += jb_align((long)M.npad * M.ldr * 4);
        M.gpart = (float*)base; base += jb_align((long)M.npairs * M.nch * 3 * JB * JB * 4);
        M.ubuf = (float*)base; base += jb_align((long)M.npairs * JP * JP * 4);
        M.lam = (float*)base; base += jb_align((long)M.npad * 4);
        M.inv = (float*)base; base += jb_align((long)M.npad * 4);
        M.rank = (int*)base; base += jb_align((long)M.npad * 4);
        M.pair_base = slots;
        slots += M.npairs;
        if (M.nch > max_ch) max_ch = M.nch;
    }
    hipError_t e = hipMemcpyAsync(dtab, &host_tab, sizeof(JbTable), hipMemcpyHostToDevice, s);
    ACKTR_ARG(e == hipSuccess, "sym_eig_jacobi_block: table upload failed: %s", hipGetErrorString(e));
    int active[JB_MAX_MATS] = {0}, rot[JB_MAX_MATS] = {0}, sweeps[JB_MAX_MATS] = {0};
    for (int i = 0; i < count; ++i) active[i] = 1;
    e = hipMemsetAsync(dctl, 0, sizeof(JbCtl), s);
    ACKTR_ARG(e == hipSuccess, "sym_eig_jacobi_block: control reset failed: %s", hipGetErrorString(e));
    e = hipMemcpyAsync(dctl->active, active, sizeof(active), hipMemcpyHostToDevice, s);
    ACKTR_ARG(e == hipSuccess, "sym_eig_jacobi_block: control upload failed: %s", hipGetErrorString(e));
    for (int i = 0; i < count; ++i) {
        const JbMat& M = host_tab.m[i];
        int nbk = cdiv((long)M.npad * M.ldr, 256 * 8);
        if (nbk > 2048) nbk = 2048;
        hipLaunchKernelGGL(jb_init_kernel, dim3(nbk), dim3(256), 0, s, dtab, i);
    }
    int max_n = 0;
    for (int i = 0; i < count; ++i) max_n = host_tab.m[i].n > max_n ? host_tab.m[i].n : max_n;
    hipLaunchKernelGGL(jb_scale_kernel, dim3(cdiv(max_n, 4), count), dim3(256), 0, s, dtab, dctl);
    ACKTR_LAUNCH_CHECK();

    const float rtol = 1e-6f;
    static const int inner_sweeps = []() { const char* ev = getenv("ACKTR_JB_INNER"); int v = ev ? atoi(ev) : JB_INNER_SWEEPS; return v < 1 ? 1 : v; }();
    int g = 0, nactive = count;
    bool slots_dirty = true;
    for (int gs = 0; gs < max_sweeps && nactive > 0; ++gs) {
        max_rounds = 0;
        for (int i = 0; i < count; ++i)
            if (active[i] && host_tab.m[i].nb - 1 > max_rounds) max_rounds = host_tab.m[i].nb - 1;
        if (slots_dirty) {                     // (the previous upload was consumed: every sweep ends on a stream sync)
            slots = 0; max_ch = 0;
            for (int i = 0; i < count; ++i) {
                if (!active[i]) continue;
                for (int kk = 0; kk < host_tab.m[i].npairs; ++kk) { host_slots.mat[slots] = (short)i; host_slots.pair[slots] = (short)kk; ++slots; }
                if (host_tab.m[i].nch > max_ch) max_ch = host_tab.m[i].nch;
            }
            e = hipMemcpyAsync(dslots, &host_slots, sizeof(JbSlots), hipMemcpyHostToDevice, s);
            ACKTR_ARG(e == hipSuccess, "sym_eig_jacobi_block: slot table upload failed: %s", hipGetErrorString(e));
            slots_dirty = false;
        }
        for (int r = 0; r < max_rounds; ++r, ++g) {
            hipLaunchKernelGGL(jb_gram_kernel, dim3(slots, max_ch), dim3(256), 0, s, dtab, dslots, g);
            hipLaunchKernelGGL(jb_rotate_kernel, dim3(slots), dim3(256), 0, s, dtab, dslots, dctl, g, rtol, inner_sweeps);
            hipLaunchKernelGGL(jb_apply_kernel, dim3(slots, max_ch * (JB_RCH / 64), 2), dim3(256), 0, s, dtab, dslots, g);
        }
        ACKTR_LAUNCH_CHECK();
        e = hipMemcpyAsync(rot, dctl->rot, sizeof(int) * count, hipMemcpyDeviceToHost, s);
        ACKTR_ARG(e == hipSuccess, "sym_eig_jacobi_block: counter read failed: %s", hipGetErrorString(e));
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) { set_error("sym_eig_jacobi_block: stream sync failed: %s", hipGetErrorString(e)); return (int)e; }
        static const bool verbose = getenv("ACKTR_JB_VERBOSE") != nullptr;
        if (verbose) {
            fprintf(stderr, "jacobi_block sweep %d (%d rounds):", gs, max_rounds);
            for (int i = 0; i < count; ++i) fprintf(stderr, " n%d:%d", host_tab.m[i].n, active[i] ? rot[i] : -1);
            fprintf(stderr, "\n");
        }
        bool changed = false;
        for (int i = 0; i < count; ++i) {
            if (!active[i]) continue;
            ++sweeps[i];
            if (rot[i] == 0) { active[i] = 0; --nactive; changed = true; slots_dirty = true; }
        }
        e = hipMemsetAsync(dctl->rot, 0, sizeof(int) * JB_MAX_MATS, s);
        ACKTR_ARG(e == hipSuccess, "sym_eig_jacobi_block: counter reset failed: %s", hipGetErrorString(e));
        if (changed) {
            e = hipMemcpyAsync(dctl->active, active, sizeof(active), hipMemcpyHostToDevice, s);
            ACKTR_ARG(e == hipSuccess, "sym_eig_jacobi_block: control upload failed: %s", hipGetErrorString(e));
            // the host array is reused by later uploads: make sure this one has been consumed
            e = hipStreamSynchronize(s);
            ACKTR_ARG(e == hipSuccess, "sym_eig_jacobi_block: stream sync failed: %s", hipGetErrorString(e));
        }
    }
    for (int i = 0; i < count; ++i) {
        const JbMat& M = host_tab.m[i];
        hipLaunchKernelGGL(jb_lambda_kernel, dim3(cdiv(M.npad, 4)), dim3(256), 0, s, dtab, i);
        hipLaunchKernelGGL(jb_rank_kernel, dim3(cdiv(M.npad, 256)), dim3(256), 0, s, dtab, i);
        hipLaunchKernelGGL(jb_write_kernel, dim3(M.npad), dim3(256), 0, s, dtab, i, thresh);
        if (sweeps_out) sweeps_out[i] = active[i] ? -sweeps[i] : sweeps[i];     // negative: not converged in max_sweeps
    }
    ACKTR_LAUNCH_CHECK();
    // the table upload above reads a thread-local host buffer asynchronously only until the first sync: done
    return 0;
}
