// KFAC (reference a2c_ppo_acktr/algo/kfac.py) on MI355X:
//   * Kronecker-factor statistics (compute_cov_a / compute_cov_g, kfac.py:29-64) as implicit-im2col SYRKs on
//     the fp32 MFMA tile GEMM -- the patch matrix of kfac.py:16-26 is never materialised;
//   * running averages (kfac.py:67-71) as one elementwise pass over the flat factor bucket (after the
//     data-parallel all-reduce of the sample factors);
//   * symmetric eigendecompositions (kfac.py:205-214): hand-written batched parallel-order Jacobi in LDS for
//     small factors (n <= 128, one workgroup per matrix), rocSOLVER syevd for the large ones;
//   * Kronecker-eigenbasis preconditioning (kfac.py:216-227) as four MFMA GEMMs with the damped division
//     fused into the second one's epilogue;
//   * KL clip (kfac.py:229-239) computed on the device (no host sync) feeding the SGD-momentum kernel.
#include <type_traits>
#include <rocsolver/rocsolver.h>

#include <stdlib.h>
#include <thread>

#include "conv_ops.hpp"
#include "kfac_cov_sa.hpp"
#include "gemm_group.hpp"
#include "../../include/acktr_hip.h"

namespace acktr {
int pick_slices(long tiles, long K, long max_by_ws);
int gemm_rowmajor(const float* a, long a_srow, long a_sk, const float* b, long b_srow, long b_sk, const RowMajorEpi& epi,
                  int M, int N, int K, int nslices, hipStream_t s);

// ------------------------------------------------------------------------------------------ factor SYRKs
template <int KH, int KW, int S, int H, int W>
static int cov_a_conv_t(const float* x, const int64_t* idx, int B, int C, float alpha, float* out, float* ws,
                        long ws_floats, hipStream_t s) {
    using Patch = Im2col<KH, KW, S, H, W, false>;
    const int Kc = C * KH * KW, R = B * Patch::SPP;      // reduction over positions, images padded to SPP (zeros)
    Patch pop{x, idx, C, (long)C * H * W, R, Kc};
    // the factor is symmetric: tiles strictly below the diagonal are not computed, their elements arrive as mirror
    // images from the upper triangle (gemm_core.hpp EpiSym) -- T(T+1)/2 of the T^2 tiles do work
    const long tt = cdiv(Kc, 128), tiles = tt * (tt + 1) / 2;
    long max_by_ws = ws_floats / ((long)Kc * Kc) - SLAB_GROUP_EXTRA;
    int ns = pick_slices(tiles, R, max_by_ws < 1 ? 1 : max_by_ws);
    // (384 / 512 / 1024 workgroups instead of pick_slices' 768 measured the same update time within 2 %)
    if (ns > 1 && max_by_ws < ns) ns = 1;
    if (ns == 1) {
        RowMajorEpi e = row_major_epi(out, Kc, Kc, Kc);
        e.scale = alpha;
        e.symmetric = 1;
        int rc1 = launch_gemm<128, 128, 2, 2>(pop, pop, e, Kc, Kc, R, 1, s);
        return rc1 ? rc1 : launch_symmetrize_lower(out, Kc, s);
    }
    RowMajorEpi e = row_major_epi(ws, Kc, Kc, Kc);
    e.symmetric = 1;
    e.slab_stride = (long)Kc * Kc;
    int rc = launch_gemm<128, 128, 2, 2>(pop, pop, e, Kc, Kc, R, ns, s);
    if (rc) return rc;
    rc = launch_slab_reduce(out, ws, ns, (long)Kc * Kc, (long)Kc * Kc, alpha, 0, s);
    return rc ? rc : launch_symmetrize_lower(out, Kc, s);
}

// 8x8 stride-4 layer on 84x84x4 frames: stationary accumulators, one workgroup per CU (kfac_cov_sa.hpp)
template <class XT>
static int cov_a_conv1_sa(const XT* x, const int64_t* idx, int B, float alpha, float* out, float* ws, long ws_floats,
                          hipStream_t s, bool any_batch = false) {
    static const bool off = getenv("ACKTR_COV_A_TILE") != nullptr && !std::is_same<XT, uint8_t>::value;   // tuning switch
    hipDeviceProp_t prop;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -2;
        cus = prop.multiProcessorCount;
    }
    int ns = B * C1A_GROUPS / 10;                       // at least ten row-pair groups per workgroup
    if (ns > cus) ns = cus;
    if (ns < 1) ns = 1;
    if (off || (B < 64 && !any_batch) || ws_floats < (long)(ns + SLAB_GROUP_EXTRA) * C1A_D * C1A_D) return -2;
    hipLaunchKernelGGL(cov_a_conv1_sa_kernel<XT>, dim3(ns), dim3(256), 0, s, x, idx, ws, B);
    ACKTR_LAUNCH_CHECK();
    int rc = launch_slab_reduce(out, ws, ns, (long)C1A_D * C1A_D, (long)C1A_D * C1A_D, alpha, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(cov_a_conv1_mirror_kernel, dim3(C1A_D * C1A_D / 256), dim3(256), 0, s, out);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

// 4x4 stride-2 layer on 20x20x32 and 3x3 stride-1 layer on 9x9x64 activations: stationary accumulators by channel-group
// pair (kfac_cov_sa.hpp).  Workgroups are dealt to the pairs so that the slowest wave (images x tiles) is as short as
// possible.
template <class G>
static int cov_a_sa(const float* x, int B, float alpha, float* out, float* ws, long ws_floats, hipStream_t s) {
    using S = CovSa<G>;
    static const bool off = getenv("ACKTR_COV_A_TILE") != nullptr;          // tuning / comparison switch
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -2;
        cus = prop.multiProcessorCount;
    }
    constexpr int NCROSS = S::PAIRS - G::NG, NDIAG = G::NG;
    if (off || B < 64 || cus < NCROSS + NDIAG) return -2;
    int best_c = 1, best_d = 1;
    long best = -1;
    for (int nc = 1; NCROSS * nc + NDIAG <= cus; ++nc) {
        const int nd = (cus - NCROSS * nc) / NDIAG;
        const long tc = (long)cdiv(B, 4 * nc) * S::NT_CROSS, td = (long)cdiv(B, 4 * nd) * S::NT_DIAG, t = tc > td ? tc : td;
        if (best < 0 || t < best) { best = t; best_c = nc; best_d = nd; }
    }
    CovPlan plan;
    int p = 0, wg = 0;
    for (int a = 0; a < G::NG; ++a)
        for (int b = a; b < G::NG; ++b) {
            plan.ga[p] = a; plan.gb[p] = b; plan.wg0[p] = wg;
            wg += (a == b) ? best_d : best_c;
            ++p;
        }
    for (; p <= COVSA_MAX_PAIRS; ++p) plan.wg0[p] = wg;
    if (ws_floats < (long)wg * S::SLAB) return -2;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cov_a_sa_kernel<G>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, S::LDS_BYTES);
        if (e != hipSuccess) { set_error("cov_a_sa: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL(cov_a_sa_kernel<G>, dim3(wg), dim3(256), S::LDS_BYTES, s, x, ws, B, plan);
    ACKTR_LAUNCH_CHECK();
    hipLaunchKernelGGL(cov_sa_reduce_kernel<G>, dim3(S::NT_CROSS * 4, S::PAIRS), dim3(256), 0, s, (const float*)ws, out, alpha, plan);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

// stride-1 layer without a gather: both operands stream 4 padded positions per 16-byte load (zero patches at the pad)
template <int KH, int KW, int H, int W, int OWP>
static int cov_a_conv_pos_t(const float* x, int B, int C, float alpha, float* out, float* ws, long ws_floats,
                            hipStream_t s) {
    using Patch = Im2colPos<KH, KW, H, W, OWP, true, true>;
    const int Kc = C * KH * KW, RP = B * Patch::PP;
    Patch pop{x, x + (long)B * C * H * W, C, RP, Kc};
    const long tt = cdiv(Kc, 128), tiles = tt * (tt + 1) / 2;
    long max_by_ws = ws_floats / ((long)Kc * Kc) - SLAB_GROUP_EXTRA;
    int ns = pick_slices(tiles, RP, max_by_ws < 1 ? 1 : max_by_ws);
    if (ns > 1 && max_by_ws < ns) ns = 1;
    RowMajorEpi e = row_major_epi(ns == 1 ? out : ws, Kc, Kc, Kc);
    e.symmetric = 1;
    if (ns == 1) e.scale = alpha; else e.slab_stride = (long)Kc * Kc;
    int rc = launch_gemm<128, 128, 2, 2>(pop, pop, e, Kc, Kc, RP, ns, s);
    if (rc) return rc;
    if (ns > 1) rc = launch_slab_reduce(out, ws, ns, (long)Kc * Kc, (long)Kc * Kc, alpha, 0, s);
    return rc ? rc : launch_symmetrize_lower(out, Kc, s);
}

template <int SP>
static int cov_g_conv_t(const float* dy, int B, int OC, float alpha, float* out, float* ws, long ws_floats,
                        hipStream_t s) {
    const int R = B * ConvOutKC<SP>::SPP;               // positions of an image padded to a multiple of 4 (zeros)
    ConvOutKC<SP> gop{dy, OC, R};
    long max_by_ws = ws_floats / ((long)OC * OC) - SLAB_GROUP_EXTRA;
    int ns = pick_slices(1, R, max_by_ws < 1 ? 1 : max_by_ws);
    if (ns > 512) ns = 512;
    if (ns > 1 && max_by_ws < ns) ns = 1;
    RowMajorEpi e = row_major_epi(ns == 1 ? out : ws, OC, OC, OC);
    e.symmetric = 1;
    if (ns == 1) e.scale = alpha; else e.slab_stride = (long)OC * OC;
    int rc;
    if (OC <= 32) rc = launch_gemm<32, 128, 1, 4>(gop, gop, e, OC, OC, R, ns, s);
    else rc = launch_gemm<64, 128, 2, 2>(gop, gop, e, OC, OC, R, ns, s);
    if (rc) return rc;
    if (ns > 1) rc = launch_slab_reduce(out, ws, ns, (long)OC * OC, (long)OC * OC, alpha, 0, s);
    // OC <= the tile height of either form: no tile lies below the diagonal, so none was skipped, and (i, j) / (j, i)
    // are the same products summed in the same order -- the factor is exactly symmetric without a mirroring pass
    if (OC <= (OC <= 32 ? 32 : 64)) return rc;
    return rc ? rc : launch_symmetrize_lower(out, OC, s);
}

// s[b, oc] = sum_sp g[b, oc, sp]     (kfac.py:59-61)
__global__ __launch_bounds__(256) void spatial_sum_kernel(const float* g, int rows, int SP, float* out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);     // one wave per (b, oc) row
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int i = lane; i < SP; i += 64) s += g[(long)r * SP + i];
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
}

// m = ((m * d/(1-d)) + x) * (1-d)   (three roundings, kfac.py:67-71); first step: m := x before the update
__global__ __launch_bounds__(256) void running_stat_kernel(float* m, const float* x, long n, float c1, float c2,
                                                           int init) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float xi = x[i];
        float mi = init ? xi : m[i];
        mi = mi * c1;
        mi = mi + xi;
        mi = mi * c2;
        m[i] = mi;
    }
}

// ------------------------------------------------------------------------------------------ batched Jacobi
// One workgroup per symmetric matrix (n <= 128), A and V resident in LDS (pitch n+1: conflict-free column
// walks).  Parallel-order cyclic Jacobi: each of the m-1 rounds of a sweep applies m/2 disjoint rotations
// (round-robin "circle" pairing), column update, then row update, V accumulates the rotations.
// Output: eigenvalues ascending (thresholded: d *= (d > thresh), kfac.py:213-214), eigenvectors as ROWS of qt.
struct JacobiDesc {
    const float* a;   // n x n input (row major, symmetric)
    float* d;         // n eigenvalues
    float* qt;        // n x n, row j = eigenvector j
    int n;
};

__global__ __launch_bounds__(256) void jacobi_batched_kernel(const JacobiDesc* descs, float thresh, int max_sweeps) {
    extern __shared__ __attribute__((aligned(16))) float jl[];
    __shared__ float cs[64][2];
    __shared__ int pp[64], qq[64];
    __shared__ int any_rot;
    __shared__ float red[16];
    const JacobiDesc de = descs[blockIdx.x];
    const int n = de.n, P = n + 1, tid = threadIdx.x;
    float* A = jl;
    float* V = jl + (long)n * P;
    float fro = 0.f;
    for (int i = tid; i < n * n; i += 256) {
        const int r = i / n, c = i % n;
        const float v = de.a[i];
        A[r * P + c] = v;
        V[r * P + c] = (r == c) ? 1.f : 0.f;
        fro += v * v;
    }
    fro = block_sum(fro, red);
    if (tid == 0) red[0] = fro;
    __syncthreads();
    const float tiny = 1e-7f * sqrtf(red[0]) + 1e-37f;   // ~fp32 eps * ||A||_F: below this an entry is rounding noise
    __syncthreads();
    const int m = (n + 1) & ~1;          // even number of players (one dummy if n is odd)
    const int half = m / 2;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (tid == 0) any_rot = 0;
        __syncthreads();
        for (int round = 0; round < m - 1; ++round) {
            if (tid < half) {
                int p, q;
                if (tid == 0) { p = m - 1; q = round; }
                else { p = (round + tid) % (m - 1); q = (round - tid + (m - 1)) % (m - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                float c = 1.f, s = 0.f;
                if (q < n) {
                    const float apq = A[p * P + q];
                    if (fabsf(apq) > tiny) {
                        const float tau = (A[q * P + q] - A[p * P + p]) / (2.f * apq);
                        const float t = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
                        c = 1.f / sqrtf(1.f + t * t);
                        s = t * c;
                        any_rot = 1;
                    }
                } else {
                    q = p;                                   // dummy partner: identity
                }
                pp[tid] = p; qq[tid] = q; cs[tid][0] = c; cs[tid][1] = s;
            }
            __syncthreads();
            // columns: A <- A J, V <- V J
            for (int w = tid; w < half * n; w += 256) {
                const int k = w / n, i = w % n;
                const int p = pp[k], q = qq[k];
                if (p != q) {
                    const float c = cs[k][0], s = cs[k][1];
                    const float aip = A[i * P + p], aiq = A[i * P + q];
                    A[i * P + p] = c * aip - s * aiq;
                    A[i * P + q] = s * aip + c * aiq;
                    const float vip = V[i * P + p], viq = V[i * P + q];
                    V[i * P + p] = c * vip - s * viq;
                    V[i * P + q] = s * vip + c * viq;
                }
            }
            __syncthreads();
            // rows: A <- J^T A
            for (int w = tid; w < half * n; w += 256) {
                const int k = w / n, j = w % n;
                const int p = pp[k], q = qq[k];
                if (p != q) {
                    const float c = cs[k][0], s = cs[k][1];
                    const float apj = A[p * P + j], aqj = A[q * P + j];
                    A[p * P + j] = c * apj - s * aqj;
                    A[q * P + j] = s * apj + c * aqj;
                }
            }
            __syncthreads();
        }
        if (!any_rot) break;
        __syncthreads();
    }
    // ascending order by rank (stable), threshold, write eigenvectors as rows
    for (int i = tid; i < n; i += 256) {
        const float di = A[i * P + i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float dj = A[j * P + j];
            rank += (dj < di || (dj == di && j < i)) ? 1 : 0;
        }
        de.d[rank] = di > thresh ? di : 0.f;
        V[n * P + i] = (float)rank;       // spare row n of V (the LDS carve has room for n+1 rows)
        float nrm = 0.f;                  // re-normalise eigenvector i: removes the c^2+s^2 rounding drift
        for (int r = 0; r < n; ++r) nrm += V[r * P + i] * V[r * P + i];
        cs[i >> 1][i & 1] = rsqrtf(nrm);
    }
    __syncthreads();
    for (int w = tid; w < n * n; w += 256) {
        const int col = w / n, i = w % n;         // eigenvector = column `col` of V
        const int rank = (int)V[n * P + col];
        de.qt[(long)rank * n + i] = V[i * P + col] * cs[col >> 1][col & 1];
    }
}

__global__ __launch_bounds__(256) void eig_threshold_kernel(float* d, int n, float thresh) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float v = d[i]; d[i] = v > thresh ? v : 0.f; }
}

// ------------------------------------------------------------------------------------------ KL clip
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* a, const float* b, long n, double* partial) {
    __shared__ double sm[16];
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += (double)a[i] * (double)b[i];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// out[0] = vg_sum = lr^2 * dot ; out[1] = nu = min(1, sqrt(kl_clip / vg_sum))   (kfac.py:229-234)
__global__ __launch_bounds__(64) void kl_clip_kernel(const double* partial, int nparts, float lr, float kl_clip,
                                                     float* out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        const float vg = (float)s * lr * lr;
        float nu = sqrtf(kl_clip / vg);
        if (!(nu < 1.f)) nu = 1.f;       // also catches NaN (vg <= 0)
        out[0] = vg;
        out[1] = nu;
    }
}

static rocblas_handle g_handle = nullptr;
static int ensure_handle(hipStream_t s) {
    if (!g_handle) {
        if (rocblas_create_handle(&g_handle) != rocblas_status_success) {
            set_error("rocblas_create_handle failed");
            g_handle = nullptr;
            return -1;
        }
    }
    if (rocblas_set_stream(g_handle, s) != rocblas_status_success) {
        set_error("rocblas_set_stream failed");
        return -1;
    }
    return 0;
}
}  // namespace acktr

using namespace acktr;

extern "C" long acktr_kfac_cov_workspace_bytes(int D, long R) {
    const long tt = cdiv(D, 128), tiles = tt * (tt + 1) / 2;
    int ns = pick_slices(tiles, R, 1024);
    if (ns > 512) ns = 512;
    return (long)(ns + SLAB_GROUP_EXTRA) * D * D * (long)sizeof(float);
}

extern "C" int acktr_kfac_cov_a_conv1_u8(const uint8_t* x, const int64_t* gather_idx, int B, float alpha, float* out,
                                         void* workspace, long workspace_bytes, void* stream) {
    ACKTR_ARG(x && out && workspace, "kfac_cov_a_conv1_u8: null pointer");
    ACKTR_ARG(B > 0 && ((uintptr_t)x & 3) == 0, "kfac_cov_a_conv1_u8: bad batch size or alignment");
    const int rc = cov_a_conv1_sa<uint8_t>(x, gather_idx, B, alpha, out, (float*)workspace,
                                           workspace_bytes / (long)sizeof(float), (hipStream_t)stream, true);
    ACKTR_ARG(rc != -2, "kfac_cov_a_conv1_u8: workspace too small (acktr_kfac_cov_workspace_bytes(256, B * 400))");
    return rc;
}

extern "C" int acktr_kfac_cov_a_conv_f32(const float* x, const int64_t* gather_idx, int B, int C, int H, int W,
                                         int KH, int KW, int S, float alpha, float* out, void* workspace,
                                         long workspace_bytes, void* stream) {
    ACKTR_ARG(x && out, "kfac_cov_a_conv: null pointer");
    ACKTR_ARG(B > 0 && C > 0, "kfac_cov_a_conv: bad sizes");
    ACKTR_ARG(((uintptr_t)x & 15) == 0, "kfac_cov_a_conv: x must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const long wf = workspace ? workspace_bytes / (long)sizeof(float) : 0;
    if (H == 84 && W == 84 && KH == 8 && KW == 8 && S == 4 && C == C1A_C) {
        const int rc = cov_a_conv1_sa(x, gather_idx, B, alpha, out, ws, wf, s);
        if (rc != -2) return rc;
    }
    if (H == 84 && W == 84 && KH == 8 && KW == 8 && S == 4) return cov_a_conv_t<8, 8, 4, 84, 84>(x, gather_idx, B, C, alpha, out, ws, wf, s);
    if (H == 20 && W == 20 && KH == 4 && KW == 4 && S == 2 && C == 32 && !gather_idx && (long)B * CovGeom2::IMG < (1L << 31)) {
        const int rc = cov_a_sa<CovGeom2>(x, B, alpha, out, ws, wf, s);
        if (rc != -2) return rc;
    }
    if (H == 9 && W == 9 && KH == 3 && KW == 3 && S == 1 && C == 64 && !gather_idx && (long)B * CovGeom3::IMG < (1L << 31)) {
        const int rc = cov_a_sa<CovGeom3>(x, B, alpha, out, ws, wf, s);
        if (rc != -2) return rc;
    }
    if (H == 20 && W == 20 && KH == 4 && KW == 4 && S == 2) return cov_a_conv_t<4, 4, 2, 20, 20>(x, gather_idx, B, C, alpha, out, ws, wf, s);
    if (H == 9 && W == 9 && KH == 3 && KW == 3 && S == 1) {
        if (!gather_idx && (long)B * C * 81 < (1L << 31)) return cov_a_conv_pos_t<3, 3, 9, 9, 8>(x, B, C, alpha, out, ws, wf, s);
        return cov_a_conv_t<3, 3, 1, 9, 9>(x, gather_idx, B, C, alpha, out, ws, wf, s);
    }
    set_error("kfac_cov_a_conv: unsupported geometry %dx%d k%dx%d s%d", H, W, KH, KW, S);
    return -1;
}

extern "C" int acktr_kfac_cov_g_conv_f32(const float* dy, int B, int OC, int SP, float alpha, float* out,
                                         void* workspace, long workspace_bytes, void* stream) {
    ACKTR_ARG(dy && out, "kfac_cov_g_conv: null pointer");
    ACKTR_ARG(B > 0 && OC > 0, "kfac_cov_g_conv: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const long wf = workspace ? workspace_bytes / (long)sizeof(float) : 0;
    if (SP == 400) return cov_g_conv_t<400>(dy, B, OC, alpha, out, ws, wf, s);
    if (SP == 81) return cov_g_conv_t<81>(dy, B, OC, alpha, out, ws, wf, s);
    if (SP == 49) return cov_g_conv_t<49>(dy, B, OC, alpha, out, ws, wf, s);
    set_error("kfac_cov_g_conv: unsupported spatial size %d (supported 400, 81, 49)", SP);
    return -1;
}

static int cov_dense_sa(const float* a, long lda, int B, int D, float alpha, float* out, hipStream_t s);

extern "C" int acktr_kfac_cov_dense_f32(const float* a, long lda, int B, int D, float alpha, float* out,
                                        void* workspace, long workspace_bytes, void* stream) {
    ACKTR_ARG(a && out, "kfac_cov_dense: null pointer");
    ACKTR_ARG(B > 0 && D > 0 && lda >= D, "kfac_cov_dense: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    // wide factors (the FC layer's 1568 x 1568 input factor, the 512 x 512 ones): upper-triangle 64 x 64 tiles on
    // stationary accumulators, both operands read straight along the feature index (precond_sa_kernel below) -- the tile
    // GEMM ran the 1568 factor at 22 TFLOP/s (60 us + a slab sum)
    static const bool sa_off = getenv("ACKTR_COV_DENSE_TILE") != nullptr;      // comparison switch
    // (measured at 640 rows: D = 1568 69.7 -> 47 us; D = 512 and 256 are quicker on the tile GEMM: 18.9 / 14.3 against 24.0 / 22.7 us)
    if (!sa_off && D >= 1024 && B >= 64 && lda < (1L << 30)) return cov_dense_sa(a, lda, B, D, alpha, out, s);
    float* ws = (float*)workspace;
    const long wf = workspace ? workspace_bytes / (long)sizeof(float) : 0;
    const long tiles = (long)cdiv(D, 128) * cdiv(D, 128);
    long max_by_ws = wf / ((long)D * D) - SLAB_GROUP_EXTRA;
    int ns = pick_slices(tiles, B, max_by_ws < 1 ? 1 : max_by_ws);
    if (ns > 1 && max_by_ws < ns) ns = 1;
    // out[i, j] = alpha * sum_b a[b, i] a[b, j]  : both operands row contiguous (k = b)
    if (ns == 1) {
        RowMajorEpi e = row_major_epi(out, D, D, D);
        e.scale = alpha;
        e.symmetric = 1;
        int rc1 = gemm_rowmajor(a, 1, lda, a, 1, lda, e, D, D, B, 1, s);
        if (D <= 32) return rc1;           // one 128x32 / 32x128 tile row: nothing skipped, exactly symmetric as computed
        return rc1 ? rc1 : launch_symmetrize_lower(out, D, s);
    }
    RowMajorEpi e = row_major_epi(ws, D, D, D);
    e.symmetric = 1;
    e.slab_stride = (long)D * D;
    int rc = gemm_rowmajor(a, 1, lda, a, 1, lda, e, D, D, B, ns, s);
    if (rc) return rc;
    rc = launch_slab_reduce(out, ws, ns, (long)D * D, (long)D * D, alpha, 0, s);
    if (D <= 32) return rc;
    return rc ? rc : launch_symmetrize_lower(out, D, s);
}

// fast_cnn statistics (kfac.py:33-36): out[b, (c, ky, kx)] = scale * mean over the OH x OW window positions of
// x[img(b), c, S*oy + ky, S*ox + kx] -- one mean patch per image.  One workgroup per (image, channel): the plane goes
// through LDS once, thread (part, tap) sums every `parts`-th window position of its tap, the parts are added in order.
__global__ __launch_bounds__(256) void patch_mean_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx, int C,
                                                         int H, int W, int KH, int KW, int S, float scale,
                                                         float* __restrict__ out) {
    extern __shared__ float pm_smem[];
    const int b = blockIdx.x / C, c = blockIdx.x % C;
    const long img = idx ? (long)idx[b] : (long)b;
    const float* plane = x + (img * C + c) * (long)H * W;
    const int OH = (H - KH) / S + 1, OW = (W - KW) / S + 1, taps = KH * KW, parts = 256 / taps;
    float* part_sum = pm_smem + H * W;
    for (int i = threadIdx.x; i < H * W; i += 256) pm_smem[i] = plane[i];
    __syncthreads();
    const int tap = threadIdx.x % taps, part = threadIdx.x / taps;
    if (part < parts) {
        const int ky = tap / KW, kx = tap % KW;
        float s = 0.f;
        for (int p = part; p < OH * OW; p += parts) s += pm_smem[((p / OW) * S + ky) * W + (p % OW) * S + kx];
        part_sum[part * taps + tap] = s;
    }
    __syncthreads();
    if (threadIdx.x < taps) {
        float s = 0.f;
        for (int q = 0; q < parts; ++q) s += part_sum[q * taps + threadIdx.x];
        out[((long)b * C + c) * taps + threadIdx.x] = scale * (s / (float)(OH * OW));
    }
}

extern "C" int acktr_kfac_patch_mean_f32(const float* x, const int64_t* gather_idx, int B, int C, int H, int W, int KH, int KW,
                                         int S, float scale, float* out, void* stream) {
    ACKTR_ARG(x && out && B > 0 && C > 0, "kfac_patch_mean: bad arguments");
    ACKTR_ARG(KH > 0 && KW > 0 && S > 0 && KH <= H && KW <= W && KH * KW <= 256 && (long)H * W <= 8192,
              "kfac_patch_mean: unsupported geometry %dx%d k%dx%d s%d", H, W, KH, KW, S);
    ACKTR_ARG((long)B * C < (1L << 31), "kfac_patch_mean: too many planes");
    const size_t lds = ((size_t)H * W + 256) * sizeof(float);
    hipLaunchKernelGGL(patch_mean_kernel, dim3(B * C), dim3(256), lds, (hipStream_t)stream, x, gather_idx, C, H, W, KH, KW, S,
                       scale, out);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_spatial_sum_f32(const float* g, int rows, int SP, float* out, void* stream) {
    ACKTR_ARG(g && out && rows > 0 && SP > 0, "spatial_sum: bad arguments");
    hipLaunchKernelGGL(spatial_sum_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, g, rows, SP, out);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_kfac_running_stat_f32(float* m, const float* x, long n, float c1, float c2, int init,
                                           void* stream) {
    ACKTR_ARG(m && x && n > 0, "kfac_running_stat: bad arguments");
    int nb = cdiv(n, 1024);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(running_stat_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, m, x, n, c1, c2, init);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_jacobi_max_n(void) { return 128; }

extern "C" int acktr_sym_eig_jacobi_batched_f32(const void* descs_dev, int count, int max_n, float thresh,
                                                void* stream) {
    ACKTR_ARG(descs_dev && count > 0, "sym_eig_jacobi: bad arguments");
    ACKTR_ARG(max_n >= 1 && max_n <= 128, "sym_eig_jacobi: n=%d exceeds the LDS-resident limit of 128", max_n);
    const int smem = 2 * (max_n + 1) * (max_n + 1) * (int)sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_batched_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 129 * 129 * 4);
        if (e != hipSuccess) { set_error("jacobi: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL(jacobi_batched_kernel, dim3(count), dim3(256), smem, (hipStream_t)stream,
                       (const JacobiDesc*)descs_dev, thresh, 30);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_sym_eig_rocsolver_f32(float* a_inout_qt, int n, float* d, float* e_work, int* info_dev,
                                           float thresh, void* stream) {
    ACKTR_ARG(a_inout_qt && d && e_work && info_dev && n > 0, "sym_eig_rocsolver: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_handle(s);
    if (rc) return rc;
    // column-major eigenvectors as columns == row-major rows: a_inout_qt[j*n + i] = component i of eigenvector j
    rocblas_status st = rocsolver_ssyevd(g_handle, rocblas_evect_original, rocblas_fill_upper, n, a_inout_qt, n, d,
                                         e_work, info_dev);
    if (st != rocblas_status_success) {
        set_error("rocsolver_ssyevd(n=%d) failed with status %d", n, (int)st);
        return -1;
    }
    hipLaunchKernelGGL(eig_threshold_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, d, n, thresh);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

// Several syevd problems at once.  Each solve is a long chain of small, latency-bound kernels (31 ms at n = 1569,
// 11 ms at n = 513 on MI355X) that leaves the chip nearly idle, so the solves are spread over side streams
// (longest first) that fork from the caller's stream with an event; every worker thread drains its own stream before
// it returns, so the join is on the host.  (An event join back into the caller's stream was measured to deadlock
// against a collective library's own record-on-current-stream / wait-on-pool-stream pattern when the caller's
// stream is the null stream; the host join costs nothing here because syevd's host driver blocks anyway.)
namespace acktr {
constexpr int EIG_STREAMS = 8;          // upper bound; ACKTR_EIG_STREAMS (default 8) picks how many are used:
                                        // NatureCNN's 8 large factors took 55 / 45.5 / 41.5 ms on 2 / 4 / 7 streams
                                        // (stream priorities -- longest solve urgent, the rest low -- made it 204 ms)
static int g_eig_nstreams = 0;
struct EigSide { hipStream_t s; rocblas_handle h; };
static EigSide g_side[EIG_STREAMS];
static hipEvent_t g_fork = nullptr;
static int g_side_dev = -1;
static int ensure_side_streams() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("hipGetDevice failed"); return -1; }
    if (g_side_dev == dev) return 0;
    if (g_side_dev != -1) { set_error("sym_eig_rocsolver_multi: one device per process (first used on %d)", g_side_dev); return -1; }
    {
        const char* ev = getenv("ACKTR_EIG_STREAMS");
        int n = ev ? atoi(ev) : EIG_STREAMS;
        g_eig_nstreams = n < 1 ? 1 : (n > EIG_STREAMS ? EIG_STREAMS : n);
    }
    if (hipEventCreateWithFlags(&g_fork, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return -1; }
    for (int i = 0; i < EIG_STREAMS; ++i) {
        if (hipStreamCreateWithFlags(&g_side[i].s, hipStreamNonBlocking) != hipSuccess ||
            rocblas_create_handle(&g_side[i].h) != rocblas_status_success ||
            rocblas_set_stream(g_side[i].h, g_side[i].s) != rocblas_status_success) {
            set_error("sym_eig_rocsolver_multi: could not create side stream %d", i);
            return -1;
        }
    }
    g_side_dev = dev;
    return 0;
}
}  // namespace acktr

extern "C" int acktr_sym_eig_rocsolver_multi_f32(const acktr_eig_job* jobs, int count, float* e_work, int* info_dev,
                                                 float thresh, void* stream) {
    ACKTR_ARG(jobs && e_work && info_dev && count >= 0, "sym_eig_rocsolver_multi: bad arguments");
    if (count == 0) return 0;
    ACKTR_ARG(count <= 64, "sym_eig_rocsolver_multi: at most 64 problems per call (got %d)", count);
    hipStream_t main_s = (hipStream_t)stream;
    if (ensure_side_streams()) return -1;
    int order[64];
    long eoff[64];
    long off = 0;
    for (int i = 0; i < count; ++i) {
        ACKTR_ARG(jobs[i].a_inout_qt && jobs[i].d && jobs[i].n > 0, "sym_eig_rocsolver_multi: bad job %d", i);
        order[i] = i; eoff[i] = off; off += jobs[i].n;
    }
    for (int i = 1; i < count; ++i)                    // longest first, then greedy onto the least loaded stream
        for (int j = i; j > 0 && jobs[order[j]].n > jobs[order[j - 1]].n; --j) { int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    double load[EIG_STREAMS] = {0};
    int assign[64];
    bool used[EIG_STREAMS] = {false};
    for (int k = 0; k < count; ++k) {
        const acktr_eig_job& jb = jobs[order[k]];
        int si = 0;
        for (int i = 1; i < g_eig_nstreams; ++i) if (load[i] < load[si]) si = i;
        load[si] += (double)jb.n * jb.n * jb.n + 4e6 * jb.n;          // ~ measured cost shape (latency term dominates)
        assign[k] = si;
        used[si] = true;
    }
    if (hipEventRecord(g_fork, main_s) != hipSuccess) { set_error("hipEventRecord failed"); return -1; }
    for (int i = 0; i < EIG_STREAMS; ++i)
        if (used[i] && hipStreamWaitEvent(g_side[i].s, g_fork, 0) != hipSuccess) { set_error("hipStreamWaitEvent failed"); return -1; }
    // one host thread per side stream: syevd's host driver blocks on the device between its phases, so calls issued
    // from a single thread would still run one after the other
    int dev = 0;
    (void)hipGetDevice(&dev);
    int status[EIG_STREAMS] = {0};
    int failed_n[EIG_STREAMS] = {0};
    std::thread workers[EIG_STREAMS];
    for (int i = 0; i < EIG_STREAMS; ++i) {
        if (!used[i]) continue;
        workers[i] = std::thread([&, i]() {
            if (hipSetDevice(dev) != hipSuccess) { status[i] = -2; return; }
            EigSide& sd = g_side[i];
            for (int k = 0; k < count; ++k) {
                if (assign[k] != i) continue;
                const acktr_eig_job& jb = jobs[order[k]];
                rocblas_status st = rocsolver_ssyevd(sd.h, rocblas_evect_original, rocblas_fill_upper, jb.n, jb.a_inout_qt,
                                                     jb.n, jb.d, e_work + eoff[order[k]], info_dev + order[k]);
                if (st != rocblas_status_success) { status[i] = (int)st; failed_n[i] = jb.n; return; }
                hipLaunchKernelGGL(eig_threshold_kernel, dim3(cdiv(jb.n, 256)), dim3(256), 0, sd.s, jb.d, jb.n, thresh);
                if (hipGetLastError() != hipSuccess) { status[i] = -3; failed_n[i] = jb.n; return; }
            }
            if (hipStreamSynchronize(sd.s) != hipSuccess) status[i] = -4;
        });
    }
    for (int i = 0; i < EIG_STREAMS; ++i) if (used[i]) workers[i].join();
    for (int i = 0; i < EIG_STREAMS; ++i) {
        if (used[i] && status[i] != 0) {
            set_error("sym_eig_rocsolver_multi: solve (n=%d) on side stream %d failed with status %d", failed_n[i], i, status[i]);
            return -1;
        }
    }
    return 0;
}

extern "C" int acktr_kfac_precondition_f32(const float* grad, const float* qt_g, const float* d_g, const float* qt_a,
                                           const float* d_a, int OUT, int IN, float damping, float* tmp1, float* tmp2,
                                           float* v_out, void* stream) {
    ACKTR_ARG(grad && qt_g && d_g && qt_a && d_a && tmp1 && tmp2 && v_out, "kfac_precondition: null pointer");
    ACKTR_ARG(OUT > 0 && IN > 0, "kfac_precondition: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // T1 = Qg^T G      (Qg^T = qt_g)           [OUT x IN] : A(m,k) = qt_g[m,k], B(n,k) = grad[k,n]
    RowMajorEpi e1 = row_major_epi(tmp1, OUT, IN, IN);
    rc = gemm_rowmajor(qt_g, OUT, 1, grad, 1, IN, e1, OUT, IN, OUT, 1, s);
    if (rc) return rc;
    // T2 = (T1 Qa) / (d_g d_a^T + lambda)      : B(n,k) = Qa[k,n] = qt_a[n,k]
    RowMajorEpi e2 = row_major_epi(tmp2, OUT, IN, IN);
    e2.div_m = d_g; e2.div_n = d_a; e2.div_add = damping;
    rc = gemm_rowmajor(tmp1, IN, 1, qt_a, IN, 1, e2, OUT, IN, IN, 1, s);
    if (rc) return rc;
    // T3 = Qg T2       : A(m,k) = Qg[m,k] = qt_g[k,m], B(n,k) = T2[k,n]
    RowMajorEpi e3 = row_major_epi(tmp1, OUT, IN, IN);
    rc = gemm_rowmajor(qt_g, 1, OUT, tmp2, 1, IN, e3, OUT, IN, OUT, 1, s);
    if (rc) return rc;
    // V = T3 Qa^T      : B(n,k) = Qa^T[k,n] = Qa[n,k] = qt_a[k,n]
    RowMajorEpi e4 = row_major_epi(v_out, OUT, IN, IN);
    return gemm_rowmajor(tmp1, IN, 1, qt_a, 1, IN, e4, OUT, IN, IN, 1, s);
}

// All layers at once: stage s of every layer's four-product chain is one grouped launch (gemm_group.hpp); 16-byte or
// scalar operand loads are chosen per operand and per layer inside the kernel.
namespace acktr {
static int precondition_group(const acktr_precond_job* jb, int n, float damping, hipStream_t s) {
    GroupDense<true> kc_a, kc_b;
    GroupDense<false> rc_a, rc_b;
    GroupEpi e;
    int maxOUT = 0, maxIN = 0;
    for (int i = 0; i < n; ++i) {
        if (jb[i].OUT > maxOUT) maxOUT = jb[i].OUT;
        if (jb[i].IN > maxIN) maxIN = jb[i].IN;
    }
    int rc;
    // T1 = Qg^T G : A(m,k) = qt_g[m*OUT + k], B(n,k) = grad[k*IN + n]
    for (int i = 0; i < n; ++i) {
        const acktr_precond_job& J = jb[i];
        kc_a.j[i] = group_op(J.qt_g, J.OUT, J.OUT, J.OUT, true);
        rc_b.j[i] = group_op(J.grad, J.IN, J.OUT, J.IN, false);
        e.j[i] = GroupEpiJob{J.tmp1, J.OUT, J.IN, (long)J.IN, J.d_g, J.d_a};
    }
    e.div_add = damping; e.use_div = 0;
    rc = launch_gemm_z<64, 64, 2, 2>(kc_a, rc_b, e, maxOUT, maxIN, maxOUT, n, s);
    if (rc) return rc;
    // T2 = (T1 Qa) / (dg da^T + lambda) : A(m,k) = tmp1[m*IN + k], B(n,k) = qt_a[n*IN + k]
    for (int i = 0; i < n; ++i) {
        const acktr_precond_job& J = jb[i];
        kc_a.j[i] = group_op(J.tmp1, J.OUT, J.IN, J.IN, true);
        kc_b.j[i] = group_op(J.qt_a, J.IN, J.IN, J.IN, true);
        e.j[i].c = J.tmp2;
    }
    e.use_div = 1;
    rc = launch_gemm_z<64, 64, 2, 2>(kc_a, kc_b, e, maxOUT, maxIN, maxIN, n, s);
    if (rc) return rc;
    // T3 = Qg T2 : A(m,k) = qt_g[k*OUT + m], B(n,k) = tmp2[k*IN + n]
    for (int i = 0; i < n; ++i) {
        const acktr_precond_job& J = jb[i];
        rc_a.j[i] = group_op(J.qt_g, J.OUT, J.OUT, J.OUT, false);
        rc_b.j[i] = group_op(J.tmp2, J.IN, J.OUT, J.IN, false);
        e.j[i].c = J.tmp1;
    }
    e.use_div = 0;
    rc = launch_gemm_z<64, 64, 2, 2>(rc_a, rc_b, e, maxOUT, maxIN, maxOUT, n, s);
    if (rc) return rc;
    // V = T3 Qa^T : A(m,k) = tmp1[m*IN + k], B(n,k) = qt_a[k*IN + n]
    for (int i = 0; i < n; ++i) {
        const acktr_precond_job& J = jb[i];
        kc_a.j[i] = group_op(J.tmp1, J.OUT, J.IN, J.IN, true);
        rc_b.j[i] = group_op(J.qt_a, J.IN, J.IN, J.IN, false);
        e.j[i].c = J.v_out;
    }
    return launch_gemm_z<64, 64, 2, 2>(kc_a, rc_b, e, maxOUT, maxIN, maxIN, n, s);
}
}  // namespace acktr

// ---------------------------------------------------------------------------------------------------------------------
// The same four stages on stationary accumulators (cf. fc_wgrad_sa_kernel, linear.hip).  With BOTH the eigenvector matrix
// Q and its transpose at hand (acktr_precond_job.q_g / q_a), and with stages 1 and 3 storing their result transposed,
// every operand of every stage is contiguous along its OUTPUT index:
//   T1^T[n,m] = sum_k Qg[k,m] G[k,n]        T2[m,n] = (sum_k T1^T[k,m] Qa[k,n]) / (dg[m] da[n] + lambda)
//   T3^T[n,m] = sum_k Qg^T[k,m] T2[k,n]     V[m,n]  = sum_k T3^T[k,m] Qa^T[k,n]
// so MFMA operands are coalesced loads straight into registers; one workgroup per 64x64 tile of a layer's result, its
// eight waves split K and meet in LDS in wave order (deterministic).  All layers of a stage share one launch (a workgroup
// finds its layer by the tile prefix in the kernel arguments).  The tile GEMM ran the 512 x 1568 layer's 200 tiles at one
// 4-wave workgroup per CU over a K of 1568: 28-43 TFLOP/s.
// ---------------------------------------------------------------------------------------------------------------------
namespace acktr {
constexpr int PS_WAVES = 8, PS_PITCH = 65, PS_LDS_BYTES = PS_WAVES * 64 * PS_PITCH * 4;
struct PsJob {
    const float* a; const float* b; float* c; const float* div_m; const float* div_n;
    int M, N, K, lda, ldb, ldc, tile0, tiles_n, v2, transposed;
    int upper;        // symmetric product (a == b, M == N): only the tiles on and above the diagonal, row by row
    float scale;
};
struct PsJobs { PsJob j[GROUP_MAX]; int count; float div_add; };

// (V2 as a template parameter, chosen once per workgroup at the top of the kernel: with the flag tested inside the K loop's
//  load block the compiler could not interleave loads and MFMAs across the branch -- the 2.5 GFLOP stages of the 512 x 1568
//  layer ran 66 us here against 43 us for the same product in fc_wgrad_sa_kernel)
template <bool V2>
__device__ __forceinline__ void ps_tile(const PsJob& J, const float div_add, const int tm, const int tn, float* ps_smem) {
    typedef float ps2 __attribute__((ext_vector_type(2)));
    typedef ps2 ps2u __attribute__((aligned(8)));
    const int m0 = tm * 64, n0 = tn * 64;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, li = lane & 31, kh = lane >> 5;
    const int M = J.M, N = J.N, K = J.K, lda = J.lda, ldb = J.ldb;
    const int per = ((K + 2 * PS_WAVES - 1) / (2 * PS_WAVES)) * 2;
    const int k0 = wave * per, k1 = min(K, k0 + per);
    const int mc0 = V2 ? m0 + 2 * li : m0 + li, mc1 = V2 ? mc0 + 1 : mc0 + 32;
    const int nc0 = V2 ? n0 + 2 * li : n0 + li, nc1 = V2 ? nc0 + 1 : nc0 + 32;
    const bool mok0 = mc0 < M, mok1 = mc1 < M, nok0 = nc0 < N, nok1 = nc1 < N;
    const float* ap = J.a + mc0;
    const float* bp = J.b + nc0;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    constexpr int U = 2;                                                       // (cf. fc_wgrad_sa_kernel: fine interleave beats deep prefetch)
    float av[2][U][2], bv[2][U][2];
    auto load_blk = [&](int buf, int k) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k + 2 * u + kh;
            const bool ok = kk < k1;
            const long ra = (long)(ok ? kk : 0) * lda, rb = (long)(ok ? kk : 0) * ldb;
            if constexpr (V2) {                                                // (pairs are whole: M, N, m0, n0 even)
                const ps2 a2 = (ok && mok0) ? *reinterpret_cast<const ps2u*>(ap + ra) : ps2{0.f, 0.f};
                const ps2 b2 = (ok && nok0) ? *reinterpret_cast<const ps2u*>(bp + rb) : ps2{0.f, 0.f};
                av[buf][u][0] = a2[0]; av[buf][u][1] = a2[1];
                bv[buf][u][0] = b2[0]; bv[buf][u][1] = b2[1];
            } else {
                av[buf][u][0] = (ok && mok0) ? ap[ra] : 0.f;
                av[buf][u][1] = (ok && mok1) ? ap[ra + 32] : 0.f;
                bv[buf][u][0] = (ok && nok0) ? bp[rb] : 0.f;
                bv[buf][u][1] = (ok && nok1) ? bp[rb + 32] : 0.f;
            }
        }
    };
    auto mul_blk = [&](int buf) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u][a], bv[buf][u][b], acc[a][b], 0, 0, 0);
    };
    if (k0 < k1) {
        load_blk(0, k0);
        for (int k = k0; k < k1; k += 4 * U) {
            load_blk(1, k + 2 * U);
            __builtin_amdgcn_sched_barrier(0);
            mul_blk(0);
            __builtin_amdgcn_sched_barrier(0);
            load_blk(0, k + 4 * U);
            __builtin_amdgcn_sched_barrier(0);
            mul_blk(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* mine = ps_smem + (long)wave * (64 * PS_PITCH);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) mine[((a * 2 + b) * 16 + v) * PS_PITCH + lane] = acc[a][b][v];
    __syncthreads();
    // 4096 results over 512 threads, consecutive threads along the destination's contiguous index
    const bool tr = J.transposed != 0;
    for (int e = threadIdx.x; e < 4096; e += 64 * PS_WAVES) {
        const int ml = tr ? (e & 63) : (e >> 6), nl = tr ? (e >> 6) : (e & 63);
        const int a = V2 ? (ml & 1) : (ml >> 5), i = V2 ? (ml >> 1) : (ml & 31);
        const int b = V2 ? (nl & 1) : (nl >> 5), jn = V2 ? (nl >> 1) : (nl & 31);
        const int v = (i & 3) | ((i >> 3) << 2), l = (((i >> 2) & 1) << 5) | jn;
        const int off = ((a * 2 + b) * 16 + v) * PS_PITCH + l;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < PS_WAVES; ++w) sum += ps_smem[w * (64 * PS_PITCH) + off];
        const int m = m0 + ml, n = n0 + nl;
        if (m < M && n < N) {
            if (J.div_m) sum = sum / (J.div_m[m] * J.div_n[n] + div_add);
            else sum = sum * J.scale;
            if (tr) J.c[(long)n * J.ldc + m] = sum; else J.c[(long)m * J.ldc + n] = sum;
        }
    }
    // symmetric product: a tile above the diagonal is also its mirror image below it -- the same sums (same order, so
    // bit-identical), stored along the other index so that these stores are coalesced too; no separate mirror launch
    if (J.upper && tm != tn) {
        for (int e = threadIdx.x; e < 4096; e += 64 * PS_WAVES) {
            const int ml = e & 63, nl = e >> 6;
            const int a = V2 ? (ml & 1) : (ml >> 5), i = V2 ? (ml >> 1) : (ml & 31);
            const int b = V2 ? (nl & 1) : (nl >> 5), jn = V2 ? (nl >> 1) : (nl & 31);
            const int v = (i & 3) | ((i >> 3) << 2), l = (((i >> 2) & 1) << 5) | jn;
            const int off = ((a * 2 + b) * 16 + v) * PS_PITCH + l;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < PS_WAVES; ++w) sum += ps_smem[w * (64 * PS_PITCH) + off];
            const int m = m0 + ml, n = n0 + nl;
            if (m < M && n < N) J.c[(long)n * J.ldc + m] = sum * J.scale;
        }
    }
}

__global__ __launch_bounds__(64 * PS_WAVES) void precond_sa_kernel(PsJobs jobs) {
    extern __shared__ __attribute__((aligned(16))) float ps_smem[];           // [wave][blk 4][reg 16][PS_PITCH]
    int z = 0;
    while (z + 1 < jobs.count && (int)blockIdx.x >= jobs.j[z + 1].tile0) ++z;  // (uniform)
    const PsJob& J = jobs.j[z];
    // (plain tile order: dealing a layer's column tiles to the XCDs as fc_wgrad_sa_kernel does measured SLOWER here --
    //  45 / 100 us against 30 / 77 us for the 0.8 / 2.5 GFLOP stages of the 512 x 1568 layer)
    int t = blockIdx.x - J.tile0;
    int tm = t / J.tiles_n, tn = t % J.tiles_n;
    if (J.upper) {                                                             // (uniform) row tm holds tiles tm .. tiles_n - 1
        tm = 0;
        while (t >= J.tiles_n - tm) { t -= J.tiles_n - tm; ++tm; }
        tn = tm + t;
    }
    if (J.v2) ps_tile<true>(J, jobs.div_add, tm, tn, ps_smem);
    else ps_tile<false>(J, jobs.div_add, tm, tn, ps_smem);
}

static int precondition_sa_group(const acktr_precond_job* jb, int n, float damping, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(precond_sa_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS_BYTES);
        if (e != hipSuccess) { set_error("precondition_sa: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    for (int stage = 0; stage < 4; ++stage) {
        PsJobs jobs;
        jobs.count = n; jobs.div_add = damping;
        int tiles = 0;
        for (int i = 0; i < n; ++i) {
            const acktr_precond_job& J = jb[i];
            PsJob& P = jobs.j[i];
            P.M = J.OUT; P.N = J.IN; P.div_m = nullptr; P.div_n = nullptr; P.upper = 0; P.scale = 1.f;
            switch (stage) {
                case 0: P.a = J.q_g; P.lda = J.OUT; P.b = J.grad; P.ldb = J.IN; P.K = J.OUT; P.c = J.tmp1; P.transposed = 1; break;
                case 1: P.a = J.tmp1; P.lda = J.OUT; P.b = J.q_a; P.ldb = J.IN; P.K = J.IN; P.c = J.tmp2; P.transposed = 0;
                        P.div_m = J.d_g; P.div_n = J.d_a; break;
                case 2: P.a = J.qt_g; P.lda = J.OUT; P.b = J.tmp2; P.ldb = J.IN; P.K = J.OUT; P.c = J.tmp1; P.transposed = 1; break;
                default: P.a = J.tmp1; P.lda = J.OUT; P.b = J.qt_a; P.ldb = J.IN; P.K = J.IN; P.c = J.v_out; P.transposed = 0; break;
            }
            P.ldc = P.transposed ? J.OUT : J.IN;
            P.v2 = (J.OUT % 2 == 0 && J.IN % 2 == 0 && ((((uintptr_t)P.a) | ((uintptr_t)P.b)) & 7) == 0) ? 1 : 0;
            P.tiles_n = cdiv(J.IN, 64);
            P.tile0 = tiles;
            tiles += cdiv(J.OUT, 64) * P.tiles_n;
        }
        hipLaunchKernelGGL(precond_sa_kernel, dim3(tiles), dim3(64 * PS_WAVES), PS_LDS_BYTES, s, jobs);
        ACKTR_LAUNCH_CHECK();
    }
    return 0;
}
}  // namespace acktr

// out[i, j] = alpha * sum_b a[b, i] a[b, j] (kfac.py:38-46 for a Linear layer's input / grad-output rows) on the same kernel
static int cov_dense_sa(const float* a, long lda, int B, int D, float alpha, float* out, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(acktr::precond_sa_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, acktr::PS_LDS_BYTES);
        if (e != hipSuccess) { set_error("cov_dense_sa: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    acktr::PsJobs jobs;
    jobs.count = 1; jobs.div_add = 0.f;
    acktr::PsJob& P = jobs.j[0];
    P.a = a; P.b = a; P.c = out; P.div_m = nullptr; P.div_n = nullptr;
    P.M = D; P.N = D; P.K = B; P.lda = (int)lda; P.ldb = (int)lda; P.ldc = D; P.tile0 = 0; P.tiles_n = cdiv(D, 64);
    P.v2 = (D % 2 == 0 && lda % 2 == 0 && (((uintptr_t)a) & 7) == 0) ? 1 : 0;
    P.transposed = 0; P.upper = 1; P.scale = alpha;
    const int tiles = P.tiles_n * (P.tiles_n + 1) / 2;
    hipLaunchKernelGGL(acktr::precond_sa_kernel, dim3(tiles), dim3(64 * acktr::PS_WAVES), acktr::PS_LDS_BYTES, s, jobs);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_kfac_precondition_batched_f32(const acktr_precond_job* jobs, int count, float damping, void* stream) {
    ACKTR_ARG(jobs && count > 0, "kfac_precondition_batched: no jobs");
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < count; ++i) {
        const acktr_precond_job& J = jobs[i];
        ACKTR_ARG(J.grad && J.qt_g && J.d_g && J.qt_a && J.d_a && J.tmp1 && J.tmp2 && J.v_out && J.OUT > 0 && J.IN > 0,
                  "kfac_precondition_batched: job %d has a null pointer or a bad size", i);
    }
    static const bool tile_only = getenv("ACKTR_PRECOND_TILE") != nullptr;      // tuning / comparison switch
    bool have_q = !tile_only;
    for (int i = 0; i < count; ++i) have_q = have_q && jobs[i].q_g && jobs[i].q_a;
    for (int i0 = 0; i0 < count; i0 += GROUP_MAX) {
        const int n = count - i0 < GROUP_MAX ? count - i0 : GROUP_MAX;
        const int rc = have_q ? precondition_sa_group(jobs + i0, n, damping, s) : precondition_group(jobs + i0, n, damping, s);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int acktr_kfac_kl_clip_f32(const float* v, const float* g, long n, float lr, float kl_clip,
                                      float* out_vgsum_nu, void* workspace, void* stream) {
    ACKTR_ARG(v && g && out_vgsum_nu && workspace && n > 0, "kfac_kl_clip: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int nb = cdiv(n, 1024);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(256), 0, s, v, g, n, (double*)workspace);
    ACKTR_LAUNCH_CHECK();
    hipLaunchKernelGGL(kl_clip_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, nb, lr, kl_clip, out_vgsum_nu);
    ACKTR_LAUNCH_CHECK();
    return 0;
}
