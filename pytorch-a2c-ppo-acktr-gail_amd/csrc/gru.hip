// Element-wise links of the recurrent (GRU) bases -- reference a2c_ppo_acktr/model.py:89-95,111-166 (`_forward_gru`
// over nn.GRU; gate order r, z, n; ATen's cell forms the new state as n + z*(h - n)).
//
// The two matrix products of a step, gi = x W_ih^T + b_ih (for ALL steps in one GEMM, the inputs do not depend on the
// recurrence) and gh = (h*mask) W_hh^T + b_hh (per step), run on the MFMA / skinny kernels of linear.hip; what is
// left of a step is lane-per-element work on [N, H]:
//   forward :  r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r*gh_n), h = n + z*(hm - n),
//              and the masked state hm' = h * mask[t+1] that the next step consumes            (model.py:113,153-155)
//   backward:  dh = dout[t] + carry*mask[t+1]; the gate derivatives d gi, d gh (written over the saved r,z,n and gh,
//              so BPTT needs no memory beyond what the forward kept) and carry = dh*z, onto which the caller's GEMM
//              accumulates d gh W_hh.
#include "common.hpp"
#include "../../include/acktr_hip.h"

#pragma clang fp contract(off)

namespace acktr {

__device__ __forceinline__ float gru_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void gru_mask_rows_kernel(const float* __restrict__ h, const float* __restrict__ m,
                                                            float* __restrict__ out, int N, int H) {
    const long n = (long)N * H;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        out[i] = h[i] * m[i / H];
}

__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(float* __restrict__ gi_rzn, const float* __restrict__ gh,
                                                            const float* __restrict__ hm, const float* __restrict__ m_next,
                                                            float* __restrict__ h_out, float* __restrict__ hm_next,
                                                            int N, int H) {
    const long n = (long)N * H;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long row = i / H;
        const int j = (int)(i - row * H);
        const long g = row * 3 * H + j;
        const float r = gru_sigmoid(gi_rzn[g] + gh[g]);
        const float z = gru_sigmoid(gi_rzn[g + H] + gh[g + H]);
        const float nn = tanhf(gi_rzn[g + 2 * H] + r * gh[g + 2 * H]);
        const float hp = hm[i];
        const float h = nn + z * (hp - nn);
        gi_rzn[g] = r;
        gi_rzn[g + H] = z;
        gi_rzn[g + 2 * H] = nn;
        h_out[i] = h;
        if (hm_next) hm_next[i] = h * m_next[row];
    }
}

__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ m_next,
                                                            float* __restrict__ rzn_dgi, float* __restrict__ gh_dgh,
                                                            const float* __restrict__ hm, float* __restrict__ carry,
                                                            int last, int N, int H) {
    const long n = (long)N * H;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long row = i / H;
        const int j = (int)(i - row * H);
        const long g = row * 3 * H + j;
        float dh = dout ? dout[i] : 0.f;
        if (!last) dh += carry[i] * m_next[row];
        const float r = rzn_dgi[g], z = rzn_dgi[g + H], nn = rzn_dgi[g + 2 * H];
        const float ghn = gh_dgh[g + 2 * H];
        const float dn_pre = (dh * (1.f - z)) * (1.f - nn * nn);
        const float dz_pre = (dh * (hm[i] - nn)) * (z * (1.f - z));
        const float dr_pre = (dn_pre * ghn) * (r * (1.f - r));
        rzn_dgi[g] = dr_pre;
        rzn_dgi[g + H] = dz_pre;
        rzn_dgi[g + 2 * H] = dn_pre;
        gh_dgh[g] = dr_pre;
        gh_dgh[g + H] = dz_pre;
        gh_dgh[g + 2 * H] = dn_pre * r;
        carry[i] = dh * z;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One launch per time step (few rows: a PPO minibatch of whole trajectories has N // num_mini_batch environments, so
// the per-step product (h*mask) W_hh^T is 8 x 512 x 1536 -- pure launch latency as a GEMM plus a gate kernel).  A wave
// owns ONE hidden unit j: it streams the three gate rows of W_hh (16-byte loads, lanes stride k), dots them with up to
// GS_NR state rows at a time, butterfly-reduces, and the lanes 0..GS_NR-1 finish the gates of their row -- the step's
// only grid-wide dependency (every unit of step t+1 needs every unit of step t) is the launch boundary.
// The backward step does the same for  carry[n, i] += sum_k d gh[t+1][n, k] W_hh[k, i]  (rows of the TRANSPOSED weight,
// written once per pass by acktr_transpose_f32) followed by the gate derivatives of step t for unit i.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int GS_NR = 8;
typedef float g4 __attribute__((ext_vector_type(4)));
template <int CTRL>
__device__ __forceinline__ float gs_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// Sums over the 64 lanes of 8 x G values at once; lane l ends up with the G sums of ROW l % 8 (same bits in the eight
// lanes that share l % 8).  Three halving levels (partners l^1, l^2, l^4: a lane keeps the rows whose index bit matches
// its own lane bit and hands the others over) leave one row per lane, three more levels (l^8, l^16, l^32) add the eight
// lane groups: 7G + 3G exchanges instead of 8G full butterflies (24 values: the sums were 0.7 us of a 6.5 us step).
template <int G>
__device__ __forceinline__ void gs_reduce8(const float (&v)[GS_NR][G], float (&out)[G], int lane) {
    static_assert(GS_NR == 8, "three halving levels");
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float a[4][G], b[2][G];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float keep = b0 ? v[2 * i + 1][g] : v[2 * i][g], send = b0 ? v[2 * i][g] : v[2 * i + 1][g];
            a[i][g] = keep + gs_dpp<0xB1>(send);                       // partner l ^ 1
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float keep = b1 ? a[2 * i + 1][g] : a[2 * i][g], send = b1 ? a[2 * i][g] : a[2 * i + 1][g];
            b[i][g] = keep + gs_dpp<0x4E>(send);                       // partner l ^ 2
        }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float keep = b2 ? b[1][g] : b[0][g], send = b2 ? b[0][g] : b[1][g];
        float c = keep + __shfl_xor(send, 4, 64);
        c += gs_dpp<0x128>(c);                                         // row_ror:8 = partner l ^ 8
        c += __shfl_xor(c, 16, 64);
        c += __shfl_xor(c, 32, 64);
        out[g] = c;
    }
}
__device__ __forceinline__ float gs_dot4(const g4& a, const g4& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

__global__ __launch_bounds__(256) void gru_step_fwd_kernel(float* __restrict__ gi_rzn, float* __restrict__ gh,
                                                           const float* __restrict__ hm, const float* __restrict__ w_hh,
                                                           const float* __restrict__ b_hh, const float* __restrict__ m_next,
                                                           float* __restrict__ h_out, float* __restrict__ hm_next, int N, int H) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wave;
    if (j >= H) return;
    const int nq = H / 4;
    for (int n0 = 0; n0 < N; n0 += GS_NR) {
        // the finishing lanes request their gate operands BEFORE the dot products: the loads fly under the k loop
        const int row = n0 + lane;
        const bool fin = lane < GS_NR && row < N;
        const long g = (long)(fin ? row : 0) * 3 * H + j, i = (long)(fin ? row : 0) * H + j;
        const float gi_r = gi_rzn[g], gi_z = gi_rzn[g + H], gi_n = gi_rzn[g + 2 * H], hp = hm[i];
        const float bh_r = b_hh[j], bh_z = b_hh[H + j], bh_n = b_hh[2 * H + j];
        const float mn = (hm_next && fin) ? m_next[row] : 0.f;
        float acc[GS_NR][3];
#pragma unroll
        for (int r = 0; r < GS_NR; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.f;
        for (int q = lane; q < nq; q += 64) {
            const g4 w0 = *reinterpret_cast<const g4*>(w_hh + (long)j * H + 4 * q);
            const g4 w1 = *reinterpret_cast<const g4*>(w_hh + (long)(H + j) * H + 4 * q);
            const g4 w2 = *reinterpret_cast<const g4*>(w_hh + (long)(2 * H + j) * H + 4 * q);
            g4 hv[GS_NR];
#pragma unroll
            for (int r = 0; r < GS_NR; ++r) {
                const int row = n0 + r < N ? n0 + r : N - 1;
                hv[r] = *reinterpret_cast<const g4*>(hm + (long)row * H + 4 * q);
            }
#pragma unroll
            for (int r = 0; r < GS_NR; ++r) {
                acc[r][0] += gs_dot4(hv[r], w0);
                acc[r][1] += gs_dot4(hv[r], w1);
                acc[r][2] += gs_dot4(hv[r], w2);
            }
        }
        float my[3];
        gs_reduce8<3>(acc, my, lane);                                   // lane l: the sums of row n0 + l % 8
        if (fin) {
            const float ghr = my[0] + bh_r, ghz = my[1] + bh_z, ghn = my[2] + bh_n;
            const float r = gru_sigmoid(gi_r + ghr);
            const float z = gru_sigmoid(gi_z + ghz);
            const float nn = tanhf(gi_n + r * ghn);
            const float h = nn + z * (hp - nn);
            gh[g] = ghr; gh[g + H] = ghz; gh[g + 2 * H] = ghn;
            gi_rzn[g] = r; gi_rzn[g + H] = z; gi_rzn[g + 2 * H] = nn;
            h_out[i] = h;
            if (hm_next) hm_next[i] = h * mn;
        }
    }
}

__global__ __launch_bounds__(256) void gru_step_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ m_next,
                                                           float* __restrict__ rzn_dgi, float* __restrict__ gh_dgh,
                                                           const float* __restrict__ hm, const float* __restrict__ dgh_next,
                                                           const float* __restrict__ w_hh_t, float* __restrict__ carry, int last,
                                                           int N, int H) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blockIdx.x * 4 + wave;
    if (u >= H) return;
    const int nq = 3 * H / 4;
    for (int n0 = 0; n0 < N; n0 += GS_NR) {
        const int row = n0 + lane;
        const bool fin = lane < GS_NR && row < N;
        const long g = (long)(fin ? row : 0) * 3 * H + u, i = (long)(fin ? row : 0) * H + u;
        const float r = rzn_dgi[g], z = rzn_dgi[g + H], nn = rzn_dgi[g + 2 * H], ghn = gh_dgh[g + 2 * H], hmv = hm[i];
        const float dov = dout ? dout[i] : 0.f;
        const float cin = last ? 0.f : carry[i], mn = (last || !fin) ? 0.f : m_next[row];
        float mine = 0.f;
        if (!last) {                                                    // (uniform)
            float acc[GS_NR][1];
#pragma unroll
            for (int r = 0; r < GS_NR; ++r) acc[r][0] = 0.f;
            for (int q = lane; q < nq; q += 64) {
                const g4 wv = *reinterpret_cast<const g4*>(w_hh_t + (long)u * 3 * H + 4 * q);
                g4 dv[GS_NR];
#pragma unroll
                for (int r = 0; r < GS_NR; ++r) {
                    const int row = n0 + r < N ? n0 + r : N - 1;
                    dv[r] = *reinterpret_cast<const g4*>(dgh_next + (long)row * 3 * H + 4 * q);
                }
#pragma unroll
                for (int r = 0; r < GS_NR; ++r) acc[r][0] += gs_dot4(dv[r], wv);
            }
            float red[1];
            gs_reduce8<1>(acc, red, lane);
            mine = red[0];
        }
        if (fin) {
            float dh = dov;
            if (!last) dh += (cin + mine) * mn;
            const float dn_pre = (dh * (1.f - z)) * (1.f - nn * nn);
            const float dz_pre = (dh * (hmv - nn)) * (z * (1.f - z));
            const float dr_pre = (dn_pre * ghn) * (r * (1.f - r));
            rzn_dgi[g] = dr_pre; rzn_dgi[g + H] = dz_pre; rzn_dgi[g + 2 * H] = dn_pre;
            gh_dgh[g] = dr_pre; gh_dgh[g + H] = dz_pre; gh_dgh[g + 2 * H] = dn_pre * r;
            carry[i] = dh * z;
        }
    }
}

// dst[c, r] = src[r, c] (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < rows && c0 + tx < cols) tile[k][tx] = src[(long)(r0 + k) * cols + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < cols && r0 + tx < rows) dst[(long)(c0 + k) * rows + r0 + tx] = tile[tx][k];
}

static inline int gru_blocks(long n) {
    long b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace acktr

using namespace acktr;

extern "C" int acktr_gru_mask_rows_f32(const float* h, const float* m, float* out, int N, int H, void* stream) {
    ACKTR_ARG(h && m && out && N > 0 && H > 0, "gru_mask_rows: bad arguments");
    hipLaunchKernelGGL(gru_mask_rows_kernel, dim3(gru_blocks((long)N * H)), dim3(256), 0, (hipStream_t)stream, h, m, out, N, H);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_gru_gates_fwd_f32(float* gi_rzn, const float* gh, const float* hm, const float* m_next, float* h_out,
                                       float* hm_next, int N, int H, void* stream) {
    ACKTR_ARG(gi_rzn && gh && hm && h_out && N > 0 && H > 0, "gru_gates_fwd: bad arguments");
    ACKTR_ARG((hm_next == nullptr) == (m_next == nullptr), "gru_gates_fwd: hm_next and m_next go together");
    hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(gru_blocks((long)N * H)), dim3(256), 0, (hipStream_t)stream, gi_rzn, gh,
                       hm, m_next, h_out, hm_next, N, H);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_gru_step_supported(int N, int H) { return (N >= 1 && N <= 64 && H >= 4 && H % 4 == 0) ? 1 : 0; }

extern "C" int acktr_gru_step_fwd_f32(float* gi_rzn, float* gh, const float* hm, const float* w_hh, const float* b_hh,
                                      const float* m_next, float* h_out, float* hm_next, int N, int H, void* stream) {
    ACKTR_ARG(gi_rzn && gh && hm && w_hh && b_hh && h_out, "gru_step_fwd: null pointer");
    ACKTR_ARG(acktr_gru_step_supported(N, H), "gru_step_fwd: needs 1 <= N <= 64 rows and H %% 4 == 0 (N=%d H=%d)", N, H);
    ACKTR_ARG((hm_next == nullptr) == (m_next == nullptr), "gru_step_fwd: hm_next and m_next go together");
    ACKTR_ARG(((((uintptr_t)hm) | ((uintptr_t)w_hh)) & 15) == 0, "gru_step_fwd: hm and w_hh must be 16-byte aligned");
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(cdiv(H, 4)), dim3(256), 0, (hipStream_t)stream, gi_rzn, gh, hm, w_hh, b_hh,
                       m_next, h_out, hm_next, N, H);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_gru_step_bwd_f32(const float* dout, const float* m_next, float* rzn_dgi, float* gh_dgh, const float* hm,
                                      const float* dgh_next, const float* w_hh_t, float* carry, int last, int N, int H,
                                      void* stream) {
    ACKTR_ARG(rzn_dgi && gh_dgh && hm && carry && w_hh_t, "gru_step_bwd: null pointer");
    ACKTR_ARG(acktr_gru_step_supported(N, H), "gru_step_bwd: needs 1 <= N <= 64 rows and H %% 4 == 0 (N=%d H=%d)", N, H);
    ACKTR_ARG(last || (m_next && dgh_next), "gru_step_bwd: m_next and dgh_next are required unless this is the last step");
    ACKTR_ARG(((((uintptr_t)dgh_next) | ((uintptr_t)w_hh_t)) & 15) == 0, "gru_step_bwd: dgh_next and w_hh_t must be 16-byte aligned");
    hipLaunchKernelGGL(gru_step_bwd_kernel, dim3(cdiv(H, 4)), dim3(256), 0, (hipStream_t)stream, dout, m_next, rzn_dgi, gh_dgh,
                       hm, dgh_next, w_hh_t, carry, last ? 1 : 0, N, H);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_transpose_f32(const float* src, int rows, int cols, float* dst, void* stream) {
    ACKTR_ARG(src && dst && rows > 0 && cols > 0 && src != dst, "transpose: bad arguments");
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, src, rows, cols, dst);
    ACKTR_LAUNCH_CHECK();
    return 0;
}

extern "C" int acktr_gru_gates_bwd_f32(const float* dout, const float* m_next, float* rzn_dgi, float* gh_dgh,
                                       const float* hm, float* carry, int last, int N, int H, void* stream) {
    ACKTR_ARG(rzn_dgi && gh_dgh && hm && carry && N > 0 && H > 0, "gru_gates_bwd: bad arguments");
    ACKTR_ARG(last || m_next, "gru_gates_bwd: m_next is required unless this is the last step of the sequence");
    hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(gru_blocks((long)N * H)), dim3(256), 0, (hipStream_t)stream, dout, m_next,
                       rzn_dgi, gh_dgh, hm, carry, last, N, H);
    ACKTR_LAUNCH_CHECK();
    return 0;
}
