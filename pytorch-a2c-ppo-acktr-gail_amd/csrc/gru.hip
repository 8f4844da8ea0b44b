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

extern "C" int acktr_gru_gates_bwd_f32(const float* dout, const float* m_next, float* rzn_dgi, float* gh_dgh,
                                       const float* hm, float* carry, int last, int N, int H, void* stream) {
    ACKTR_ARG(rzn_dgi && gh_dgh && hm && carry && N > 0 && H > 0, "gru_gates_bwd: bad arguments");
    ACKTR_ARG(last || m_next, "gru_gates_bwd: m_next is required unless this is the last step of the sequence");
    hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(gru_blocks((long)N * H)), dim3(256), 0, (hipStream_t)stream, dout, m_next,
                       rzn_dgi, gh_dgh, hm, carry, last, N, H);
    ACKTR_LAUNCH_CHECK();
    return 0;
}
