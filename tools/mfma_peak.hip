// Bare fp32 MFMA rate of the box: v_mfma_f32_32x32x2_f32 on register operands, nothing else in the loop.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_peak tools/mfma_peak.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// RANDOM: eight different pseudo-random operand pairs per lane rotate through the loop (data toggling like a real
// GEMM, which is what sets the clock the chip sustains); otherwise one near-constant pair.
template <int CHAINS, bool RANDOM>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a[8], b[8];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int u = 0; u < 8; ++u) {
        if (RANDOM) {
            h = h * 1664525u + 1013904223u; a[u] = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
            h = h * 1664525u + 1013904223u; b[u] = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
        } else { a[u] = a0 + threadIdx.x * 1e-3f; b[u] = b0 - threadIdx.x * 1e-3f; }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS, bool RANDOM>
void run(int blocks_per_cu, int cus, float* out) {
    const int iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<CHAINS, RANDOM>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)cus * blocks_per_cu * 4 * iters * 8 * CHAINS * 4096.0;
        if (rep == 2) printf("%s operands, chains %d, %d workgroups/CU (x4 waves): %.3f ms  %.1f TFLOP/s\n", RANDOM ? "random  " : "constant", CHAINS, blocks_per_cu, ms, flop / ms / 1e9);
    }
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    float* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    run<1, false>(1, cus, out); run<4, false>(1, cus, out); run<1, false>(3, cus, out);
    run<1, true>(1, cus, out); run<4, true>(1, cus, out); run<1, true>(3, cus, out); run<2, true>(2, cus, out);
    hipFree(out);
    return 0;
}
