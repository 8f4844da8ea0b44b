// Bare fp32 MFMA loops on random operands: what clock does the chip hold, and what TFLOP/s does that make?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_clock_probe.hip -o tools/bin/mfma_clock_probe && tools/bin/mfma_clock_probe
// Two waves per SIMD on every CU, operands in registers (no memory traffic in the loop), 8 independent accumulator
// chains; the in-kernel clock is delta(s_memtime) / delta(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ out, long long* stamps, int iters) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 0xfffff]; b[i] = in[(t * 16 + 8 + i) & 0xfffff]; }
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[8];
        for (int q = 0; q < 8; ++q) for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[q], acc[q], 0, 0, 0);
        for (int q = 0; q < 8; ++q) for (int i = 0; i < 16; ++i) sum += acc[q][i];
    } else {
        f32x4 acc[32];
        for (int q = 0; q < 32; ++q) for (int i = 0; i < 4; ++i) acc[q][i] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < 32; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(s + q) & 7], b[q & 7], acc[q], 0, 0, 0);
        for (int q = 0; q < 32; ++q) for (int i = 0; i < 4; ++i) sum += acc[q][i];
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[t] = sum;
    if ((threadIdx.x & 63) == 0) { stamps[2 * (t >> 6)] = c1 - c0; stamps[2 * (t >> 6) + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(const char* name, const float* in, float* out, long long* stamps, int cus, int iters, double flop_per_wave_iter) {
    const int waves = cus * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 40; ++rep) hipLaunchKernelGGL(probe<SHAPE>, dim3(cus), dim3(512), 0, 0, in, out, stamps, iters);   // warm: ~2 s
    hipEventRecord(e0);
    const int reps = 5;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(probe<SHAPE>, dim3(cus), dim3(512), 0, 0, in, out, stamps, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> st(2 * waves);
    hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk(waves);
    for (int w = 0; w < waves; ++w) clk[w] = (double)st[2 * w] / (double)st[2 * w + 1] * 0.1;      // GHz
    std::sort(clk.begin(), clk.end());
    const double tf = flop_per_wave_iter * iters * waves * reps / (ms * 1e-3) * 1e-12;
    printf("%-10s %8.3f ms/launch  %7.1f TFLOP/s  in-kernel clock median %.3f GHz (min %.3f max %.3f)\n", name, ms / reps, tf,
           clk[waves / 2], clk.front(), clk.back());
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *in, *out;
    long long* stamps;
    std::vector<float> h(1 << 20);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMalloc(&in, h.size() * 4);
    hipMalloc(&out, (size_t)cus * 512 * 4);
    hipMalloc(&stamps, (size_t)cus * 8 * 16);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 12000;      // 64 MFMAs of 64 cycles (or 128 of 32) per iteration: ~49 M cycles per wave, 2 waves per SIMD
    run<32>("32x32x2", in, out, stamps, cus, iters, 64.0 * 2 * 32 * 32 * 2);
    run<16>("16x16x4", in, out, stamps, cus, iters, 128.0 * 2 * 16 * 16 * 4);
    run<32>("32x32x2", in, out, stamps, cus, iters, 64.0 * 2 * 32 * 32 * 2);
    run<16>("16x16x4", in, out, stamps, cus, iters, 128.0 * 2 * 16 * 16 * 4);
    return 0;
}
