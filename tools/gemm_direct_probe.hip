// Stand-alone timing / correctness probe of tools/gemm_direct.hpp on the FC-layer shapes of NatureCNN at B = 1024
// (forward 1024x512x1568, dgrad 1024x1568x512, wgrad 512x1568x1024).  Build + run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I tools -I pytorch-a2c-ppo-acktr-gail_amd/csrc -I include \
//         tools/gemm_direct_probe.hip pytorch-a2c-ppo-acktr-gail_amd/csrc/errors.hip -o tools/bin/gemm_direct_probe   (argument: dbg bits)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gemm_direct.hpp"

using namespace acktr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static std::vector<float> rnd(size_t n, unsigned seed) {
    std::vector<float> v(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; v[i] = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    return v;
}

static int g_dbg = 0;
struct Case { const char* name; int M, N, K; bool akc, bkc; long lda, ldb; };

template <int MB, int NB, bool AKC, bool BKC, int WM, int WN>
static void run(const char* tag, const Case& c, const float* dA, const float* dB, float* dC, int ns, const std::vector<float>& hA,
                const std::vector<float>& hB) {
    DirectStore epi{dC, (long)c.N, (long)c.M * c.N};
    hipStream_t s = 0;
    for (int i = 0; i < 3; ++i)
        if (launch_gemm_direct<MB, NB, AKC, BKC, WM, WN>(dA, c.lda, dB, c.ldb, epi, c.M, c.N, c.K, ns, s, g_dbg)) { printf("launch failed\n"); exit(1); }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int it = 50;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < it; ++i) launch_gemm_direct<MB, NB, AKC, BKC, WM, WN>(dA, c.lda, dB, c.ldb, epi, c.M, c.N, c.K, ns, s, g_dbg);
    CK(hipEventRecord(e1, s));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / it;
    std::vector<float> hC((size_t)ns * c.M * c.N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 400; ++t) {
        const int m = (t * 7919 + (t < 8 ? c.M - 1 - t : 0)) % c.M, n = (t * 104729 + (t < 8 ? c.N - 1 - t : 3)) % c.N;
        double ref = 0;
        for (int k = 0; k < c.K; ++k) {
            const double av = c.akc ? hA[(size_t)m * c.lda + k] : hA[(size_t)k * c.lda + m];
            const double bv = c.bkc ? hB[(size_t)n * c.ldb + k] : hB[(size_t)k * c.ldb + n];
            ref += av * bv;
        }
        double got = 0;
        for (int z = 0; z < ns; ++z) got += hC[(size_t)z * c.M * c.N + (size_t)m * c.N + n];
        worst = fmax(worst, fabs(got - ref));
    }
    printf("%-6s %-28s slices %2d : %7.2f us  %6.1f TFLOP/s   max abs err %.2e\n", c.name, tag, ns, us,
           2.0 * c.M * c.N * c.K / (us * 1e-6) / 1e12, worst);
}

int main(int argc, char** argv) {
    if (argc > 1) g_dbg = atoi(argv[1]);
    const int B = 1024, IN = 1568, OUT = 512;
    std::vector<float> X = rnd((size_t)B * IN, 1), W = rnd((size_t)OUT * IN, 2), dY = rnd((size_t)B * OUT, 3);
    float *dX, *dW, *ddY, *dC;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&ddY, dY.size() * 4));
    CK(hipMalloc(&dC, (size_t)16 * B * IN * 4));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ddY, dY.data(), dY.size() * 4, hipMemcpyHostToDevice));
    const Case fwd{"fwd", B, OUT, IN, true, true, IN, IN};
    const Case dg{"dgrad", B, IN, OUT, true, false, OUT, IN};
    const Case wg{"wgrad", OUT, IN, B, false, false, OUT, IN};
    {
        const Case tiny{"K=32", B, OUT, 32, true, true, IN, IN}, k64{"K=64", B, OUT, 64, true, true, IN, IN}, k256{"K=256", B, OUT, 256, true, true, IN, IN};
        run<2, 2, true, true, 2, 2>("64x64 waves, 2x2", tiny, dX, dW, dC, 1, X, W);
        run<2, 2, true, true, 2, 2>("64x64 waves, 2x2", k64, dX, dW, dC, 1, X, W);
        run<2, 2, true, true, 2, 2>("64x64 waves, 2x2", k256, dX, dW, dC, 1, X, W);
        run<2, 2, true, true, 2, 2>("64x64 waves, 2x2", k256, dX, dW, dC, 7, X, W);
    }
    run<2, 2, true, true, 2, 2>("64x64 waves, 2x2", fwd, dX, dW, dC, 7, X, W);
    run<2, 2, true, true, 2, 2>("64x64 waves, 2x2", fwd, dX, dW, dC, 8, X, W);
    run<2, 1, true, true, 2, 2>("64x32 waves, 2x2", fwd, dX, dW, dC, 4, X, W);
    run<2, 1, true, true, 1, 1>("64x32 waves, 1x1", fwd, dX, dW, dC, 4, X, W);
    run<2, 2, true, true, 1, 1>("64x64 waves, 1x1", fwd, dX, dW, dC, 7, X, W);
    run<1, 2, true, true, 2, 2>("32x64 waves, 2x2", fwd, dX, dW, dC, 4, X, W);
    run<2, 2, true, true, 2, 1>("64x64 waves, 2x1", fwd, dX, dW, dC, 4, X, W);
    run<1, 2, true, false, 2, 2>("32x64 waves, 2x2", dg, ddY, dW, dC, 1, dY, W);
    run<2, 2, true, false, 2, 2>("64x64 waves, 2x2", dg, ddY, dW, dC, 2, dY, W);
    run<2, 4, true, false, 2, 1>("64x128 waves, 2x1", dg, ddY, dW, dC, 4, dY, W);
    run<1, 4, true, false, 2, 1>("32x128 waves, 2x1", dg, ddY, dW, dC, 2, dY, W);
    run<1, 4, true, false, 4, 1>("32x128 waves, 4x1", dg, ddY, dW, dC, 2, dY, W);
    run<2, 2, false, false, 2, 2>("64x64 waves, 2x2", wg, ddY, dX, dC, 5, dY, X);
    run<4, 2, false, false, 1, 2>("128x64 waves, 1x2", wg, ddY, dX, dC, 8, dY, X);
    run<2, 4, false, false, 2, 1>("64x128 waves, 2x1", wg, ddY, dX, dC, 8, dY, X);
    run<2, 2, false, false, 2, 2>("64x64 waves, 2x2", wg, ddY, dX, dC, 4, dY, X);
    run<2, 2, false, false, 1, 1>("64x64 waves, 1x1", wg, ddY, dX, dC, 5, dY, X);
    return 0;
}
