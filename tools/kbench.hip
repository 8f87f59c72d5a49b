// kbench.hip -- standalone micro-benchmark of individual engine kernels (dev tool, not shipped).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/kbench.hip -o /tmp/kb.o && hipcc --offload-arch=gfx950 /tmp/kb.o mpopis_amd/lib/obj/kernels_{linalg,mfma,sample}.o -o tools/kbench_bin
#include "../mpopis_amd/csrc/engine.h"
#include <cstdio>
#include <vector>
#include <random>
#include <cstring>
#include <algorithm>
using namespace mpopis;


#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class F> float timeit(F f, int reps, hipStream_t s) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, cs = argc > 2 ? atoi(argv[2]) : 100, K = argc > 3 ? atoi(argv[3]) : 4096;
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t nn = (size_t)cs * cs;
    std::mt19937_64 rng(1); std::normal_distribution<double> nd;
    std::vector<double> A(nn * B), Z((size_t)B * cs * K), w((size_t)B * K), mu((size_t)B * cs, 0.0);
    for (int b = 0; b < B; ++b) {
        std::vector<double> M(nn);
        for (auto& v : M) v = nd(rng) * 0.1;
        for (int i = 0; i < cs; ++i) for (int j = 0; j < cs; ++j) {
            double v = 0; for (int k = 0; k < cs; ++k) v += M[i + (size_t)k * cs] * M[j + (size_t)k * cs];
            A[b * nn + i + (size_t)j * cs] = v + (i == j ? 0.1 : 0.0);
        }
    }
    for (auto& v : Z) v = nd(rng);
    for (auto& v : w) v = 1.0 / K;
    double *dA, *dL, *dZ, *dE, *dw, *dmu, *dS, *dpart; int *dstatus, *dact;
    CK(hipMalloc(&dA, nn * B * 8)); CK(hipMalloc(&dL, nn * B * 8)); CK(hipMalloc(&dZ, Z.size() * 8)); CK(hipMalloc(&dE, Z.size() * 8));
    CK(hipMalloc(&dw, w.size() * 8)); CK(hipMalloc(&dmu, mu.size() * 8)); CK(hipMalloc(&dS, nn * B * 8));
    const int ksplit = getenv("KSPLIT") ? atoi(getenv("KSPLIT")) : std::max(1, std::min(std::min(32, K / 128), std::max(1, 512 / B)));
    printf("ksplit=%d\n", ksplit);
    CK(hipMalloc(&dpart, wcov_mfma_workspace_doubles(B, cs, ksplit) * 8));
    CK(hipMalloc(&dstatus, B * 4)); CK(hipMalloc(&dact, B * 4));
    CK(hipMemcpy(dA, A.data(), nn * B * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dZ, Z.data(), Z.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, w.data(), w.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dmu, mu.data(), mu.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dstatus, 0, B * 4));
    std::vector<int> ones(B, 1); CK(hipMemcpy(dact, ones.data(), B * 4, hipMemcpyHostToDevice));
    printf("B=%d cs=%d K=%d\n", B, cs, K);
    printf("potrf            %8.1f us\n", timeit([&] { launch_potrf(dA, nn, dL, B, cs, nullptr, dstatus, dact, s); }, 20, s));
    // check potrf: L L' = A for b = 0
    {
        std::vector<double> L(nn); CK(hipMemcpy(L.data(), dL, nn * 8, hipMemcpyDeviceToHost));
        double err = 0; for (int i = 0; i < cs; ++i) for (int j = 0; j <= i; ++j) { double v = 0; for (int k = 0; k <= j; ++k) v += L[i + (size_t)k * cs] * L[j + (size_t)k * cs]; err = fmax(err, fabs(v - A[i + (size_t)j * cs])); }
        printf("   potrf max |LL'-A| = %.3e\n", err);
    }
    {
        uint64_t* dseeds; CK(hipMalloc(&dseeds, B * 8));
        std::vector<uint64_t> hs(B); for (int b = 0; b < B; ++b) hs[b] = 20240000 + b;
        CK(hipMemcpy(dseeds, hs.data(), B * 8, hipMemcpyHostToDevice));
        float ts = timeit([&] { launch_sample_normal(dE, B, cs, K, 2, 0, dseeds, 3u, 1u, nullptr, dact, s); }, 20, s);
        printf("sample_normal    %8.1f us  (%.2f Gnormals/s)\n", ts, (double)B * cs * K / (ts * 1e-6) / 1e9);
        CK(hipMemcpy(dL, A.data(), nn * B * 8, hipMemcpyHostToDevice));
        launch_potrf(dA, nn, dL, B, cs, nullptr, dstatus, dact, s);
        float tf = timeit([&] { launch_sample_trmm_fused(dL, nn, dE, B, cs, K, dseeds, 3u, 1u, dact, s); }, 20, s);
        printf("sample+trmm fused%8.1f us\n", tf);
    }
    float t = timeit([&] { launch_trmm_LZ_mfma(dL, nn, dZ, dE, B, cs, K, dact, s); }, 20, s);
    printf("trmm_mfma        %8.1f us  (%.1f TF half-counted)\n", t, (double)B * cs * cs * K / (t * 1e-6) / 1e12);
    t = timeit([&] { launch_wcov_mfma(dZ, dw, nullptr, K, dmu, dS, dpart, B, cs, K, ksplit, 0, 0.0, 1e-8, dact, s, nullptr); }, 20, s);
    printf("wcov_mfma(+fin)  %8.1f us  (%.1f TF half-counted)\n", t, (double)B * cs * cs * K / (t * 1e-6) / 1e12);
    {
        std::vector<double> S(nn); CK(hipMemcpy(S.data(), dS, nn * 8, hipMemcpyDeviceToHost));
        double err = 0; for (int i = 0; i < cs; i += 7) for (int j = 0; j < cs; j += 5) { double v = 0; for (int k = 0; k < K; ++k) v += Z[(size_t)i * K + k] * Z[(size_t)j * K + k] / K; if (i == j) v += 1e-8; err = fmax(err, fabs(v - S[i + (size_t)j * cs])); }
        printf("   wcov max err = %.3e\n", err);
    }
    return 0;
}
