// kbench_c5.hip -- the two matrix-core kernels of the C5 AIS iteration (fused sampler, moments scatter) alone on the chip, each against a private
// build with its VALU side work compiled out: the floor a cheaper generator / free staging would leave (dev tool, not shipped).
//   build: bash tools/build_kbench_c5.sh      usage: tools/kbench_c5_bin [B] [cs] [K]   (and tools/kbench_c5_nodraw_bin, tools/kbench_c5_nostage_bin)
#include "../mpopis_amd/csrc/engine.h"
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
using namespace mpopis;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <class F> float timeit(F f, int reps, hipStream_t s) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, cs = argc > 2 ? atoi(argv[2]) : 100, K = argc > 3 ? atoi(argv[3]) : 4096;
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t nn = (size_t)cs * cs, per = (size_t)cs * K;
    std::mt19937_64 rng(1); std::normal_distribution<double> nd;
    std::vector<double> A(nn * B, 0.0);
    for (int b = 0; b < B; ++b) for (int i = 0; i < cs; ++i) A[b * nn + i + (size_t)i * cs] = (i & 1) ? 0.1 : 0.0625;
    double *dA, *dL, *dLp, *dE, *dmu, *dS, *dpart, *dtab, *dcost; int* dstatus; uint64_t* dseeds; unsigned long long* dcmin;
    const int ksplit = std::max(1, std::min(std::min(32, K / 128), std::max(1, 512 / B)));
    CK(hipMalloc(&dA, nn * B * 8)); CK(hipMalloc(&dL, nn * B * 8)); CK(hipMalloc(&dLp, std::max<size_t>(1, potrf_panel_doubles(cs)) * B * 8)); CK(hipMalloc(&dE, per * B * 8));
    CK(hipMalloc(&dmu, (size_t)B * cs * 8)); CK(hipMalloc(&dS, nn * B * 8)); CK(hipMalloc(&dpart, wcov_mfma_workspace_doubles(B, cs, ksplit) * 8)); CK(hipMalloc(&dtab, 4096 * 8));
    CK(hipMalloc(&dcost, (size_t)B * K * 8)); CK(hipMalloc(&dstatus, B * 4)); CK(hipMalloc(&dseeds, B * 8)); CK(hipMalloc(&dcmin, B * 8));
    CK(hipMemcpy(dA, A.data(), nn * B * 8, hipMemcpyHostToDevice)); CK(hipMemset(dstatus, 0, B * 4)); CK(hipMemset(dmu, 0, (size_t)B * cs * 8));
    std::vector<uint64_t> seeds(B); for (int b = 0; b < B; ++b) seeds[b] = 1000 + b;
    CK(hipMemcpy(dseeds, seeds.data(), B * 8, hipMemcpyHostToDevice));
    std::vector<double> cost((size_t)B * K); for (auto& v : cost) v = 100.0 + 30.0 * nd(rng);
    CK(hipMemcpy(dcost, cost.data(), cost.size() * 8, hipMemcpyHostToDevice));
    launch_rng_tab_init(dtab, s);
    launch_potrf(dA, nn, dL, B, cs, nullptr, dstatus, nullptr, s, CoopCtx(), dLp, potrf_panel_doubles(cs));
    CK(hipStreamSynchronize(s));
    printf("B=%d cs=%d K=%d ksplit=%d\n", B, cs, K, ksplit);
    const float tf = timeit([&] { launch_sample_trmm_fused(dL, nn, dE, B, cs, K, dseeds, 3u, 1u, nullptr, s, dtab, dLp, potrf_panel_doubles(cs), nullptr); }, 20, s);
    printf("fused sampler (k_trmm_LZ_mfma<true>)        %8.1f us\n", tf);
    std::vector<double> w((size_t)B * K, 1.0 / K); double* dw; CK(hipMalloc(&dw, w.size() * 8)); CK(hipMemcpy(dw, w.data(), w.size() * 8, hipMemcpyHostToDevice));
    const float tw = timeit([&] { launch_wcov_mfma(dE, dw, nullptr, K, dmu, dS, dpart, B, cs, K, ksplit, -1, 1.0, 1e-8, nullptr, s, nullptr, dmu, nullptr, nullptr); }, 20, s);
    printf("moments, compact form (partial + finish)    %8.1f us\n", tw);
    const float tr = timeit([&] { launch_wcov_mfma(dE, dw, nullptr, K, dmu, dS, dpart, B, cs, K, ksplit, 0, 1.0, 1e-8, nullptr, s, nullptr, dmu, nullptr, nullptr); }, 20, s);
    printf("moments, row form where it applies          %8.1f us\n", tr);
    return 0;
}
