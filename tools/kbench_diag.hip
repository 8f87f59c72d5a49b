// kbench_diag.hip -- one wave factoring + inverting a 16x16 SPD block R times (the critical path of every Cholesky panel step)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Impopis_amd/csrc tools/kbench_diag.hip -o tools/kbench_diag_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "linalg_diag.h"
using namespace mpopis;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(1024) k_diag(const double* A, const double* A21, double* Lout, double* Xout, unsigned long long* cyc, int R) {
    __shared__ DiagScratch dsh;
    __shared__ double W0[16 * 17], W[16 * 17];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (wv == 0) {
        for (int e = lane; e < 256; e += 64) W0[(e & 15) + (e >> 4) * 17] = A[e];
        unsigned long long t0 = 0, t1 = 0;
        for (int r = 0; r < R + 1; ++r) {
            for (int e = lane; e < 256; e += 64) W[(e & 15) + (e >> 4) * 17] = W0[(e & 15) + (e >> 4) * 17];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (r == 1) t0 = wall_clock64();
            diag16_factor(lane, [&](int i, int c) { return W[i + c * 17]; }, [&](int i, int c, double v) { W[i + c * 17] = v; }, dsh);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        t1 = wall_clock64();
        if (lane == 0) cyc[0] = t1 - t0;
        for (int e = lane; e < 256; e += 64) { const int i = e & 15, c = e >> 4; Lout[e] = dsh.L[i][c]; }
        // panel solve of one tile, timed over R repetitions
        const int li = lane & 15, lk = lane >> 4;
        double a[4];
        t0 = wall_clock64();
        for (int r = 0; r < R; ++r) {
            const PanelOps o = panel_solve_operands(lane, dsh);
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = A21[li + (4 * q + lk) * 16];
            panel_solve_tile(o, a);
            __builtin_amdgcn_sched_barrier(0);
        }
        t1 = wall_clock64();
        if (lane == 0) cyc[1] = t1 - t0;
#pragma unroll
        for (int q = 0; q < 4; ++q) Xout[li + (4 * q + lk) * 16] = a[q];
    }
    __syncthreads();
}

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 200, nthr = argc > 2 ? atoi(argv[2]) : 64;
    std::vector<double> A(256), M(256), A21(256);
    srand(3);
    for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
    for (auto& v : A21) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = (i == j) ? 0.5 : 0.0; for (int k = 0; k < 16; ++k) s += M[i + 16 * k] * M[j + 16 * k]; A[i + 16 * j] = s; }
    double *dA, *dA21, *dL, *dX; unsigned long long* dc;
    CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dA21, 2048)); CK(hipMalloc(&dL, 2048)); CK(hipMalloc(&dX, 2048)); CK(hipMalloc(&dc, 16));
    CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dA21, A21.data(), 2048, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_diag, dim3(1), dim3(nthr), 0, 0, dA, dA21, dL, dX, dc, R);
    CK(hipDeviceSynchronize());
    unsigned long long c[2]; CK(hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost));
    std::vector<double> L(256), X(256);
    CK(hipMemcpy(L.data(), dL, 2048, hipMemcpyDeviceToHost)); CK(hipMemcpy(X.data(), dX, 2048, hipMemcpyDeviceToHost));
    // host reference factor in long double
    std::vector<long double> Lr(256, 0.0L);
    for (int j = 0; j < 16; ++j) for (int i = j; i < 16; ++i) { long double s = A[i + 16 * j]; for (int k = 0; k < j; ++k) s -= Lr[i + 16 * k] * Lr[j + 16 * k]; Lr[i + 16 * j] = (i == j) ? sqrtl(s) : s / Lr[j + 16 * j]; }
    double e1 = 0, e2 = 0, e3 = 0, up = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0, t = 0;
        for (int k = 0; k < 16; ++k) { s += L[i + 16 * k] * L[j + 16 * k]; t += X[i + 16 * k] * L[j + 16 * k]; }
        e1 = fmax(e1, fabs(s - A[i + 16 * j])); e2 = fmax(e2, fabs(t - A21[i + 16 * j]));
        if (i >= j) e3 = fmax(e3, fabs((double)(L[i + 16 * j] - Lr[i + 16 * j]) / (double)Lr[j + 16 * j])); else up = fmax(up, fabs(L[i + 16 * j]));
    }
    printf("diag16_factor: %.3f us per block; panel_solve_tile %.3f us per tile (R = %d, %d threads)\n", c[0] * 0.01 / R, c[1] * 0.01 / R, R, nthr);
    printf("   |LL'-A| = %.2e, |L - Lref|/l_jj = %.2e, upper %.1e, |X L' - A21| = %.2e\n", e1, e3, up, e2);
    return 0;
}
