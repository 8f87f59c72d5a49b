// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950 (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_mfma(double* out, int iters) {
    v4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k_fma(double* out, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = threadIdx.x * 1e-3 + 1.0, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* d; hipMalloc(&d, 8 * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wpb : {1, 2, 4, 8}) {
        float ms;
        k_mfma<4><<<1024, 64 * wpb>>>(d, 10); hipDeviceSynchronize();
        hipEventRecord(e0); k_mfma<4><<<1024, 64 * wpb>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        double n = 1024.0 * wpb * iters * 4;            // wave-level MFMAs
        printf("mfma_f64_16x16x4  waves/block=%d: %.2f ms  -> %.1f TFLOP/s, %.1f cycles per MFMA per SIMD (at 2.4 GHz, 1024 SIMDs)\n", wpb, ms,
               n * 2048 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 * 1024 / n);
        k_fma<8><<<1024, 64 * wpb>>>(d, 10); hipDeviceSynchronize();
        hipEventRecord(e0); k_fma<8><<<1024, 64 * wpb>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        n = 1024.0 * wpb * iters * 8;
        printf("v_fma_f64         waves/block=%d: %.2f ms  -> %.1f TFLOP/s, %.1f cycles per wave-FMA per SIMD\n", wpb, ms,
               n * 128 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 * 1024 / n);
    }
    return 0;
}
