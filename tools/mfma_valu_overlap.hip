// micro-benchmark (dev tool): do FP64 MFMAs of one wave and FP64 / INT32 VALU work of ANOTHER wave on the same SIMD overlap on gfx950?
// Workgroups of 8 waves (2 per SIMD): waves 0-3 run MFMAs, waves 4-7 run v_fma_f64 or integer mads; each role alone, then both together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) k_mix(double* out, int iters_m, int iters_f, int iters_i) {
    const int wv = threadIdx.x >> 6;
    double s = 0;
    if (wv < 4) {
        v4 acc[4]; for (int i = 0; i < 4; ++i) acc[i] = (v4){0, 0, 0, 0};
        double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
        for (int it = 0; it < iters_m; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[8]; for (int i = 0; i < 8; ++i) acc[i] = i;
        double a = threadIdx.x * 1e-3 + 1.0, b = 1e-9;
        for (int it = 0; it < iters_f; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], a, b);
        }
        unsigned x[8]; for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
        for (int it = 0; it < iters_i; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = x[i] * 0xD2511F53u + 0x9E3779B9u;
        }
        for (int i = 0; i < 8; ++i) s += acc[i] + x[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* d; (void)hipMalloc(&d, 8 * 512 * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](int im, int jf, int ji) {
        float ms;
        hipLaunchKernelGGL(k_mix, dim3(256), dim3(512), 0, 0, d, 10, 10, 10); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k_mix, dim3(256), dim3(512), 0, 0, d, im, jf, ji); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        return ms;
    };
    const int im = 4000, jf = 32000, ji = 32000;          // 16k MFMAs (~1.02M cycles), 256k v_fma_f64, 256k v_mul_lo+add
    printf("mfma alone          %.3f ms\n", run(im, 0, 0));
    printf("fma_f64 alone       %.3f ms\n", run(0, jf, 0));
    printf("int32 alone         %.3f ms\n", run(0, 0, ji));
    printf("mfma + fma_f64      %.3f ms\n", run(im, jf, 0));
    printf("mfma + int32        %.3f ms\n", run(im, 0, ji));
    printf("mfma + fma + int32  %.3f ms\n", run(im, jf, ji));
    return 0;
}
