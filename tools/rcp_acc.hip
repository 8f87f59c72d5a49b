// accuracy of the gfx950 v_rcp_f64 / v_rsq_f64 seeds and of the Newton refinements used in car_dynamics.h
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdint.h>
__global__ void k(const double* x, double* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r0 = __builtin_amdgcn_rcp(v);
    double r1 = fma(fma(-v, r0, 1.0), r0, r0);
    double r2 = fma(fma(-v, r1, 1.0), r1, r1);
    double y = __builtin_amdgcn_rsq(v);
    double g = v * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    double g1 = fma(fma(-g, g, v), h, g);
    double g2 = fma(fma(-g1, g1, v), h, g1);
    o[i * 8 + 0] = r0; o[i * 8 + 1] = r1; o[i * 8 + 2] = r2; o[i * 8 + 3] = y; o[i * 8 + 4] = g; o[i * 8 + 5] = g1; o[i * 8 + 6] = g2; o[i * 8 + 7] = sqrt(v);
}
int main() {
    const int n = 1 << 20;
    double* hx = (double*)malloc(n * 8); double* ho = (double*)malloc(n * 64);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); hx[i] = exp((u - 0.5) * 60.0); }
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 64);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(ho, dout, n * 64, hipMemcpyDeviceToHost);
    double e[8] = {0};
    for (int i = 0; i < n; ++i) {
        long double v = hx[i], rc = 1.0L / v, sq = sqrtl(v), rs = 1.0L / sq;
        long double ref[8] = {rc, rc, rc, rs, sq, sq, sq, sq};
        for (int j = 0; j < 8; ++j) { double err = (double)fabsl(((long double)ho[i * 8 + j] - ref[j]) / ref[j]); if (err > e[j]) e[j] = err; }
    }
    const char* nm[8] = {"rcp raw", "rcp 1NR", "rcp 2NR", "rsq raw", "sqrt coupled", "sqrt +1corr", "sqrt +2corr", "sqrt builtin"};
    for (int j = 0; j < 8; ++j) printf("%-14s max rel err %.3e (%.2f ulp)\n", nm[j], e[j], e[j] / 1.11e-16);
    return 0;
}
