// invsqrt_quad.h -- quadrature nodes for x^-1/2 on [m, M] (host + device).
//
// Hale, Higham & Trefethen, "Computing A^alpha, log(A) and related matrix functions by contour integrals"
// (SIAM J. Numer. Anal. 46, 2008), method 3 specialised to the square root: with k^2 = m/M, K' = K(1 - k^2) and
// u_j = (j - 1/2) K'/N, all Jacobi functions taken with parameter 1 - k^2,
//     x^-1/2  ~=  sum_j  w_j / (x + s_j),     s_j = m sn_j^2/cn_j^2,    w_j = (2 K' sqrt(m) / (pi N)) dn_j / cn_j^2
// for every x in [m, M], with error O(exp(-2 pi^2 N / (ln(M/m) + 3))): N = 64 nodes (one wavefront, lane = node) give
// < 1e-15 up to M/m = 1e12.  The engine applies it to the Lanczos tridiagonal of the CMA covariance (kernels_invsqrt.hip),
// where every term is a positive-definite tridiagonal solve.
//
// sn, cn, dn and K' come from the arithmetic-geometric-mean (descending Landen) recurrence, Abramowitz & Stegun 16.4/17.6;
// dn is formed as sqrt(cn^2 + k^2 sn^2) (no cancellation when the parameter is close to 1).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MPQ_HD __host__ __device__ inline
#else
#define MPQ_HD inline
#endif

namespace mpopis {

constexpr int kQuadAgmMax = 16;

// node `j` of `N` for the interval [m, M]; returns false when m/M is outside what the AGM resolves in double precision
// a, c: scratch for the AGM sequence (kQuadAgmMax + 1 doubles each; the values are the same for every node).  On the device the caller passes LDS: a
// dynamically indexed private array lives in scratch MEMORY, and the two recurrences then pay a memory round trip per level (k_lanczos_prep: 89 us
// beside a busy chip, with these arrays in LDS a quarter of that)
MPQ_HD bool invsqrt_quad_node(double m, double M, int j, int N, double* shift, double* weight, double* a, double* c) {
    const double k2 = m / M;
    if (!(k2 > 1e-14) || !(k2 <= 0.75)) return false;         // callers clamp m <= M/2 (a smaller lower bound stays valid)
    const double mpar = 1.0 - k2;
    a[0] = 1.0; c[0] = sqrt(mpar);
    double b = sqrt(k2);
    int n = 0;
    while (fabs(c[n]) > 1e-17 * a[n] && n < kQuadAgmMax) {
        const double an = 0.5 * (a[n] + b), cn_ = 0.5 * (a[n] - b);
        b = sqrt(a[n] * b);
        ++n; a[n] = an; c[n] = cn_;
    }
    const double Kp = 1.5707963267948966 / a[n];
    const double u = (j + 0.5) * Kp / N;
    double phi = ldexp(a[n] * u, n);
    for (int i = n; i >= 1; --i) phi = 0.5 * (phi + asin(c[i] * sin(phi) / a[i]));
    const double sn = sin(phi), cn = cos(phi);
    const double dn = sqrt(cn * cn + k2 * sn * sn);
    const double icn2 = 1.0 / (cn * cn);
    *shift = m * (sn * sn) * icn2;
    *weight = (2.0 * Kp * sqrt(m) / (3.14159265358979323846 * N)) * dn * icn2;
    return true;
}
MPQ_HD bool invsqrt_quad_node(double m, double M, int j, int N, double* shift, double* weight) {
    double a[kQuadAgmMax + 1], c[kQuadAgmMax + 1];
    return invsqrt_quad_node(m, M, j, N, shift, weight, a, c);
}

}  // namespace mpopis
