// philox.h -- Philox4x32-10 (Salmon et al., SC'11) + Box-Muller on device; shared by the sampler and the fused
// sampler+unwhiten kernel.  Bit-identical counters/keys to oracle/mpopis_oracle.c (orc_philox_normals).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpopis {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;   // one v_mad_u64_u32 each
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// log(u) for a normal-range u in (0, 1): the classic argument reduction u = 2^k m, m in [sqrt(1/2), sqrt(2)), s = f/(2+f),
// f = m - 1, and the degree-14 odd minimax polynomial in s of Sun's fdlibm (e_log.c; error < 1 ulp).  No special cases
// (the Box-Muller uniforms are (i + 0.5) 2^-53, clamped below 1), no double-double arithmetic: ~35 VALU ops.
__device__ __forceinline__ double log_unit(double u) {
    int k = __builtin_amdgcn_frexp_exp(u);                     // u = m 2^k, m in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(u);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m; k = lo ? k - 1 : k;                    // m in [sqrt(1/2), sqrt(2))
    const double f = m - 1.0;
    const double d = 2.0 + f;
    double rd = __builtin_amdgcn_rcp(d);
    rd = fma(fma(-d, rd, 1.0), rd, rd);
    rd = fma(fma(-d, rd, 1.0), rd, rd);
    const double s = f * rd, z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)k;
    return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

// (sin, cos)(2 pi u) for u in (0, 1): octant q = floor(8u), f = frac(8u) (both exact), reflected in odd octants, then the
// fdlibm kernel polynomials on [0, pi/4] (k_sin.c / k_cos.c; < 1 ulp each) and the octant symmetries.  No large-argument
// path, no special cases: ~35 VALU ops (OCML's sincospi: ~50 + its constants).
__device__ __forceinline__ void sincos_2pi_unit(double u, double* sn, double* cs) {
    const double t = 8.0 * u;
    const double fl = __builtin_floor(t);
    const int q = (int)fl;
    double f = t - fl;
    f = (q & 1) ? 1.0 - f : f;
    const double y = f * 0.78539816339744830962, z = y * y;
    double r = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    r = fma(z, r, 2.75573137070700676789e-06);
    r = fma(z, r, -1.98412698298579493134e-04);
    r = fma(z, r, 8.33333333332248946124e-03);
    const double s = fma(z * y, fma(z, r, -1.66666666666666324348e-01), y);
    double c = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    c = fma(z, c, -2.75573143513906633035e-07);
    c = fma(z, c, 2.48015872894767294178e-05);
    c = fma(z, c, -1.38888888888741095749e-03);
    c = fma(z, c, 4.16666666666666019037e-02);
    c = fma(z * z, c, fma(-0.5, z, 1.0));
    const bool swap = ((q + 1) & 2) != 0;                      // octants 1, 2, 5, 6
    const double ss = swap ? c : s, cc = swap ? s : c;
    *sn = (q & 4) ? -ss : ss;                                  // lower half plane
    *cs = ((q + 2) & 4) ? -cc : cc;                            // left half plane (octants 2..5)
}

__device__ __forceinline__ void philox_normal_pair(uint64_t seed, uint32_t slo, uint32_t shi, uint64_t j, double* z0, double* z1) {
    uint32_t r[4];
    philox4x32_10((uint32_t)j, (uint32_t)(j >> 32), slo, shi, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    // u = ((a >> 11) + 0.5) 2^-53 for the 64-bit word a = (hi:lo), one rounding: (a >> 11) = hi 2^21 + (lo >> 11), so
    // u = hi 2^-32 + ((lo >> 11) + 0.5) 2^-53 with both conversions exact -- two v_cvt_f64_u32 and two fmas per uniform
    // (the largest word, a >> 11 = 2^53 - 1, would round to u1 = 1.0 -> log = 0 -> rsq(0) = inf -> NaN normals: keep u1 < 1)
    const double u1 = fmin(fma((double)r[1], 0x1p-32, fma((double)(r[0] >> 11), 0x1p-53, 0x1p-54)), 1.0 - 0x1p-53);
    const double u2 = fma((double)r[3], 0x1p-32, fma((double)(r[2] >> 11), 0x1p-53, 0x1p-54));
    // sqrt of a positive normal-range number: v_rsq_f64 seed + coupled Newton step + residual correction (1 ulp, see
    // tools/rcp_acc.hip) instead of the library sqrt with its denormal rescaling (8 instead of 18 VALU ops)
    const double v = -2.0 * log_unit(u1);
    const double y = __builtin_amdgcn_rsq(v);
    double g = v * y, h = 0.5 * y;
    const double rr = fma(-h, g, 0.5);
    g = fma(g, rr, g); h = fma(h, rr, h);
    const double R = fma(fma(-g, g, v), h, g);
    double s, c;
    sincos_2pi_unit(u2, &s, &c);                                // = sin/cos(2π u2)
    *z0 = R * c; *z1 = R * s;
}

}  // namespace mpopis
