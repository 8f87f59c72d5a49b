// philox.h -- Philox4x32-10 (Salmon et al., SC'11) + Box-Muller on device; shared by the sampler and the fused
// sampler+unwhiten kernel.  Bit-identical counters/keys to oracle/mpopis_oracle.c (orc_philox_normals).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpopis {

#ifndef MPOPIS_PHILOX_ROUNDS
#define MPOPIS_PHILOX_ROUNDS 10
#endif
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
    for (int r = 0; r < MPOPIS_PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;   // one v_mad_u64_u32 each
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---- Box-Muller transcendentals from small tables (LDS) -------------------------------------------------------------------------------
// The Philox / Box-Muller work of the fused sampler shares the SIMD's FP64 datapath with its MFMAs (kernels_mfma.hip), so every VALU
// instruction counts.  Round 2 evaluated log and sin/cos with fdlibm-style argument reduction + degree-14 polynomials (~40 + ~35 VALU per
// pair); round 3: a 128-entry table for log and a 64-entry rotation table for sin/cos (|r| <= 2^-8 / |x| <= 0.05: ~19 + ~23 VALU per pair);
// round 4: 256 / 128 entries (6 KB of LDS), |r| <= 2^-9 / |x| <= 0.0246, two Horner steps less in log1p and one less in each of sin and cos --
// same accuracy class (every normal within 1e-13 of the libm evaluation, tests/test_gpu_parity.py; the top log interval is exact at u -> 1,
// where sqrt(-2 log u) is most sensitive).  Table: kRngTabLog x {1/c_i, log c_i}, c_i = (1 + (i + 0.5)/kRngTabLog)/2 (last entry: c = 1),
// then kRngTabSc x {sin, cos}(2 pi (j + 0.5)/kRngTabSc); filled once per handle by k_rng_tab_init with the library's correctly rounded functions.
constexpr int kRngTabLog = 256, kRngTabSc = 128, kRngTabDoubles = 2 * (kRngTabLog + kRngTabSc);

// log(u) for a normal-range u in (0, 1)
__device__ __forceinline__ double log_unit(double u, const double* __restrict__ tab) {
    const int k = __builtin_amdgcn_frexp_exp(u);               // u = m 2^k, m in [0.5, 1)
    const double m = __builtin_amdgcn_frexp_mant(u);
    const int i = (__double2hiint(m) >> 12) & 255;             // top 8 fraction bits of m
    const double inv = tab[2 * i], logc = tab[2 * i + 1];
    const double r = fma(m, inv, -1.0);                        // |r| <= 2^-9
    double p = fma(r, 1.0 / 5.0, -1.0 / 4.0);                  // log1p(r) to r^5: the next term is r^6 / 6 <= 9.2e-18 ABSOLUTE -- below half an ulp of |log u| except in the top interval (c = 1, log u = log1p(r) itself), where it is up to ~20 ulp at the far end (|log u| ~ 2e-3); harmless at the 1e-13 the normals are tested to
    p = fma(p, r, 1.0 / 3.0);
    p = fma(p, r, -0.5);
    const double lp = fma(p * r, r, r);                        // log1p(r)
    return fma((double)k, 6.93147180559945286227e-01, logc) + lp;
}

// (sin, cos)(2 pi u) for u = (w + 0.5) 2^-32 straight from one Philox word: sector j = top 7 bits, x = 2 pi (frac - 0.5)/128 (exact), degree-5 /
// degree-6 Taylor kernels on |x| <= 0.0246 (next terms: x^7/5040 <= 1.1e-15 x relative 4.3e-14 ... see below; x^8/40320 <= 3.3e-18) and one
// rotation by the tabulated sector centre.  The sine's truncation is 4.3e-14 RELATIVE TO x, i.e. <= 1.1e-15 absolute: half an ulp of the
// rotated result, whose magnitude is O(1) except within 1e-15 of the axes.
__device__ __forceinline__ void sincos_2pi_u32(uint32_t w, const double* __restrict__ tab, double* sn, double* cs) {
    const int j = w >> 25;
    const double f = fma((double)(w & 0x1ffffffu), 0x1p-25, 0x1p-26 - 0.5);      // 128 u - j - 0.5, exact
    const double x = f * 4.90873852123405193510e-02, z = x * x;                  // 2 pi / 128
    const double ps = fma(z, 1.0 / 120.0, -1.0 / 6.0);
    const double s = fma(z * x, ps, x);
    double pc = fma(z, -1.0 / 720.0, 1.0 / 24.0);
    pc = fma(z, pc, -0.5);
    const double c = fma(z, pc, 1.0);
    const double S = tab[2 * kRngTabLog + 2 * j], C = tab[2 * kRngTabLog + 2 * j + 1];
    *sn = fma(S, c, C * s);
    *cs = fma(C, c, -(S * s));
}

// ---- the normal stream ---------------------------------------------------------------------------------------------------------------------
// One Philox4x32-10 call (counter = q, stream words, key = seed) yields FOUR standard normals, numbers 4q .. 4q+3 of the stream: Box-Muller on
// (w0, w1) and on (w2, w3), radius word first, with 32-bit uniforms u = (w + 0.5) 2^-32 in (0, 1).  Round 2 spent a whole call on one pair
// (two 53-bit uniforms); the twenty v_mad_u64_u32 of a call were the largest single item of the fused sampler, whose VALU work does NOT overlap
// the FP64 MFMAs of its neighbours (tools/mfma_valu_overlap.hip: the two serialise on a SIMD).  What 32 bits cost: the radius sqrt(-2 log u)
// takes 2^32 values and stops at 6.76 = sqrt(-2 log 2^-33) (mass beyond: 1.4e-11 -- one draw in 2700 full C5 steps); neighbouring radii differ by <= 4e-10 relative in
// the bulk.  The reference's stream (MersenneTwister + ziggurat) is not reproduced either way; oracle/mpopis_oracle.c (orc_philox_normals)
// defines the same stream with libm.
__device__ __forceinline__ void box_muller_u32(uint32_t wr, uint32_t wa, const double* __restrict__ tab, double* z0, double* z1) {
    const double u1 = fma((double)wr, 0x1p-32, 0x1p-33);       // exact, < 1
    // sqrt of a positive normal-range number: v_rsq_f64 seed + coupled Newton step + residual correction (1 ulp, see
    // tools/rcp_acc.hip) instead of the library sqrt with its denormal rescaling (8 instead of 18 VALU ops)
    const double v = -2.0 * log_unit(u1, tab);
    const double y = __builtin_amdgcn_rsq(v);
    double g = v * y, h = 0.5 * y;
    const double rr = fma(-h, g, 0.5);
    g = fma(g, rr, g); h = fma(h, rr, h);
    const double R = fma(fma(-g, g, v), h, g);
    double s, c;
    sincos_2pi_u32(wa, tab, &s, &c);
    *z0 = R * c; *z1 = R * s;
}
// normals 4q .. 4q+3
__device__ __forceinline__ void philox_normal_quad(uint64_t seed, uint32_t slo, uint32_t shi, uint64_t q, const double* __restrict__ tab, double* z) {
    uint32_t r[4];
    philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), slo, shi, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    box_muller_u32(r[0], r[1], tab, &z[0], &z[1]);
    box_muller_u32(r[2], r[3], tab, &z[2], &z[3]);
}
// normals 2j, 2j+1 (half a call's output: for consumers that cannot use all four)
__device__ __forceinline__ void philox_normal_pair(uint64_t seed, uint32_t slo, uint32_t shi, uint64_t j, const double* __restrict__ tab, double* z0, double* z1) {
    uint32_t r[4];
    const uint64_t q = j >> 1;
    philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), slo, shi, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const bool hi = j & 1;
    box_muller_u32(hi ? r[2] : r[0], hi ? r[3] : r[1], tab, z0, z1);
}
// the tables live in global memory (one copy per handle, filled by launch_rng_tab_init at creation); kernels stage them into LDS
__device__ __forceinline__ void stage_rng_tab(double* sh_tab, const double* __restrict__ gtab, int tid, int nthreads) {
    for (int i = tid; i < kRngTabDoubles; i += nthreads) sh_tab[i] = gtab[i];
}
void launch_rng_tab_init(double* gtab, hipStream_t s);

}  // namespace mpopis
