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
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void philox_normal_pair(uint64_t seed, uint32_t slo, uint32_t shi, uint64_t j, double* z0, double* z1) {
    uint32_t r[4];
    philox4x32_10((uint32_t)j, (uint32_t)(j >> 32), slo, shi, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const uint64_t a = ((uint64_t)r[1] << 32) | r[0], b = ((uint64_t)r[3] << 32) | r[2];
    const double two_m53 = 1.0 / 9007199254740992.0;
    const double u1 = ((double)(a >> 11) + 0.5) * two_m53;
    const double u2 = ((double)(b >> 11) + 0.5) * two_m53;
    // sqrt of a positive normal-range number: v_rsq_f64 seed + coupled Newton step + residual correction (1 ulp, see
    // tools/rcp_acc.hip) instead of the library sqrt with its denormal rescaling (8 instead of 18 VALU ops)
    const double v = -2.0 * log(u1);
    const double y = __builtin_amdgcn_rsq(v);
    double g = v * y, h = 0.5 * y;
    const double rr = fma(-h, g, 0.5);
    g = fma(g, rr, g); h = fma(h, rr, h);
    const double R = fma(fma(-g, g, v), h, g);
    double s, c;
    sincospi(2.0 * u2, &s, &c);                                 // = sin/cos(2π u2), no Payne-Hanek reduction
    *z0 = R * c; *z1 = R * s;
}

}  // namespace mpopis
