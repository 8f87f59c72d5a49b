// linalg_diag.h -- the 16x16 diagonal-block kernel of the blocked Cholesky factorisations (kernels_linalg.hip): factor + inverse by one wave.
#pragma once
#include <hip/hip_runtime.h>

namespace mpopis {

constexpr int kNB = 16;
typedef double v4f64_l __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double bcast_lane(double v, int src) {     // src is a compile-time constant after unrolling
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// 16x16 diagonal block of a blocked Cholesky, factored by ONE wave.  lane = (row i, column group g) holds the entries of columns
// 4g .. 4g+3, loaded through `load(i, c)` (identity beyond the matrix edge; only i >= c is used).  A lone wave issues an FP64
// instruction only every ~8 cycles whatever its dependencies, so the cost is the instruction count: the block is processed as a
// 4 x 4 grid of 4x4 sub-blocks with ONE LDS exchange per sub-block column G.  Per G: every lane gathers the diagonal 4x4
// sub-block (v_readlane), factors it and inverts it redundantly in registers (rsqrt: v_rsq_f64 + one Newton step, <= 2 ulp --
// the error rescales the whole column consistently, |L L' - A| stays at rounding level), the lanes of column group G turn their
// row into L (rows below: x T', T = L4^-1), the L column block goes through LDS and every lane updates its trailing entries.
// Out: the factor through `store(i, c, v)` (c <= i) and, for the panel solve that follows (panel_solve_tile), sh.L = the factor
// with zeros above the diagonal and sh.T[G] = the inverses of the four diagonal 4x4 sub-blocks.  Returns true on a non-positive pivot.
struct DiagScratch { double Lc[2][kNB][5], T[4][4][4], L[kNB][kNB + 1]; };
// rounds (1..4, wave-uniform): sub-block columns to process.  A LAST diagonal block whose rows beyond 4 rounds are the identity tail of the padding
// (n = 100: the seventh block holds rows 96..99 and twelve identity rows) needs only its leading rounds -- the identity rows are already their own
// factor, nothing below the block consumes sh.T -- which takes ~0.5 us per skipped round off the serial chain.
template <class LoadF, class StoreF>
__device__ __forceinline__ bool diag16_factor(int lane, LoadF load, StoreF store, DiagScratch& sh, int rounds = 4) {
    auto& Lc = sh.Lc;
    const int i = lane & 15, g = lane >> 4;
    double e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) e[q] = load(i, 4 * g + q);
    bool bad = false;
    auto rsqrt1 = [&](double piv) {
        if (!(piv > 0.0)) bad = true;
        const double rs = __builtin_amdgcn_rsq(piv);
        return rs * fma((-0.5 * piv) * rs, rs, 1.5);
    };
#pragma unroll
    for (int G = 0; G < 4; ++G) {
        if (G >= rounds) break;
        const int buf = G & 1;
        // (a) the 4x4 diagonal sub-block (lower part), from lanes (4G + r, G)
        const double s00 = bcast_lane(e[0], 4 * G + 0 + 16 * G);
        const double s10 = bcast_lane(e[0], 4 * G + 1 + 16 * G), s11 = bcast_lane(e[1], 4 * G + 1 + 16 * G);
        const double s20 = bcast_lane(e[0], 4 * G + 2 + 16 * G), s21 = bcast_lane(e[1], 4 * G + 2 + 16 * G), s22 = bcast_lane(e[2], 4 * G + 2 + 16 * G);
        const double s30 = bcast_lane(e[0], 4 * G + 3 + 16 * G), s31 = bcast_lane(e[1], 4 * G + 3 + 16 * G), s32 = bcast_lane(e[2], 4 * G + 3 + 16 * G),
                     s33 = bcast_lane(e[3], 4 * G + 3 + 16 * G);
        // Cholesky of the sub-block and its inverse T (both lower triangular), in registers
        const double r0 = rsqrt1(s00), l00 = s00 * r0, l10 = s10 * r0, l20 = s20 * r0, l30 = s30 * r0;
        const double d1 = fma(-l10, l10, s11), r1 = rsqrt1(d1), l11 = d1 * r1;
        const double l21 = fma(-l20, l10, s21) * r1, l31 = fma(-l30, l10, s31) * r1;
        const double d2 = fma(-l21, l21, fma(-l20, l20, s22)), r2 = rsqrt1(d2), l22 = d2 * r2;
        const double l32 = fma(-l31, l21, fma(-l30, l20, s32)) * r2;
        const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, s33))), r3 = rsqrt1(d3), l33 = d3 * r3;
        const double t00 = r0, t11 = r1, t22 = r2, t33 = r3;
        const double t10 = -(l10 * t00) * r1, t21 = -(l21 * t11) * r2, t32 = -(l32 * t22) * r3;
        const double t20 = -fma(l21, t10, l20 * t00) * r2, t31 = -fma(l32, t21, l31 * t11) * r3;
        const double t30 = -fma(l32, t20, fma(l31, t10, l30 * t00)) * r3;
        if (lane == 0) {                                        // wave-uniform values: one lane stores T_G (zeros above its diagonal)
            double* Tg = &sh.T[G][0][0];
            Tg[0] = t00; Tg[1] = 0.0; Tg[2] = 0.0; Tg[3] = 0.0;
            Tg[4] = t10; Tg[5] = t11; Tg[6] = 0.0; Tg[7] = 0.0;
            Tg[8] = t20; Tg[9] = t21; Tg[10] = t22; Tg[11] = 0.0;
            Tg[12] = t30; Tg[13] = t31; Tg[14] = t32; Tg[15] = t33;
        }
        // (b) column block G of L: the sub-block rows take L4, rows below x T' (x = the row's four entries), rows above 0 --
        // branch-free: one-hot row weights (a 6-way branch costs more scalar bookkeeping than the 14 extra FMAs)
        if (g == G) {
            const int r = i - 4 * G;
            const double w0 = (r == 0) ? 1.0 : 0.0, w1 = (r == 1) ? 1.0 : 0.0, w2 = (r == 2) ? 1.0 : 0.0, w3 = (r == 3) ? 1.0 : 0.0,
                         wb = (r > 3) ? 1.0 : 0.0;
            const double x0 = wb * e[0], x1 = wb * e[1], x2 = wb * e[2], x3 = wb * e[3];
            const double n0 = fma(w3, l30, fma(w2, l20, fma(w1, l10, fma(w0, l00, x0 * t00))));
            const double n1 = fma(w3, l31, fma(w2, l21, fma(w1, l11, fma(x1, t11, x0 * t10))));
            const double n2 = fma(w3, l32, fma(w2, l22, fma(x2, t22, fma(x1, t21, x0 * t20))));
            const double n3 = fma(w3, l33, fma(x3, t33, fma(x2, t32, fma(x1, t31, x0 * t30))));
            e[0] = n0; e[1] = n1; e[2] = n2; e[3] = n3;
            if (G < 3) { Lc[buf][i][0] = n0; Lc[buf][i][1] = n1; Lc[buf][i][2] = n2; Lc[buf][i][3] = n3; }
        }
        if (G < 3) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (g > G) {                                        // trailing entries: a_ic -= sum_k l_ik l_ck   (only i >= c is ever read)
                const double li0 = Lc[buf][i][0], li1 = Lc[buf][i][1], li2 = Lc[buf][i][2], li3 = Lc[buf][i][3];   // own row of the L column block
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 4 * g + q;
                    e[q] = fma(-li3, Lc[buf][c][3], fma(-li2, Lc[buf][c][2], fma(-li1, Lc[buf][c][1], fma(-li0, Lc[buf][c][0], e[q]))));
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = 4 * g + q;
        if (c <= i) store(i, c, e[q]);
        sh.L[i][c] = (c <= i) ? e[q] : 0.0;
    }
    return bad;
}

// Panel solve behind diag16_factor: X = A21_tile * L11^-T for one 16-row tile, by one wave, as a blocked forward substitution on
// the matrix cores.  a[r] on lane (li, lk) = A21[row li][column 4r + lk] (the accumulator layout of v_mfma_f64_16x16x4 when
// the tile rows run along the lanes); on return a[r] = X[li][4r + lk].  Per sub-block column G: X_G = A'_G T_G' (one MFMA against
// the padded 4x4 inverse), then A' -= X_G L[:, 4G..4G+3]' for the columns to the right (one MFMA).  ops.t[G] / ops.l[G]: the
// tile-independent first operands, loaded once per wave and panel by panel_solve_operands.
typedef double v4f64_d __attribute__((ext_vector_type(4)));
struct PanelOps { double t[4], l[3]; };
__device__ __forceinline__ PanelOps panel_solve_operands(int lane, const DiagScratch& sh) {
    const int li = lane & 15, lk = lane >> 4;
    PanelOps o;
#pragma unroll
    for (int G = 0; G < 4; ++G) o.t[G] = (li < 4) ? sh.T[G][li][lk] : 0.0;
#pragma unroll
    for (int G = 0; G < 3; ++G) o.l[G] = -sh.L[li][4 * G + lk];
    return o;
}
__device__ __forceinline__ void panel_solve_tile(const PanelOps& o, double (&a)[4]) {
    v4f64_d acc = {a[0], a[1], a[2], a[3]};
    const v4f64_d zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int G = 0; G < 4; ++G) {
        const v4f64_d res = __builtin_amdgcn_mfma_f64_16x16x4f64(o.t[G], acc[G], zero, 0, 0, 0);   // res[0] on lane (li, lk) = X[li][4G + lk]
        a[G] = res[0];
        if (G < 3) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(o.l[G], a[G], acc, 0, 0, 0);
    }
}

}  // namespace mpopis
