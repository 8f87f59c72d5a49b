// kernels_linalg.hip -- the small dense FP64 linear algebra the reference gets from LAPACK via
// Distributions/PDMats/StatsBase, restated as device kernels (one workgroup per trial slot):
//   MvNormal(Σ') -> PDMat -> cholesky(Σ')            src/mppi_mpopi_policies.jl:192,307,352,447,551-553,650,723,796
//   invcov(P) as used in the control cost             :194,309,353,449,555,651,725,798 (only the row γ U_orig' Σ⁻¹ is needed)
//   mean(elite, dims=2) / mean of resampled columns / CMA δw (gather + mean)  :465,:807,:573-576
// (the covariance contractions live in kernels_mfma.hip)
#include "engine.h"

namespace mpopis {

constexpr int kNB = 16;

// ---------------------------------------------------------------------------------------------
// Blocked right-looking Cholesky, one workgroup per matrix, panel width 16 (k_potrf_lds: working copy in LDS, cs <= 128, all
// 1-car configs; k_potrf_global: working copy = the output buffer, L2 resident, cs = 300 for 3 cars).  Per panel:
//   (1) one wave factors the 16x16 diagonal block AND inverts it (diag16_factor_inv: 4x4 sub-blocks, one LDS exchange each),
//   (2) the panel below it is L21 = A21 * L11^-T on the matrix cores (v_mfma_f64_16x16x4, one 16-row tile per wave),
//   (3) the rank-16 trailing update runs on the matrix cores too, 4 MFMAs per 16x16 tile, read-modify-write; the wave that owns
//       the next diagonal block updates it first and factors it while the others finish the update (look-ahead).
// scale[b] (nullable) multiplies A first (CMA: MvNormal(σ²Σ), :551).  On a non-positive pivot
// status[b] = MPOPIS_ERR_NOT_PD and active[b] = 0 (the reference throws PosDefException).
// ---------------------------------------------------------------------------------------------
typedef double v4f64_l __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double bcast_lane(double v, int src) {     // src is a compile-time constant after unrolling
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// 16x16 diagonal block: Cholesky factor (written through `store`) and its inverse (into sh.Li), by ONE wave; lane = (row i, column
// group g), entries of columns 4g .. 4g+3, loaded through `load(i, c)` (identity beyond the matrix edge).  Processed as a 4 x 4
// grid of 4x4 sub-blocks with ONE LDS exchange per sub-block column G (a lone wave issues a dependent instruction only every ~8
// cycles and an LDS round trip costs ~100+: a per-pivot formulation took 900 cycles per pivot).  Per G: every lane gathers the
// diagonal 4x4 sub-block (v_readlane), factors and inverts it redundantly in registers, the lanes of column group G turn their
// row into L (rows below: x T', T = L4^-1), the L column block and the R rows of group G go through LDS, and every lane updates its
// trailing entries and its rows of R (= the rows of L11^-1 in the making: right-looking forward substitution on the identity).
// Returns true on a non-positive pivot.
struct DiagScratch { double Lc[2][kNB][5], Rr[2][4][kNB + 1], Li[kNB][kNB + 1]; };
template <class LoadF, class StoreF>
__device__ __forceinline__ bool diag16_factor_inv(int lane, LoadF load, StoreF store, DiagScratch& sh) {
    auto& Lc = sh.Lc; auto& Rr = sh.Rr; auto& Li = sh.Li;
    {
        const int i = lane & 15, g = lane >> 4;
        double e[4], mi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { e[q] = load(i, 4 * g + q); mi[q] = (i == 4 * g + q) ? 1.0 : 0.0; }
        bool bad = false;
        auto rsqrt_full = [&](double piv) {                     // v_rsq_f64 seed + 2 Newton steps (full precision)
            if (!(piv > 0.0)) bad = true;
            double rs = __builtin_amdgcn_rsq(piv);
            rs = rs * fma(-0.5 * piv * rs, rs, 1.5);
            return rs * fma(-0.5 * piv * rs, rs, 1.5);
        };
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            const int buf = G & 1;
            // (a) the 4x4 diagonal sub-block (lower part), from lanes (4G + r, G)
            const double s00 = bcast_lane(e[0], 4 * G + 0 + 16 * G);
            const double s10 = bcast_lane(e[0], 4 * G + 1 + 16 * G), s11 = bcast_lane(e[1], 4 * G + 1 + 16 * G);
            const double s20 = bcast_lane(e[0], 4 * G + 2 + 16 * G), s21 = bcast_lane(e[1], 4 * G + 2 + 16 * G), s22 = bcast_lane(e[2], 4 * G + 2 + 16 * G);
            const double s30 = bcast_lane(e[0], 4 * G + 3 + 16 * G), s31 = bcast_lane(e[1], 4 * G + 3 + 16 * G), s32 = bcast_lane(e[2], 4 * G + 3 + 16 * G),
                         s33 = bcast_lane(e[3], 4 * G + 3 + 16 * G);
            // Cholesky of the sub-block and its inverse T (both lower triangular), in registers
            const double r0 = rsqrt_full(s00), l00 = s00 * r0, l10 = s10 * r0, l20 = s20 * r0, l30 = s30 * r0;
            const double d1 = fma(-l10, l10, s11), r1 = rsqrt_full(d1), l11 = d1 * r1;
            const double l21 = fma(-l20, l10, s21) * r1, l31 = fma(-l30, l10, s31) * r1;
            const double d2 = fma(-l21, l21, fma(-l20, l20, s22)), r2 = rsqrt_full(d2), l22 = d2 * r2;
            const double l32 = fma(-l31, l21, fma(-l30, l20, s32)) * r2;
            const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, s33))), r3 = rsqrt_full(d3), l33 = d3 * r3;
            const double t00 = r0, t11 = r1, t22 = r2, t33 = r3;
            const double t10 = -(l10 * t00) * r1, t21 = -(l21 * t11) * r2, t32 = -(l32 * t22) * r3;
            const double t20 = -fma(l21, t10, l20 * t00) * r2, t31 = -fma(l32, t21, l31 * t11) * r3;
            const double t30 = -fma(l32, t20, fma(l31, t10, l30 * t00)) * r3;
            // (b) column block G of L: the sub-block rows take L4, rows below x T' (x = the row's four entries), rows above 0
            if (g == G) {
                const int r = i - 4 * G;
                double n0, n1, n2, n3;
                if (r < 0) { n0 = n1 = n2 = n3 = 0.0; }
                else if (r == 0) { n0 = l00; n1 = n2 = n3 = 0.0; }
                else if (r == 1) { n0 = l10; n1 = l11; n2 = n3 = 0.0; }
                else if (r == 2) { n0 = l20; n1 = l21; n2 = l22; n3 = 0.0; }
                else if (r == 3) { n0 = l30; n1 = l31; n2 = l32; n3 = l33; }
                else {
                    n0 = e[0] * t00;
                    n1 = fma(e[1], t11, e[0] * t10);
                    n2 = fma(e[2], t22, fma(e[1], t21, e[0] * t20));
                    n3 = fma(e[3], t33, fma(e[2], t32, fma(e[1], t31, e[0] * t30)));
                }
                e[0] = n0; e[1] = n1; e[2] = n2; e[3] = n3;
                Lc[buf][i][0] = n0; Lc[buf][i][1] = n1; Lc[buf][i][2] = n2; Lc[buf][i][3] = n3;
            }
            if ((i >> 2) == G) {                                // R rows of group G, all column groups
#pragma unroll
                for (int q = 0; q < 4; ++q) Rr[buf][i & 3][4 * g + q] = mi[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // rows 4G..4G+3 of L11^-1 restricted to this lane's columns: M[k][q] = sum_{k' <= k} T[k][k'] R[4G+k'][4g+q]
            double M0[4], M1[4], M2[4], M3[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double a0 = Rr[buf][0][4 * g + q], a1 = Rr[buf][1][4 * g + q], a2 = Rr[buf][2][4 * g + q], a3 = Rr[buf][3][4 * g + q];
                M0[q] = t00 * a0;
                M1[q] = fma(t11, a1, t10 * a0);
                M2[q] = fma(t22, a2, fma(t21, a1, t20 * a0));
                M3[q] = fma(t33, a3, fma(t32, a2, fma(t31, a1, t30 * a0)));
            }
            const double li0 = Lc[buf][i][0], li1 = Lc[buf][i][1], li2 = Lc[buf][i][2], li3 = Lc[buf][i][3];   // own row of the L column block
            if (g > G) {                                        // trailing entries: a_ic -= sum_k l_ik l_ck   (only i >= c is ever read)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 4 * g + q;
                    e[q] = fma(-li3, Lc[buf][c][3], fma(-li2, Lc[buf][c][2], fma(-li1, Lc[buf][c][1], fma(-li0, Lc[buf][c][0], e[q]))));
                }
            }
            if ((i >> 2) == G) {                                // these rows of L11^-1 are final
                const int r = i & 3;
#pragma unroll
                for (int q = 0; q < 4; ++q) mi[q] = (r == 0) ? M0[q] : ((r == 1) ? M1[q] : ((r == 2) ? M2[q] : M3[q]));
            } else if ((i >> 2) > G) {                          // R[i,:] -= sum_k l_ik M[k,:]
#pragma unroll
                for (int q = 0; q < 4; ++q) mi[q] = fma(-li3, M3[q], fma(-li2, M2[q], fma(-li1, M1[q], fma(-li0, M0[q], mi[q]))));
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * g + q;
            if (c <= i) store(i, c, e[q]);
            Li[i][c] = (c <= i) ? mi[q] : 0.0;
        }
        return bad;
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-resident Cholesky for n <= 128 (every 1-car configuration), one workgroup of 8 waves per matrix.  Same blocked
// right-looking scheme as k_potrf, but the two serial pieces of a panel -- which bound it: 7 panels x (diagonal block 3.9 us +
// panel solve 2.9 us) of 63 us at n = 100 -- are reorganised:
//   * diagonal 16x16 block: lane = (row i, column group g), 4 entries per lane, processed as 4x4 sub-blocks with one LDS
//     exchange per sub-block column (instead of 15 v_readlane broadcasts per pivot), and the SAME loop carries a second 16x16
//     block along that ends up as L11^-1 (right-looking forward substitution on the identity);
//   * panel solve L21 = A21 L11^-T becomes one 16x16x16 product per row tile on the matrix cores (B operand = L11^-1),
//     instead of a 136-term dependent substitution per row fed by LDS reads.
// Using the explicit inverse of the (well-conditioned) 16x16 diagonal block costs ~cond(L11) eps in L21, far inside the
// parity budget (|L L' - A| stays at 1e-16 |A| in tools/kbench_linalg).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_potrf_lds(const double* __restrict__ A, size_t Astride, double* __restrict__ Lout,
                                                   int n, int npad, const double* scale, int* status, int* active) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int failed;
    __shared__ DiagScratch dsh;
    constexpr int NTHR = 512, NW = NTHR / 64;
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double* Ab = A + (size_t)b * Astride;
    double* Lb = Lout + (size_t)b * n * n;
    const double sc = scale ? scale[b] : 1.0;
    double* W = smem;                                  // [npad][npad] column-major, identity tail beyond n
    const int ldw = npad, m = npad;
    if (tid == 0) failed = 0;
    {   // copy in the lower triangle only (see k_potrf): column pair c (columns c and m-1-c) holds m+1 triangle entries
        const int npairs2 = m / 2, tchunks = (m + 1 + 63) / 64;
        for (int c0 = wv * 4; c0 < npairs2; c0 += NW * 4) {
            for (int tc = 0; tc < tchunks; tc += 2) {
                double av[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = min(c0 + (u >> 1), npairs2 - 1), t = min((tc + (u & 1)) * 64 + lane, m);
                    const int j = (t < m - c) ? c : m - 1 - c, i = (t < m - c) ? c + t : m - 1 - c + (t - (m - c));
                    av[u] = Ab[(size_t)min(i, n - 1) + (size_t)min(j, n - 1) * n];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + (u >> 1), t = (tc + (u & 1)) * 64 + lane;
                    if (c < npairs2 && t <= m && tc + (u & 1) < tchunks) {
                        const int j = (t < m - c) ? c : m - 1 - c, i = (t < m - c) ? c + t : m - 1 - c + (t - (m - c));
                        W[(size_t)i + (size_t)j * ldw] = (i < n && j < n) ? sc * av[u] : ((i == j) ? 1.0 : 0.0);
                    }
                }
            }
        }
    }
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;
    auto factor_diag_inv = [&](int j0) {
        const bool bad = diag16_factor_inv(lane,
            [&](int i, int c) { return W[(size_t)(j0 + i) + (size_t)(j0 + c) * ldw]; },
            [&](int i, int c, double v) { W[(size_t)(j0 + i) + (size_t)(j0 + c) * ldw] = v; }, dsh);
        if (bad && lane == 0) failed = 1;
    };
    // rows r0 .. r0+15 of the panel below the diagonal block: L21 tile = A21 tile * L11^-T on the matrix cores
    auto panel_tile = [&](int j0, int r0) {
        v4f64_l acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const double av = W[(size_t)(r0 + li) + (size_t)(j0 + kk * 4 + lk) * ldw];
            const double bv = dsh.Li[li][kk * 4 + lk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc, 0, 0, 0);       // acc[r] = sum_k Linv[lk+4r][k] A21[r0+li][k]
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                                           // every lane has read the tile: overwrite in place
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int r = 0; r < 4; ++r) W[(size_t)(r0 + li) + (size_t)(j0 + lk + 4 * r) * ldw] = acc[r];
    };
    auto trail_pair = [&](int j0, int t1, int q) {
        int ta = 0, qq = q;
        while (qq >= ta + 1) { qq -= ta + 1; ++ta; }
        const int r0 = (t1 + ta) * 16, c0 = (t1 + qq) * 16;
        v4f64_l acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int col = j0 + kk * 4 + lk;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(W[(size_t)(c0 + li) + (size_t)col * ldw], W[(size_t)(r0 + li) + (size_t)col * ldw], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = r0 + li, c = c0 + lk + 4 * r;
            if (i >= c) W[(size_t)i + (size_t)c * ldw] -= acc[r];
        }
    };
    if (wv == 0) factor_diag_inv(0);
    __syncthreads();
    for (int j0 = 0; j0 < m; j0 += kNB) {
        if (failed) break;
        const int i1 = j0 + kNB, ntile = (m - i1) / 16, t1 = i1 / 16;
        for (int t = wv; t < ntile; t += NW) panel_tile(j0, i1 + 16 * t);
        __syncthreads();
        if (ntile > 0) {
            const int npair = ntile * (ntile + 1) / 2;
            if (wv == 0) { trail_pair(j0, t1, 0); factor_diag_inv(i1); }         // look-ahead: next diagonal block while the others update
            else for (int q = wv; q < npair; q += NW - 1) trail_pair(j0, t1, q);
        }
        __syncthreads();
    }
    if (failed) {
        if (tid == 0) { if (status) atomicMin(&status[b], MPOPIS_ERR_NOT_PD); if (active) active[b] = 0; }
        return;
    }
    for (int j = wv; j < n; j += NW)
        for (int i = lane; i < n; i += 64) Lb[(size_t)i + (size_t)j * n] = (i >= j) ? W[(size_t)i + (size_t)j * ldw] : 0.0;
}

// Global-memory variant for matrices that do not fit in LDS (cs = 300: three cars), one workgroup of 16 waves: the working
// matrix is the output buffer (L2 resident); the solved panel strip P[m][17] -- both operands of the rank-16 trailing update --
// lives in LDS, only the trailing read-modify-write goes to L2.  Diagonal blocks and panel solves as in k_potrf_lds.
__global__ void __launch_bounds__(1024) k_potrf_global(const double* __restrict__ A, size_t Astride, double* __restrict__ Lout,
                                                       int n, const double* scale, int* status, int* active) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int failed;
    __shared__ DiagScratch dsh;
    constexpr int NTHR = 1024, NW = NTHR / 64, kPS = kNB + 1;
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const double* Ab = A + (size_t)b * Astride;
    double* W = Lout + (size_t)b * n * n;
    const double sc = scale ? scale[b] : 1.0;
    const int ldw = n, m = n;
    double* P = smem;                                   // [m][17] solved panel strip
    if (tid == 0) failed = 0;
    for (int e0 = tid; e0 < m * m; e0 += NTHR * 8) {    // W = lower(sc * A), zeros above the diagonal
        double av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = min(e0 + u * NTHR, m * m - 1); av[u] = Ab[(size_t)(e % m) + (size_t)(e / m) * n]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * NTHR;
            if (e < m * m) { const int i = e % m, j = e / m; W[(size_t)i + (size_t)j * ldw] = (i >= j) ? sc * av[u] : 0.0; }
        }
    }
    __threadfence_block();
    __syncthreads();
    auto factor_diag_inv = [&](int j0) {
        const bool bad = diag16_factor_inv(lane,
            [&](int i, int c) { return (j0 + i < m && j0 + c < m) ? W[(size_t)(j0 + i) + (size_t)(j0 + c) * ldw] : ((i == c) ? 1.0 : 0.0); },
            [&](int i, int c, double v) { if (j0 + i < m && j0 + c < m) W[(size_t)(j0 + i) + (size_t)(j0 + c) * ldw] = v; }, dsh);
        if (bad && lane == 0) failed = 1;
    };
    auto panel_tile = [&](int j0, int r0) {              // L21 tile = A21 tile * L11^-T on the matrix cores; result to W (L2) and P (LDS)
        const int ra = r0 + li;
        v4f64_l acc = {0.0, 0.0, 0.0, 0.0};
        double av[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { const int col = j0 + kk * 4 + lk; av[kk] = (ra < m && col < m) ? W[(size_t)ra + (size_t)col * ldw] : 0.0; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(dsh.Li[li][kk * 4 + lk], av[kk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = lk + 4 * r;
            if (ra < m) { P[(size_t)ra * kPS + c] = acc[r]; if (j0 + c < m) W[(size_t)ra + (size_t)(j0 + c) * ldw] = acc[r]; }
        }
    };
    auto trail_pair = [&](int j0, int t1, int q) {       // one 16x16 tile of the rank-16 trailing update: pair q -> tile (t1 + ta, t1 + tb), ta >= tb
        int ta = 0, qq = q;
        while (qq >= ta + 1) { qq -= ta + 1; ++ta; }
        const int r0 = (t1 + ta) * 16, c0 = (t1 + qq) * 16;
        v4f64_l acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ra = r0 + li, rb = c0 + li;
            const double av = (ra < m) ? P[(size_t)ra * kPS + kk * 4 + lk] : 0.0;
            const double bv = (rb < m) ? P[(size_t)rb * kPS + kk * 4 + lk] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc, 0, 0, 0);     // transposed tile: lanes run along i
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = r0 + li, c = c0 + lk + 4 * r;
            if (i < m && c < m && i >= c) W[(size_t)i + (size_t)c * ldw] -= acc[r];
        }
    };
    if (wv == 0) factor_diag_inv(0);
    __threadfence_block();
    __syncthreads();
    for (int j0 = 0; j0 < m; j0 += kNB) {
        if (failed) break;
        const int i1 = j0 + kNB;                          // first row below the panel (>= m after the last, possibly partial, panel)
        const int ntile = (i1 < m) ? (m - i1 + 15) / 16 : 0, t1 = i1 / 16;
        for (int t = wv; t < ntile; t += NW) panel_tile(j0, i1 + 16 * t);
        __threadfence_block();
        __syncthreads();
        if (ntile > 0) {
            const int npair = ntile * (ntile + 1) / 2;
            if (wv == 0) { trail_pair(j0, t1, 0); __threadfence_block(); factor_diag_inv(i1); }   // look-ahead: next diagonal block first
            else for (int q = wv; q < npair; q += NW - 1) trail_pair(j0, t1, q);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (failed && tid == 0) { if (status) atomicMin(&status[b], MPOPIS_ERR_NOT_PD); if (active) active[b] = 0; }
}

void launch_potrf(const double* A, size_t Astride, double* L, int B, int n, const double* scale, int* status, int* active, hipStream_t s) {
    const int npad = (n + kNB - 1) / kNB * kNB;
    const size_t bytes = (size_t)npad * npad * sizeof(double);
    if (bytes <= 150 * 1024) {
        static std::atomic<unsigned long long> seen{0};
        ensure_dyn_lds((const void*)k_potrf_lds, 150 * 1024, seen);
        hipLaunchKernelGGL(k_potrf_lds, dim3(B), dim3(512), bytes, s, A, Astride, L, n, npad, scale, status, active);
    } else {
        const size_t strip = (size_t)n * (kNB + 1) * sizeof(double);                         // panel strip
        static std::atomic<unsigned long long> seen2{0};
        ensure_dyn_lds((const void*)k_potrf_global, 150 * 1024, seen2);
        hipLaunchKernelGGL(k_potrf_global, dim3(B), dim3(1024), strip, s, A, Astride, L, n, scale, status, active);
    }
}

// g = Σ⁻¹ (γ U_orig) through the Cholesky factor (Σ symmetric => row vector γ U_orig' Σ⁻¹ = g').
// Slow path: only taken when α != 1 (γ != 0); no BASELINE config uses it.
__global__ void __launch_bounds__(256) k_chol_solve_gvec(const double* __restrict__ L, size_t Lstride, const double* __restrict__ Uorig,
                                                         double gamma, double* __restrict__ g, int n, const int* active) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) double y[];
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const double* Lb = L + (size_t)b * Lstride;
    for (int i = threadIdx.x; i < n; i += 256) y[i] = gamma * Uorig[(size_t)b * n + i];
    __syncthreads();
    for (int j = 0; j < n; ++j) {                     // forward: L y = γU
        if (threadIdx.x == 0) y[j] = y[j] / Lb[(size_t)j + (size_t)j * n];
        __syncthreads();
        const double yj = y[j];
        for (int i = j + 1 + threadIdx.x; i < n; i += 256) y[i] = fma(-Lb[(size_t)i + (size_t)j * n], yj, y[i]);
        __syncthreads();
    }
    for (int j = n - 1; j >= 0; --j) {                // backward: L' g = y
        if (threadIdx.x == 0) y[j] = y[j] / Lb[(size_t)j + (size_t)j * n];
        __syncthreads();
        const double yj = y[j];
        for (int i = threadIdx.x; i < j; i += 256) y[i] = fma(-Lb[(size_t)j + (size_t)i * n], yj, y[i]);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += 256) g[(size_t)b * n + i] = y[i];
}
void launch_chol_solve_gvec(const double* L, size_t Lstride, const double* Uorig, double gamma, double* g, int B, int n, const int* active, hipStream_t s) {
    hipLaunchKernelGGL(k_chol_solve_gvec, dim3(B), dim3(256), n * sizeof(double), s, L, Lstride, Uorig, gamma, g, n, active);
}

// Level-1 entry: caller supplies Σ_inv; g[j] = Σ_i (γ U_orig[i]) Σ_inv[i][j]   (:272)
__global__ void __launch_bounds__(256) k_gvec_from_inv(const double* __restrict__ Sinv, const double* __restrict__ Uorig, double gamma,
                                                       double* __restrict__ g, int n) {
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double v = 0.0;
    for (int i = 0; i < n; ++i) v = fma(gamma * Uorig[(size_t)b * n + i], Sinv[(size_t)i + (size_t)j * n], v);
    g[(size_t)b * n + j] = v;
}
void launch_gvec_from_inv(const double* Sinv, const double* Uorig, double gamma, double* g, int B, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_gvec_from_inv, dim3((n + 255) / 256, B), dim3(256), 0, s, Sinv, Uorig, gamma, g, n);
}

// mean over gathered columns: mu[r] = (1/m) Σ_j X[r][idx[j]] ; optionally weighted by cw[j] (CMA δw, no division)
__global__ void __launch_bounds__(256) k_gather_mean(const double* __restrict__ X, const int32_t* __restrict__ idx, const double* __restrict__ cw,
                                                     double* __restrict__ mu, size_t mu_stride, int cs, int K, int m, int divide, const int* active) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.y, r = blockIdx.x;
    if (active && !active[b]) return;
    __shared__ double sh[4];
    const double* x = X + ((size_t)b * cs + r) * K;
    const int32_t* ib = idx + (size_t)b * K;          // idx arrays are K long per slot (order / resample)
    double acc = 0.0;
    for (int j = threadIdx.x; j < m; j += 256) acc = cw ? fma(cw[j], x[ib[j]], acc) : acc + x[ib[j]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double t = sh[0] + sh[1] + sh[2] + sh[3]; mu[(size_t)b * mu_stride + r] = divide ? t / m : t; }
}
void launch_gather_mean(const double* X, const int32_t* idx, const double* cw, double* mu, int B, int cs, int K, int m, int divide,
                        const int* active, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_mean, dim3(cs, B), dim3(256), 0, s, X, idx, cw, mu, (size_t)cs, cs, K, m, divide, active);
}
void launch_gather_mean_strided(const double* X, const int32_t* idx, const double* cw, double* out, size_t out_stride, int B, int cs, int K, int m,
                                const int* active, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_mean, dim3(cs, B), dim3(256), 0, s, X, idx, cw, out, out_stride, cs, K, m, 0, active);
}

}  // namespace mpopis
