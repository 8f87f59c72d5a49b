// kernels_ce.hip -- the CE-MPPI proposal update for small elite sets in ONE launch:
//     elite = E[:, order[1:m]];  μ′ = mean(elite, dims = 2);  Σ′ = cov(Σ_est, elite') + 10e-9 I;  pol.U += μ′
// (src/mppi_mpopi_policies.jl:463-466; Σ_est = SimpleCovariance() | LinearShrinkage(DiagonalUnequalVariance(), :ss | :lw) |
//  LinearShrinkage(DiagonalCommonVariance(), :rblw | :oas), :414-426).
// The general path (engine_ais.hip) spends seven launches on this -- gather-mean, scatter partial + finish (twice for :ss / :lw: second and fourth
// moments), standard deviations, shrinkage, mean add -- which is right for K = 4096 (m = 819 columns of 100 rows go through the matrix cores) and
// wrong for the harness default K = 150 (m = 30: 3000 numbers per trial): there each launch is a ~8 us latency stub and the "moments" class was 59 us of
// a 247 us AIS iteration.  Here one workgroup per trial keeps the centred elite matrix X in LDS and forms S = X X'/m and -- for the
// Schäfer-Strimmer / Ledoit-Wolf intensity -- Q = (X∘X)(X∘X)' on the matrix cores (on standardised data Σ_j z_a² z_b² = Q_ab / (S_aa S_bb)),
// keeps both in registers, reduces the intensity, shrinks and writes Σ′.  Same formulas as k_ss_shrink / k_common_shrink (kernels_mfma.hip) and
// the oracle's cov_*_cols.
#include "engine.h"

namespace mpopis {

constexpr int kCeThreads = 512, kCeWaves = kCeThreads / 64, kCeMaxPairs = 5;      // 8 waves x 5 tile pairs >= 36 = pairs of 8 row tiles (cs <= 128)
typedef double v4f64_ce __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ce_decode(int q, int* ta, int* tb) {    // q -> (ta >= tb), row-major lower enumeration
    int a = 0;
    while (q >= a + 1) { q -= a + 1; ++a; }
    *ta = a; *tb = q;
}

// S = X X' / m and Q = (X.X)(X.X)' on the matrix cores (X = centred elite matrix in LDS, zero padded to 16-row tiles and 4-column k-steps):
// __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc) with av = X[ta 16 + li][4kk + lk], bv = X[tb 16 + li][4kk + lk] accumulates
// acc[r] = (tile ta)(tile tb)' at (row li, column lk + 4r) -- the idiom of kernels_linalg.hip / kernels_mfma.hip.
// cost != nullptr (K <= kCeSortMax): the kernel also does what launch_sortperm would have done before it -- order = sortperm(cost) as a rank sort
// ((cost, index) order == Julia's stable sortperm; same code as k_sortperm_rank) and the elite early break (:458-461: max_j |c[order[j+1]] -
// c[order[j]]| < 10e-3 over the elite set => active[b] = 0 and nothing of this iteration is applied) -- one launch less per AIS iteration.
constexpr int kCeSortMax = 256;
__global__ void __launch_bounds__(kCeThreads) k_ce_cov_small(const double* __restrict__ E, int32_t* __restrict__ order, double* __restrict__ mu_out,
                                                             double* __restrict__ Sg, double* __restrict__ Ucur, int cs, int K, int m, int est,
                                                             double ridge, int* active, const double* __restrict__ cost) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    __shared__ int sh_idx[64];
    if (cost) {
        __shared__ __attribute__((aligned(16))) double c[kCeSortMax], sc[kCeSortMax];
        __shared__ double eb_red[kCeThreads / 64];
        __shared__ int sh_broke;
        const int i = threadIdx.x;
        const double cr = (i < K) ? cost[(size_t)b * K + i] : INFINITY;
        const double ci = (cr != cr) ? INFINITY : cr;            // NaN ranks like +inf: unique ranks, no stale order[] entries (see k_sortperm_rank)
        if (i < kCeSortMax) { c[i] = ci; sc[i] = INFINITY; }
        if (i == 0) sh_broke = 0;
        __syncthreads();
        if (i < K) {
            int rank = 0;
            const int K4 = K & ~3;
            for (int j = 0; j < K4; j += 4) {
                const double c0 = c[j], c1 = c[j + 1], c2 = c[j + 2], c3 = c[j + 3];
                rank += ((c0 < ci) || (c0 == ci && j < i)) + ((c1 < ci) || (c1 == ci && j + 1 < i)) + ((c2 < ci) || (c2 == ci && j + 2 < i)) +
                        ((c3 < ci) || (c3 == ci && j + 3 < i));
            }
            for (int j = K4; j < K; ++j) rank += ((c[j] < ci) || (c[j] == ci && j < i));
            order[(size_t)b * K + rank] = i; sc[rank] = cr;
            if (rank < m) sh_idx[rank] = i;
        }
        __syncthreads();
        if (m >= 2 && active) {                                  // elite early break on the sorted costs (as elite_break_tail, kernels_select.hip)
            double mx = -INFINITY;
            for (int j = threadIdx.x; j + 1 < m; j += kCeThreads) mx = fmax(mx, fabs(sc[j + 1] - sc[j]));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
            if ((threadIdx.x & 63) == 0) eb_red[threadIdx.x >> 6] = mx;
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int w = 1; w < kCeThreads / 64; ++w) mx = fmax(mx, eb_red[w]);
                if (mx < 10e-3) { active[b] = 0; sh_broke = 1; }
            }
            __syncthreads();
            if (sh_broke) return;
        }
    }
    extern __shared__ __attribute__((aligned(16))) double sh_ce[];
    const int nt = (cs + 15) / 16, rows = nt * 16, mpad = (m + 3) & ~3, ld = mpad + 1;      // odd row stride: conflict-free operand reads
    double* X = sh_ce;                                           // [rows][ld]
    double* dg = X + (size_t)rows * ld;                          // [rows] diagonal of S
    __shared__ double red[2 * kCeWaves];
    __shared__ double sh_lam, sh_f;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const double* Eb = E + (size_t)b * cs * K;
    const int32_t* ob = order + (size_t)b * K;
    // ---- gather (every thread's loads in flight together: index -> value is two dependent global round trips), then mean + centre -----------
    if (!cost && tid < m) sh_idx[tid] = ob[tid];
    for (int e = tid; e < rows * ld; e += kCeThreads) X[e] = 0.0;
    __syncthreads();
    for (int e0 = tid; e0 < cs * m; e0 += kCeThreads * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = min(e0 + u * kCeThreads, cs * m - 1), r = e / m, j = e - r * m; v[u] = Eb[(size_t)r * K + sh_idx[j]]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + u * kCeThreads; if (e < cs * m) { const int r = e / m, j = e - r * m; X[(size_t)r * ld + j] = v[u]; } }
    }
    __syncthreads();
    for (int r = wv; r < cs; r += kCeWaves) {                    // one wave per row, lanes along the elite columns (m <= 64)
        const double v = (lane < m) ? X[(size_t)r * ld + lane] : 0.0;
        double t = v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        const double mean = t / m;
        if (lane == 0) { mu_out[(size_t)b * cs + r] = mean; Ucur[(size_t)b * cs + r] += mean; }     // pol.U += μ′ (:465-466)
        if (lane < m) X[(size_t)r * ld + lane] = v - mean;
    }
    __syncthreads();
    // ---- second (and, for :ss / :lw, fourth) moments: tile pairs dealt to the waves ------------------------------------------------------------
    const bool fourth = est == MPOPIS_SIGMA_EST_SS || est == MPOPIS_SIGMA_EST_LW;
    const int npairs = nt * (nt + 1) / 2;
    const double inv_m = 1.0 / m;
    v4f64_ce accS[kCeMaxPairs], accQ[kCeMaxPairs];
    int pa[kCeMaxPairs], pb[kCeMaxPairs];
#pragma unroll
    for (int p = 0; p < kCeMaxPairs; ++p) {
        accS[p] = (v4f64_ce){0.0, 0.0, 0.0, 0.0}; accQ[p] = (v4f64_ce){0.0, 0.0, 0.0, 0.0};
        const int q = wv + p * kCeWaves;
        if (q < npairs) ce_decode(q, &pa[p], &pb[p]); else { pa[p] = 0; pb[p] = 0; }
    }
    for (int kk = 0; kk < mpad; kk += 4) {
#pragma unroll
        for (int p = 0; p < kCeMaxPairs; ++p) {
            if (wv + p * kCeWaves < npairs) {                    // wave-uniform
                const double av = X[(size_t)(pa[p] * 16 + li) * ld + kk + lk], bv = X[(size_t)(pb[p] * 16 + li) * ld + kk + lk];
                accS[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, accS[p], 0, 0, 0);
                if (fourth) accQ[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv * bv, av * av, accQ[p], 0, 0, 0);
            }
        }
    }
    // entry (p, r) of this lane: S[a][c] with a = pa 16 + li, c = pb 16 + lk + 4r
#pragma unroll
    for (int p = 0; p < kCeMaxPairs; ++p) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            accS[p][r] *= inv_m;
            if (wv + p * kCeWaves < npairs && pa[p] == pb[p] && li == lk + 4 * r) dg[pa[p] * 16 + li] = accS[p][r];
        }
    }
    __syncthreads();
    // ---- shrinkage intensity -------------------------------------------------------------------------------------------------------------------
    double lam = 0.0, ftarget = 0.0;
    if (est != MPOPIS_SIGMA_EST_MLE) {
        double r0 = 0.0, r1 = 0.0;                               // two block sums: (num, den) for :ss / :lw, (tr, tr2) for :rblw / :oas
#pragma unroll
        for (int p = 0; p < kCeMaxPairs; ++p) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = pa[p] * 16 + li, c = pb[p] * 16 + lk + 4 * r;
                const bool lower = (wv + p * kCeWaves < npairs) && a < cs && c < cs && a >= c;
                const double sv = accS[p][r];
                if (fourth) {
                    if (lower && a != c) {                       // each off-diagonal pair counts twice: (a,c) and (c,a)
                        // 1 / (sd_a sd_c)²: one reciprocal (rcp seed + two Newton steps, <= 1 ulp) shared by r_ab² and the fourth moment
                        double rdd = 1.0;
                        if (est == MPOPIS_SIGMA_EST_SS) {
                            const double dd = dg[a] * dg[c];
                            rdd = __builtin_amdgcn_rcp(dd);
                            rdd = fma(fma(-dd, rdd, 1.0), rdd, rdd);
                            rdd = fma(fma(-dd, rdd, 1.0), rdd, rdd);
                        }
                        const double rab2 = sv * sv * rdd;       // r_ab²
                        r0 += 2.0 * fma(accQ[p][r], rdd, -(double)m * rab2);
                        r1 += 2.0 * rab2;
                    }
                } else if (lower) {
                    if (a == c) { r0 += sv; r1 = fma(sv, sv, r1); } else r1 = fma(2.0 * sv, sv, r1);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { r0 += __shfl_xor(r0, o, 64); r1 += __shfl_xor(r1, o, 64); }
        if (lane == 0) { red[wv] = r0; red[kCeWaves + wv] = r1; }
        __syncthreads();
        if (tid == 0) {
            double t0 = 0.0, t1 = 0.0;
            for (int w = 0; w < kCeWaves; ++w) { t0 += red[w]; t1 += red[kCeWaves + w]; }
            double l;
            if (fourth) {                                        // λ* = Σ Var^(r_ab) / Σ r_ab², Var^ = n/(n-1)³ (Q_ab - n r_ab²)
                const double num = t0 * ((double)m / ((double)(m - 1) * (m - 1) * (m - 1)));
                l = t1 > 0 ? num / t1 : 1.0;
                l = fmin(fmax(l, 0.0), 1.0);
                sh_f = 0.0;
            } else {                                             // Chen et al. 2010 eqs. (17) / (23), F = tr(S)/p I
                const double pp = cs, n = m, tr = t0, tr2 = t1, dd = tr2 - tr * tr / pp;
                l = (est == MPOPIS_SIGMA_EST_OAS) ? ((1 - 2 / pp) * tr2 + tr * tr) / ((n + 1 - 2 / pp) * dd) : ((n - 2) / n * tr2 + tr * tr) / ((n + 2) * dd);
                l = (dd > 0) ? fmin(fmax(l, 0.0), 1.0) : 1.0;
                sh_f = tr / pp;
            }
            sh_lam = l;
        }
        __syncthreads();
        lam = sh_lam; ftarget = sh_f;
    }
    // ---- Σ′ (both triangles) -------------------------------------------------------------------------------------------------------------------
    double* Sb = Sg + (size_t)b * cs * cs;
    const bool common = est == MPOPIS_SIGMA_EST_RBLW || est == MPOPIS_SIGMA_EST_OAS;
#pragma unroll
    for (int p = 0; p < kCeMaxPairs; ++p) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = pa[p] * 16 + li, c = pb[p] * 16 + lk + 4 * r;
            if ((wv + p * kCeWaves < npairs) && a < cs && c < cs && a >= c) {
                double v = accS[p][r];
                if (a == c) v = common ? (1 - lam) * v + (lam * ftarget + ridge) : v + ridge;
                else v = (est == MPOPIS_SIGMA_EST_MLE) ? v : v * (1 - lam);
                Sb[(size_t)a + (size_t)c * cs] = v;
                Sb[(size_t)c + (size_t)a * cs] = v;
            }
        }
    }
}

// applies when the elite set is small: cs <= 128 (8 row tiles), m <= 64 columns
size_t ce_cov_small_lds(int cs, int m, int est) {
    (void)est;
    const size_t rows = (size_t)((cs + 15) / 16) * 16, ld = (size_t)((m + 3) & ~3) + 1;
    return (rows * ld + rows) * sizeof(double);
}
bool ce_cov_small_ok(int cs, int m, int est) { return cs <= 128 && m >= 2 && m <= 64 && ce_cov_small_lds(cs, m, est) <= 150 * 1024; }
bool ce_sort_fusable(int K) { return K <= kCeSortMax; }
void launch_ce_cov_small(const double* E, int32_t* order, double* mu, double* S, double* Ucur, int B, int cs, int K, int m, int est, double ridge,
                         int* active, hipStream_t s, const double* cost) {
    static std::atomic<unsigned long long> seen{0};
    ensure_dyn_lds((const void*)k_ce_cov_small, 150 * 1024, seen);
    hipLaunchKernelGGL(k_ce_cov_small, dim3(B), dim3(kCeThreads), ce_cov_small_lds(cs, m, est), s, E, order, mu, S, Ucur, cs, K, m, est, ridge, active, cost);
}

}  // namespace mpopis
