// kernels_invsqrt.hip -- what CMAMPPI_Policy needs from C = Σ^-0.5 (src/mppi_mpopi_policies.jl:580) WITHOUT forming C.
//
// The reference computes the dense matrix C = Σ^-0.5 (LinearAlgebra: symmetric eigen-decomposition) every AIS iteration,
// but only ever consumes
//     C * δw                      (:581, the evolution path p_σ)                        -- a vector
//     norm(C * δs[order[ii]])     (:593, δs[..] is a SCALAR through linear indexing)    -- |δ| ||C||_F, i.e. tr(Σ^-1)
// so the n^3 matrix function (round 1: 16 coupled Newton-Schulz iterations = 48 batched n x n GEMMs per update, 48 % of
// a C4 step) is replaced by two O(n^2 m) pieces, one workgroup per trial slot / block column:
//   k_trtri_fro        ||L^-1||_F^2 for the Cholesky factor that the sampler already holds (σ²Σ = L L'):
//                      tr(Σ^-1) = σ² ||L^-1||_F^2.  Block column J of X = L^-1 depends on L only, so the n/16 block
//                      columns are independent workgroups; X stays in LDS, only the sum of squares leaves.
//   k_lanczos_invsqrt  y = Σ^-0.5 δw = ||δw|| V_m f(T_m) e_1: Lanczos with full re-orthogonalisation (CGS2) on Σ,
//                      f(T_m) e_1 by the 64-node quadrature of invsqrt_quad.h (lane = node, every node one SPD tridiagonal
//                      solve).  The same solves give the residual of each shifted system, hence a rigorous error bound
//                      (λ_min(Σ) >= 1/tr(Σ^-1)), which is the stopping rule (1e-13 relative).  CMA covariances are
//                      low-rank updates of a two-eigenvalue block-diagonal matrix: m = 4 ... 40.
#include "engine.h"
#include "invsqrt_quad.h"

namespace mpopis {

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LDS traffic between the lanes of ONE wave: make earlier LDS writes visible / keep later ones from moving up
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kTB = 16;                      // block size of the triangular inverse
constexpr int kInvThreads = 1024, kInvWaves = kInvThreads / 64;

// L(r, c) of the n x n column-major factor, padded with the identity beyond n
__device__ __forceinline__ double ld_L(const double* __restrict__ L, int n, int r, int c) {
    return (r < n && c < n) ? L[r + (size_t)c * n] : ((r == c) ? 1.0 : 0.0);
}

}  // namespace

// part[b][J] = sum of squares of the entries of block column J of L^-1 (rows/cols < n).
// Block row I of the block column:  X(I) = -L(I,I)^-1 sum_{J <= K < I} L(I,K) X(K).  The chain over I is serial, so each step
// is kept to ONE global round trip (~1.5 us on this part: a wave batches the 16 loads of its L(I,K) block before using any)
// and two barriers: the waves split K, apply -L(I,I)^-1 to their own partial product (wave-local, through LDS) and the
// 16 partial blocks are summed into X(I).  The inverses of all diagonal blocks are formed up front, in parallel over waves.
__global__ void __launch_bounds__(kInvThreads) k_trtri_fro(const double* __restrict__ Lall, size_t Lstride, int n, int nb, double* __restrict__ part,
                                                           const int* active) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.y, J = blockIdx.x;
    if (active && !active[b]) return;
    extern __shared__ __attribute__((aligned(16))) double sh_inv[];
    const int nbj = nb - J;
    double* Xs = sh_inv;                               // [nbj][16][16]        block column of X = L^-1 (row-major 16x16 blocks)
    double* Dall = Xs + (size_t)nbj * 256;             // [nbj][16][16]        Dall[I-J][i][k] = (L(I,I)^-1)[i][k]
    double* Ws = Dall + (size_t)nbj * 256;             // [kInvWaves][256]     per wave: raw partial product, then (in place) -Dinv * partial
    double* red = Ws + kInvWaves * 256;                // [kInvWaves]
    const double* L = Lall + (size_t)b * Lstride;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    // ---- inverses of the diagonal blocks I = J .. nb-1: wave wv takes I = J + wv, J + wv + 16, ... ------------------------------
    for (int I = J + wv; I < nb; I += kInvWaves) {
        double* Ld = Ws + wv * 256;                    // scratch: the diagonal block itself
        double* Di = Dall + (size_t)(I - J) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) Ld[i * 16 + 4 * g + q] = ld_L(L, n, 16 * I + i, 16 * I + 4 * g + q);
        wave_lds_sync();
        if (lane < 16) {                                // column c of L(I,I)^-1 by forward substitution (rows in order)
            const int c = lane;
#pragma unroll 1
            for (int r = 0; r < 16; ++r) {
                double sacc = (r == c) ? 1.0 : 0.0;
#pragma unroll 1
                for (int k = 0; k < r; ++k) sacc = fma(-Ld[r * 16 + k], Di[k * 16 + c], sacc);   // own column only: program order suffices
                Di[r * 16 + c] = sacc / Ld[r * 16 + r];
            }
        }
        wave_lds_sync();
    }
    __syncthreads();
    for (int e = tid; e < 256; e += kInvThreads) Xs[e] = Dall[e];          // X(J) = L(J,J)^-1
    __syncthreads();
    for (int I = J + 1; I < nb; ++I) {
        const int nact = min(kInvWaves, I - J);                              // waves that own at least one K < I
        if (wv < nact) {
            // ---- P_wv = sum_{K = J+wv, J+wv+16, ... < I} L(I,K) X(K);  lane = (row i, column group g): P[i][4g .. 4g+3]
            double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
            for (int K = J + wv; K < I; K += kInvWaves) {
                double l[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) l[k] = ld_L(L, n, 16 * I + i, 16 * K + k);      // 16 independent loads in flight
                const double* xk = Xs + (size_t)(K - J) * 256 + 4 * g;
#pragma unroll 4
                for (int k = 0; k < 16; ++k) {
                    p0 = fma(l[k], xk[k * 16 + 0], p0); p1 = fma(l[k], xk[k * 16 + 1], p1);
                    p2 = fma(l[k], xk[k * 16 + 2], p2); p3 = fma(l[k], xk[k * 16 + 3], p3);
                }
            }
            double* w = Ws + wv * 256;
            w[i * 16 + 4 * g + 0] = p0; w[i * 16 + 4 * g + 1] = p1; w[i * 16 + 4 * g + 2] = p2; w[i * 16 + 4 * g + 3] = p3;
            wave_lds_sync();
            // ---- Q_wv = -L(I,I)^-1 P_wv (lower triangular inverse: k <= i)
            const double* Di = Dall + (size_t)(I - J) * 256 + i * 16;
            double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll 4
            for (int k = 0; k < 16; ++k) {
                const double d = (k <= i) ? -Di[k] : 0.0;
                q0 = fma(d, w[k * 16 + 4 * g + 0], q0); q1 = fma(d, w[k * 16 + 4 * g + 1], q1);
                q2 = fma(d, w[k * 16 + 4 * g + 2], q2); q3 = fma(d, w[k * 16 + 4 * g + 3], q3);
            }
            wave_lds_sync();                                                  // every lane has read the raw block: overwrite it in place
            w[i * 16 + 4 * g + 0] = q0; w[i * 16 + 4 * g + 1] = q1; w[i * 16 + 4 * g + 2] = q2; w[i * 16 + 4 * g + 3] = q3;
        }
        __syncthreads();
        if (tid < 256) {                                                      // X(I) = sum of the waves' Q blocks
            double sacc = 0.0;
            for (int w = 0; w < nact; ++w) sacc += Ws[w * 256 + tid];
            Xs[(size_t)(I - J) * 256 + tid] = sacc;
        }
        __syncthreads();
    }
    double s = 0.0;
    for (int e = tid; e < nbj * 256; e += kInvThreads) {
        const int r = 16 * (J + (e >> 8)) + ((e & 255) >> 4), c = 16 * J + (e & 15);
        const double v = Xs[e];
        if (r < n && c < n) s = fma(v, v, s);
    }
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int w = 0; w < kInvWaves; ++w) t += red[w]; part[(size_t)b * nb + J] = t; }
}

// Lanczos runs until the error bound is met, at most n steps (with full re-orthogonalisation the Krylov space is then
// exhausted and the result exact): generic dense covariances need m ~ n, CMA's low-rank-updated ones m = 4 ... 40.
//
// Everything on the serial chain of one Lanczos step is arranged around the ~1.5-2 us a dependent global-memory round trip
// costs a lone workgroup on this part:
//   * w = A v: each wave owns columns c = wv, wv + 16, ...; the loads of kLanCG columns x kLanNQ row chunks (30 per lane) are
//     issued back to back before any is consumed (4 round trips at n = 300, instead of one per column);
//   * the Lanczos basis lives in LDS (first `nvl` vectors; later ones in the global workspace), so both Gram-Schmidt passes
//     are LDS-only;
//   * the stopping rule is O(1) per step: for the shifted systems (T_m + s_j I) z = e_1 the LEADING pivots d_m and the last
//     solution component z_m = g_m follow from the previous step by a two-term recurrence (lane = quadrature node, state in
//     registers).  The full coefficient vector c = sum_j w_j (T_m + s_j I)^-1 e_1 (O(m) back substitution per node) is formed
//     once, after convergence.  ||c|| >= λ_max^-1/2 >= M^-1/2 turns the relative test into a (slightly conservative) absolute one.
constexpr int kLanThreads = 1024, kLanWaves = kLanThreads / 64;
constexpr int kLanNQ = 5;                    // rows per lane and pass of the mat-vec (64*5 = 320 >= cs = 300 in one pass)
constexpr int kLanCG = 4;                    // columns whose loads are in flight together
constexpr int kLanPivLds = 32;               // pivot rows of the final back substitution kept in LDS (longer runs: global workspace)
constexpr double kLanTol = 1e-13;

// y[b] = A[b]^-1/2 bvec[b];  fro[b] = scale[b] * sum_J part[b][J]  (= tr(A^-1) = ||A^-1/2||_F^2)
// status: MPOPIS_ERR_NUMERIC when the spectrum bounds are unusable (non-finite input, M/m beyond 1e14)
__global__ void __launch_bounds__(kLanThreads) k_lanczos_invsqrt(const double* __restrict__ Aall, const double* __restrict__ bvec, size_t bstride,
                                                                 const double* __restrict__ part, int nb, const double* __restrict__ scale,
                                                                 double* __restrict__ Vall, double* __restrict__ yall, double* __restrict__ fro_out,
                                                                 int* __restrict__ msteps, int n, int nvl, int* status, const int* active) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    extern __shared__ __attribute__((aligned(16))) double sh_lan[];
    double* part_v = sh_lan;                            // [kLanWaves][n]
    double* vcur = part_v + (size_t)kLanWaves * n;      // [n]
    double* wv_ = vcur + n;                             // [n]  the working vector w
    double* coef = wv_ + n;                             // [n + 1]
    double* alpha = coef + n + 1;                       // [n]
    double* beta = alpha + n;                           // [n]
    double* cvec = beta + n;                            // [n]
    double* red = cvec + n;                             // [kLanWaves + 4]
    double* pivl = red + kLanWaves + 4;                 // [2][kLanPivLds][64]  d_i and e_i = β_i/d_i of the back substitution
    double* Vl = pivl + 2 * kLanPivLds * 64;            // [nvl][n]  Lanczos basis, LDS-resident part
    __shared__ int sh_flag;
    const double* A = Aall + (size_t)b * n * n;
    const double* bv = bvec + (size_t)b * bstride;
    double* Vg = Vall + (size_t)b * (size_t)(n + 1 + 128) * n;  // basis vectors v_0 .. v_n (only k >= nvl are ever touched)
    double* pivg = Vg + (size_t)(n + 1) * n;                    // [2][n][64]
    double* y = yall + (size_t)b * n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    auto block_sum = [&](double v) -> double {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wv] = v;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kLanWaves; ++w) t += red[w];
        return t;
    };
    double fro = 0.0;
    for (int J = 0; J < nb; ++J) fro += part[(size_t)b * nb + J];
    fro *= scale ? scale[b] : 1.0;
    if (tid == 0) fro_out[b] = fro;
    // ---- v_0 = b / ||b|| --------------------------------------------------------------------------------------------------
    double s2 = 0.0;
    for (int i = tid; i < n; i += kLanThreads) { const double v = bv[i]; s2 = fma(v, v, s2); }
    const double nrm_b = sqrt(block_sum(s2));
    if (!(nrm_b > 0.0) || !(fro > 0.0)) {                             // δw = 0 -> y = 0; NaN input / unusable trace -> numeric error
        for (int i = tid; i < n; i += kLanThreads) y[i] = 0.0;
        if (tid == 0) { msteps[b] = 0; if (nrm_b > 0.0 || !(nrm_b == nrm_b)) status[b] = MPOPIS_ERR_NUMERIC; }
        return;
    }
    for (int i = tid; i < n; i += kLanThreads) { const double v = bv[i] / nrm_b; vcur[i] = v; if (nvl > 0) Vl[i] = v; else Vg[i] = v; }
    __syncthreads();
    const int mcap = n;
    int m = 0;
    double Mhi = 0.0, mlo = 0.0, q_shift = 0.0, q_weight = 0.0, d_prev = 1.0, g_prev = 0.0;   // quadrature node + pivot recurrence of this lane
    for (int j = 0; j < mcap; ++j) {
        // ---- w = A v_j : lanes along rows (coalesced column reads), waves split the columns, kLanCG columns of loads in flight ----
        double colmax = 0.0;
        for (int r0 = 0; r0 < n; r0 += 64 * kLanNQ) {
            double acc[kLanNQ];
#pragma unroll
            for (int q = 0; q < kLanNQ; ++q) acc[q] = 0.0;
            for (int c0 = wv; c0 < n; c0 += kLanWaves * kLanCG) {
                double a[kLanCG][kLanNQ];
#pragma unroll
                for (int u = 0; u < kLanCG; ++u) {
                    // wave-uniform column base + one lane offset + immediate 512 q: rows past n read into the next column / the
                    // padding behind the last slot (launch_invsqrt_vec's contract) and are never used
                    const double* col = A + (size_t)min(c0 + kLanWaves * u, n - 1) * n + r0;
#pragma unroll
                    for (int q = 0; q < kLanNQ; ++q) a[u][q] = col[lane + 64 * q];
                }
#pragma unroll
                for (int u = 0; u < kLanCG; ++u) {
                    const int c = c0 + kLanWaves * u;
                    const double vc = (c < n) ? vcur[c] : 0.0;
#pragma unroll
                    for (int q = 0; q < kLanNQ; ++q) acc[q] = fma(a[u][q], vc, acc[q]);
                    if (j == 0 && c < n) {                                      // first pass: ||A||_inf (A symmetric: column abs sums)
                        double sa = 0.0;
#pragma unroll
                        for (int q = 0; q < kLanNQ; ++q) if (r0 + lane + 64 * q < n) sa += fabs(a[u][q]);
                        colmax = fmax(colmax, wave_sum(sa));                    // (exact for n <= 64 kLanNQ; larger n: per-chunk sums, see below)
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < kLanNQ; ++q) if (r0 + lane + 64 * q < n) part_v[(size_t)wv * n + r0 + lane + 64 * q] = acc[q];
        }
        __syncthreads();
        for (int i = tid; i < n; i += kLanThreads) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kLanWaves; ++w) s += part_v[(size_t)w * n + i];
            wv_[i] = s;
        }
        if (j == 0) {
            // M >= λ_max: max column abs sum; for n > 64 kLanNQ a column's sum is split over row chunks, so bound it by the number of
            // chunks times the largest chunk sum (still an upper bound, only looser)
            const int nchunk = (n + 64 * kLanNQ - 1) / (64 * kLanNQ);
            __syncthreads();
            if (lane == 0) red[wv] = colmax;
            __syncthreads();
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < kLanWaves; ++w) t = fmax(t, red[w]);
            Mhi = t * nchunk;
            mlo = fmin(1.0 / fro, 0.5 * Mhi);
            if (!invsqrt_quad_node(mlo, Mhi, lane, 64, &q_shift, &q_weight)) {       // uniform: depends on mlo / Mhi only
                for (int i = tid; i < n; i += kLanThreads) y[i] = 0.0;
                if (tid == 0) { msteps[b] = 0; status[b] = MPOPIS_ERR_NUMERIC; }
                return;
            }
        }
        __syncthreads();
        // ---- orthogonalise against v_0..v_j twice (classical Gram-Schmidt, CGS2); α_j = the v_j coefficient -------------------
        double a_j = 0.0;
        const int jl = min(j + 1, nvl);                                             // vectors 0 .. jl-1 in LDS, jl .. j in global memory
        for (int pass = 0; pass < 2; ++pass) {
            for (int k = wv; k <= j; k += kLanWaves) {
                double s = 0.0;
                if (k < jl) { for (int i = lane; i < n; i += 64) s = fma(Vl[(size_t)k * n + i], wv_[i], s); }
                else { for (int i = lane; i < n; i += 64) s = fma(Vg[(size_t)k * n + i], wv_[i], s); }
                s = wave_sum(s);
                if (lane == 0) coef[k] = s;
            }
            __syncthreads();
            a_j += coef[j];
            for (int i = tid; i < n; i += kLanThreads) {
                double s = wv_[i];
                for (int k = 0; k < jl; ++k) s = fma(-coef[k], Vl[(size_t)k * n + i], s);
                for (int k = jl; k <= j; ++k) s = fma(-coef[k], Vg[(size_t)k * n + i], s);
                wv_[i] = s;
            }
            __syncthreads();
        }
        double w2 = 0.0;
        for (int i = tid; i < n; i += kLanThreads) { const double v = wv_[i]; w2 = fma(v, v, w2); }
        const double bt = sqrt(block_sum(w2));
        if (tid == 0) { alpha[j] = a_j; beta[j] = bt; }
        m = j + 1;
        // ---- stopping rule, O(1): leading pivot d_m = α_m + s - β_{m-1}²/d_{m-1}, last solution component z_m = g_m = -β_{m-1} g_{m-1}/d_m
        //      (g_1 = 1/d_1); residual of node j's shifted system = β_m |z_m|, error <= sum_j w_j β_m |z_m| / (λ_min + s_j)  (per unit ||b||)
        {
            const double bprev = (j > 0) ? beta[j - 1] : 0.0;                        // written one step (and several barriers) ago
            const double d = a_j + q_shift - ((j > 0) ? bprev * bprev / d_prev : 0.0);
            const double gcur = (j > 0) ? -bprev * g_prev / d : 1.0 / d;
            d_prev = d; g_prev = gcur;
            const double eb = bt * wave_sum(q_weight * fabs(gcur) / (mlo + q_shift));
            const bool conv = (eb <= kLanTol * (1.0 / sqrt(Mhi))) || (bt <= 1e-14 * Mhi) || (m >= mcap);
            if (tid == 0) sh_flag = conv ? 1 : 0;
        }
        __syncthreads();
        if (sh_flag) break;
        for (int i = tid; i < n; i += kLanThreads) {
            const double v = wv_[i] / bt;
            vcur[i] = v;
            if (j + 1 < nvl) Vl[(size_t)(j + 1) * n + i] = v; else Vg[(size_t)(j + 1) * n + i] = v;
        }
        __syncthreads();
    }
    // ---- c = sum_nodes w (T_m + s I)^-1 e_1: LDL' per node (lane), forward g_i, back z_i = g_i - e_i z_{i+1}; wave 0 -----------------
    if (wv == 0) {
        auto dpiv = [&](int i) -> double& { return i < kLanPivLds ? pivl[i * 64 + lane] : pivg[(size_t)i * 64 + lane]; };
        auto epiv = [&](int i) -> double& { return i < kLanPivLds ? pivl[(kLanPivLds + i) * 64 + lane] : pivg[((size_t)n + i) * 64 + lane]; };
        double d = 1.0, g = 0.0;
        for (int i = 0; i < m; ++i) {
            const double bp = (i > 0) ? beta[i - 1] : 0.0;
            const double dn = alpha[i] + q_shift - ((i > 0) ? bp * bp / d : 0.0);
            g = (i > 0) ? -bp * g / dn : 1.0 / dn;
            d = dn;
            dpiv(i) = g;                                                            // g_i
            epiv(i) = (i + 1 < m) ? beta[i] / dn : 0.0;                              // e_i = β_i / d_i
        }
        double z = 0.0;
        for (int i = m - 1; i >= 0; --i) {
            z = dpiv(i) - epiv(i) * z;
            const double ci = wave_sum(q_weight * z);
            if (lane == 0) cvec[i] = ci;
        }
    }
    __syncthreads();
    const int ml = min(m, nvl);
    for (int i = tid; i < n; i += kLanThreads) {
        double s = 0.0;
        for (int k = 0; k < ml; ++k) s = fma(cvec[k], Vl[(size_t)k * n + i], s);
        for (int k = ml; k < m; ++k) s = fma(cvec[k], Vg[(size_t)k * n + i], s);
        y[i] = nrm_b * s;
    }
    if (tid == 0) msteps[b] = m;
}

size_t invsqrt_workspace_doubles(int B, int n) { return (size_t)B * (size_t)(n + 1 + 128) * n; }
int invsqrt_max_n() {
    // dynamic LDS of k_lanczos_invsqrt: (kLanWaves + 6) n + ... doubles, and of k_trtri_fro: 32 (n + 15) + 4112 doubles, both <= 150 KiB
    return std::min((int)((150 * 1024 / 8 - 64 - 2 * kLanPivLds * 64) / (kLanWaves + 8)), (int)((150 * 1024 / 8 - 4112) / 32 - 15));
}

// y = A^-1/2 b and fro = tr(A^-1) per slot, from A (n x n, SPD) and the Cholesky factor L of scale*A (scale: per-slot, nullable).
// A must stay readable for kInvsqrtPadDoubles doubles behind its last slot (the mat-vec reads whole 64-row chunks).
// part[b][nb] = per-block-column partial sums of ||L^-1||_F^2 (only L is read: may run on another stream beside the sort / elite mean)
void launch_trtri_fro(const double* L, size_t Lstride, double* part, int B, int n, const int* active, hipStream_t s) {
    const int nb = (n + kTB - 1) / kTB;
    const size_t lds1 = ((size_t)2 * nb * 256 + kInvWaves * 256 + kInvWaves) * sizeof(double);
    static std::atomic<unsigned long long> seen1{0};
    ensure_dyn_lds((const void*)k_trtri_fro, 150 * 1024, seen1);
    hipLaunchKernelGGL(k_trtri_fro, dim3(nb, B), dim3(kInvThreads), lds1, s, L, Lstride, n, nb, part, active);
}

// y = A^-1/2 b and fro = scale * sum(part) (the partial sums of launch_trtri_fro; 1/fro is also the quadrature's lower spectrum bound)
void launch_lanczos_invsqrt(const double* A, const double* scale, const double* bvec, size_t bstride, const double* part,
                            double* V, double* y, double* fro, int* msteps, int B, int n, int* status, const int* active, hipStream_t s) {
    const int nb = (n + kTB - 1) / kTB;
    static std::atomic<unsigned long long> seen2{0};
    const size_t fixed = (size_t)(kLanWaves + 6) * n + 1 + kLanWaves + 4 + 2 * kLanPivLds * 64;        // doubles
    const int nvl = (int)std::min<size_t>(n + 1, (150 * 1024 / sizeof(double) - fixed) / n);            // basis vectors that fit next to it
    const size_t lds2 = (fixed + (size_t)nvl * n) * sizeof(double);
    ensure_dyn_lds((const void*)k_lanczos_invsqrt, 150 * 1024, seen2);
    hipLaunchKernelGGL(k_lanczos_invsqrt, dim3(B), dim3(kLanThreads), lds2, s, A, bvec, bstride, part, nb, scale, V, y, fro, msteps, n, nvl, status, active);
}

void launch_invsqrt_vec(const double* A, const double* L, size_t Lstride, const double* scale, const double* bvec, size_t bstride,
                        double* part, double* V, double* y, double* fro, int* msteps, int B, int n, int* status, const int* active, hipStream_t s) {
    launch_trtri_fro(L, Lstride, part, B, n, active, s);
    launch_lanczos_invsqrt(A, scale, bvec, bstride, part, V, y, fro, msteps, B, n, status, active, s);
}

}  // namespace mpopis
