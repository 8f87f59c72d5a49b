// kernels_linalg.hip -- the small dense FP64 linear algebra the reference gets from LAPACK via
// Distributions/PDMats/StatsBase, restated as device kernels (one workgroup per trial slot):
//   MvNormal(Σ') -> PDMat -> cholesky(Σ')            src/mppi_mpopi_policies.jl:192,307,352,447,551-553,650,723,796
//   invcov(P) as used in the control cost             :194,309,353,449,555,651,725,798 (only the row γ U_orig' Σ⁻¹ is needed)
//   StatsBase.mean_and_cov(E, pw, 2) (weighted, uncorrected) + 10e-9 I      :730-733
//   StatsBase.mean_and_cov(E', 2) of resampled columns (corrected)         :806-808
//   cov(SimpleCovariance(), elite') + 10e-9 I, mean(elite, dims=2)         :464-465
#include "engine.h"

namespace mpopis {

constexpr int kNB = 16;

// ---------------------------------------------------------------------------------------------
// Blocked right-looking Cholesky, one workgroup (256 threads) per matrix, panel width 16:
//   (1) wave 0 factors the 16x16 diagonal block in registers (lane = row, v_readlane broadcasts),
//   (2) one thread per row solves the panel below against it,
//   (3) all threads apply the rank-16 trailing update.
// The working copy lives in LDS when it fits (n_pad^2*8 <= 150 KiB, i.e. cs <= 128: the 1-car
// configs), otherwise in the output buffer in global memory (L2 resident).
// scale[b] (nullable) multiplies A first (CMA: MvNormal(σ²Σ), :551).  On a non-positive pivot
// status[b] = MPOPIS_ERR_NOT_PD and active[b] = 0 (the reference throws PosDefException).
// ---------------------------------------------------------------------------------------------
template <bool LDS>
__global__ void __launch_bounds__(256) k_potrf(const double* __restrict__ A, size_t Astride, double* __restrict__ Lout,
                                               int n, int npad, const double* scale, int* status, int* active) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int failed;
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double* Ab = A + (size_t)b * Astride;
    double* Lb = Lout + (size_t)b * n * n;
    const double sc = scale ? scale[b] : 1.0;
    // working matrix W (ld = ldw): LDS copy padded to a multiple of 16 with an identity tail
    double* W; int ldw, m;
    if (LDS) { W = smem; ldw = npad; m = npad; } else { W = Lb; ldw = n; m = n; }
    if (tid == 0) failed = 0;
    for (int idx = tid; idx < m * m; idx += 256) {
        const int i = idx % m, j = idx / m;
        double v;
        if (i < n && j < n) v = (i >= j) ? sc * Ab[(size_t)i + (size_t)j * n] : 0.0;
        else v = (i == j) ? 1.0 : 0.0;
        W[(size_t)i + (size_t)j * ldw] = v;
    }
    __syncthreads();

    for (int j0 = 0; j0 < m; j0 += kNB) {
        const int nb = min(kNB, m - j0);
        // (1) diagonal block, wave 0, lane = row (rows >= nb idle; columns >= nb behave as identity)
        if (wv == 0) {
            double d[kNB];
            const bool rowok = lane < nb;
#pragma unroll
            for (int c = 0; c < kNB; ++c) d[c] = (rowok && c < nb) ? W[(size_t)(j0 + lane) + (size_t)(j0 + c) * ldw] : ((lane == c) ? 1.0 : 0.0);
            bool bad = false;
#pragma unroll
            for (int jj = 0; jj < kNB; ++jj) {
                const double piv = __shfl(d[jj], jj, 64);
                if (!(piv > 0.0)) bad = true;
                const double ljj = sqrt(piv);
                if (lane == jj) d[jj] = ljj; else if (lane > jj) d[jj] = d[jj] / ljj;
#pragma unroll
                for (int c = jj + 1; c < kNB; ++c) {
                    const double lcj = __shfl(d[jj], c, 64);
                    if (lane >= c) d[c] = fma(-d[jj], lcj, d[c]);
                }
            }
            if (rowok) {
#pragma unroll
                for (int c = 0; c < kNB; ++c) if (c < nb && c <= lane) W[(size_t)(j0 + lane) + (size_t)(j0 + c) * ldw] = d[c];
            }
            if (bad && lane == 0) failed = 1;
        }
        __syncthreads();
        if (failed) break;
        const int i1 = j0 + nb;              // first row below the panel
        // (2) panel solve: row i of L21 = A21[i,:] * L11^-T
        for (int i = i1 + tid; i < m; i += 256) {
            double x[kNB];
#pragma unroll
            for (int c = 0; c < kNB; ++c) x[c] = (c < nb) ? W[(size_t)i + (size_t)(j0 + c) * ldw] : 0.0;
#pragma unroll
            for (int c = 0; c < kNB; ++c) {
                if (c < nb) {
                    double v = x[c];
#pragma unroll
                    for (int k = 0; k < c; ++k) v = fma(-x[k], W[(size_t)(j0 + c) + (size_t)(j0 + k) * ldw], v);
                    x[c] = v / W[(size_t)(j0 + c) + (size_t)(j0 + c) * ldw];
                }
            }
#pragma unroll
            for (int c = 0; c < kNB; ++c) if (c < nb) W[(size_t)i + (size_t)(j0 + c) * ldw] = x[c];
        }
        __syncthreads();
        // (3) trailing update of the lower triangle: W[i][c] -= sum_k W[i][j0+k] W[c][j0+k]
        for (int c = i1 + wv; c < m; c += 4) {
            double lc[kNB];
#pragma unroll
            for (int k = 0; k < kNB; ++k) lc[k] = (k < nb) ? W[(size_t)c + (size_t)(j0 + k) * ldw] : 0.0;
            for (int i = c + lane; i < m; i += 64) {
                double v = W[(size_t)i + (size_t)c * ldw];
#pragma unroll
                for (int k = 0; k < kNB; ++k) if (k < nb) v = fma(-W[(size_t)i + (size_t)(j0 + k) * ldw], lc[k], v);
                W[(size_t)i + (size_t)c * ldw] = v;
            }
        }
        __syncthreads();
    }
    if (failed) {
        if (tid == 0) { if (status) atomicMin(&status[b], MPOPIS_ERR_NOT_PD); if (active) active[b] = 0; }
        return;
    }
    if (LDS) {
        for (int idx = tid; idx < n * n; idx += 256) {
            const int i = idx % n, j = idx / n;
            Lb[idx] = (i >= j) ? W[(size_t)i + (size_t)j * ldw] : 0.0;
        }
    }
}

void launch_potrf(const double* A, size_t Astride, double* L, int B, int n, const double* scale, int* status, int* active, hipStream_t s) {
    const int npad = (n + kNB - 1) / kNB * kNB;
    const size_t bytes = (size_t)npad * npad * sizeof(double);
    if (bytes <= 150 * 1024) {
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_potrf<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr_set = true; }
        hipLaunchKernelGGL(k_potrf<true>, dim3(B), dim3(256), bytes, s, A, Astride, L, n, npad, scale, status, active);
    } else {
        hipLaunchKernelGGL(k_potrf<false>, dim3(B), dim3(256), 0, s, A, Astride, L, n, n, scale, status, active);
    }
}

// g = Σ⁻¹ (γ U_orig) through the Cholesky factor (Σ symmetric => row vector γ U_orig' Σ⁻¹ = g').
// Slow path: only taken when α != 1 (γ != 0); no BASELINE config uses it.
__global__ void __launch_bounds__(256) k_chol_solve_gvec(const double* __restrict__ L, size_t Lstride, const double* __restrict__ Uorig,
                                                         double gamma, double* __restrict__ g, int n, const int* active) {
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

// ---------------------------------------------------------------------------------------------
// Weighted scatter / covariance:  S = (1/den) Σ_k w_k (X_k - μ)(X_k - μ)' + ridge I
//   X[b][cs][K] rows (K fastest); optional column gather idx[b][m] (elite order / PMC resampling);
//   w nullable (=1).  den: 0 => Σ_k w_k (ProbabilityWeights, uncorrected), else the given value.
// Stage 1: grid (lower tile pairs, ksplit, B), 256 threads = a 32x32 output tile held as 2x2 per
//          thread, K-chunks of 32 staged through LDS (coalesced 256-B row segments);
// Stage 2: deterministic reduction over the K splits, scaling, ridge, symmetrisation.
// ---------------------------------------------------------------------------------------------
constexpr int kCovT = 32, kCovKC = 32;
__global__ void __launch_bounds__(256) k_wcov_partial(const double* __restrict__ X, const double* __restrict__ w, const int32_t* __restrict__ idx,
                                                      const double* __restrict__ mu, double* __restrict__ part, int cs, int K, int m, int nt,
                                                      int ksplit, const int* active) {
    const int b = blockIdx.z;
    if (active && !active[b]) return;
    // decode lower-triangular tile pair
    int tp = blockIdx.x, ta = 0;
    while (tp >= ta + 1) { tp -= ta + 1; ++ta; }
    const int tb = tp;                               // ta >= tb
    const int a0 = ta * kCovT, b0 = tb * kCovT;
    __shared__ double sa[kCovKC][kCovT + 1], sb[kCovKC][kCovT + 1];
    const double* Xb = X + (size_t)b * cs * K;
    const double* wb = w ? w + (size_t)b * K : nullptr;
    const int32_t* ib = idx ? idx + (size_t)b * K : nullptr;   // index arrays are K long per slot
    const double* mub = mu + (size_t)b * cs;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;       // 16 x 16 threads, 2x2 outputs each
    double acc[2][2] = {{0, 0}, {0, 0}};
    const int per = (m + ksplit - 1) / ksplit;
    const int kbeg = blockIdx.y * per, kend = min(m, kbeg + per);
    const int lk = threadIdx.x & 31, lr = threadIdx.x >> 5;       // loader: 32 k x 8 rows per pass
    for (int k0 = kbeg; k0 < kend; k0 += kCovKC) {
        const int kq = k0 + lk;
        const bool kin = kq < kend;
        const int col = kin ? (ib ? ib[kq] : kq) : 0;
        const double wk = kin ? (wb ? wb[col] : 1.0) : 0.0;
        for (int r = lr; r < kCovT; r += 8) {
            const int ra = a0 + r, rb = b0 + r;
            const double xa = (kin && ra < cs) ? Xb[(size_t)ra * K + col] - mub[ra] : 0.0;
            const double xb = (kin && rb < cs) ? Xb[(size_t)rb * K + col] - mub[rb] : 0.0;
            sa[lk][r] = xa * wk;                      // (x_a - μ_a) w_k
            sb[lk][r] = xb;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < kCovKC; ++kk) {
            const double a_0 = sa[kk][ty], a_1 = sa[kk][ty + 16], b_0 = sb[kk][tx], b_1 = sb[kk][tx + 16];
            acc[0][0] = fma(a_0, b_0, acc[0][0]); acc[0][1] = fma(a_0, b_1, acc[0][1]);
            acc[1][0] = fma(a_1, b_0, acc[1][0]); acc[1][1] = fma(a_1, b_1, acc[1][1]);
        }
        __syncthreads();
    }
    const int ntp = nt * (nt + 1) / 2;
    double* pb = part + (((size_t)b * ksplit + blockIdx.y) * ntp + blockIdx.x) * (kCovT * kCovT);
    pb[(ty) * kCovT + tx] = acc[0][0]; pb[(ty) * kCovT + tx + 16] = acc[0][1];
    pb[(ty + 16) * kCovT + tx] = acc[1][0]; pb[(ty + 16) * kCovT + tx + 16] = acc[1][1];
}

__global__ void __launch_bounds__(256) k_wcov_finish(const double* __restrict__ part, const double* __restrict__ w, double* __restrict__ S,
                                                     int cs, int K, int nt, int ksplit, double den, double ridge, const int* active) {
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    __shared__ double sh[4];
    __shared__ double sden;
    if (den == 0.0) {                                  // Σ_k w_k
        double sacc = 0.0;
        for (int k = threadIdx.x; k < K; k += 256) sacc += w[(size_t)b * K + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sacc;
        __syncthreads();
        if (threadIdx.x == 0) sden = sh[0] + sh[1] + sh[2] + sh[3];
        __syncthreads();
    } else { if (threadIdx.x == 0) sden = den; __syncthreads(); }
    const double inv = 1 / sden;
    int tp = blockIdx.x, ta = 0;
    while (tp >= ta + 1) { tp -= ta + 1; ++ta; }
    const int tb = tp;
    const int ntp = nt * (nt + 1) / 2;
    for (int e = threadIdx.x; e < kCovT * kCovT; e += 256) {
        const int ia = e / kCovT, ibb = e % kCovT;
        const int ra = ta * kCovT + ia, rb = tb * kCovT + ibb;
        if (ra >= cs || rb >= cs) continue;
        double v = 0.0;
        for (int sp = 0; sp < ksplit; ++sp) v += part[(((size_t)b * ksplit + sp) * ntp + blockIdx.x) * (kCovT * kCovT) + e];
        v = v * inv;
        if (ra == rb) v += ridge;
        if (ta == tb && ibb > ia) continue;            // diagonal tile: keep lower triangle, mirrored below
        S[(size_t)b * cs * cs + (size_t)ra + (size_t)rb * cs] = v;
        S[(size_t)b * cs * cs + (size_t)rb + (size_t)ra * cs] = v;
    }
}

int wcov_num_tiles(int cs) { const int nt = (cs + kCovT - 1) / kCovT; return nt * (nt + 1) / 2; }
size_t wcov_workspace_doubles(int B, int cs, int ksplit) { return (size_t)B * ksplit * wcov_num_tiles(cs) * kCovT * kCovT; }

void launch_wcov(const double* X, const double* w, const int32_t* idx, int m, const double* mu, double* S, double* part,
                 int B, int cs, int K, int ksplit, double den, double ridge, const int* active, hipStream_t s) {
    const int nt = (cs + kCovT - 1) / kCovT, ntp = nt * (nt + 1) / 2;
    hipLaunchKernelGGL(k_wcov_partial, dim3(ntp, ksplit, B), dim3(256), 0, s, X, w, idx, mu, part, cs, K, m, nt, ksplit, active);
    hipLaunchKernelGGL(k_wcov_finish, dim3(ntp, B), dim3(256), 0, s, part, w, S, cs, K, nt, ksplit, den, ridge, active);
}

// mean over gathered columns: mu[r] = (1/m) Σ_j X[r][idx[j]] ; optionally weighted by cw[j] (CMA δw, no division)
__global__ void __launch_bounds__(256) k_gather_mean(const double* __restrict__ X, const int32_t* __restrict__ idx, const double* __restrict__ cw,
                                                     double* __restrict__ mu, size_t mu_stride, int cs, int K, int m, int divide, const int* active) {
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
