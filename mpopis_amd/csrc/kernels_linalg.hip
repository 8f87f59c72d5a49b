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
// Blocked right-looking Cholesky, one workgroup (4 waves) per matrix, panel width 16:
//   (1) wave 0 factors the 16x16 diagonal block in registers (lane = row; v_readlane broadcasts,
//       rsqrt + multiply instead of sqrt + divide),
//   (2) one thread per row solves the panel below it (forward substitution against L11, reciprocal
//       diagonal),
//   (3) the rank-16 trailing update runs on the matrix cores: per 16x16 tile 4 x v_mfma_f64_16x16x4
//       (A = panel rows of the tile row, B = panel rows of the tile column), read-modify-write.
// The working copy lives in LDS when it fits (npad^2*8 <= 150 KiB, i.e. cs <= 128: all 1-car configs),
// otherwise in the output buffer in global memory (L2 resident; cs = 300 for 3 cars).
// scale[b] (nullable) multiplies A first (CMA: MvNormal(σ²Σ), :551).  On a non-positive pivot
// status[b] = MPOPIS_ERR_NOT_PD and active[b] = 0 (the reference throws PosDefException).
// ---------------------------------------------------------------------------------------------
typedef double v4f64_l __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double bcast_lane(double v, int src) {     // src is a compile-time constant after unrolling
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

template <bool LDS, int NTHR>
__global__ void __launch_bounds__(NTHR) k_potrf(const double* __restrict__ A, size_t Astride, double* __restrict__ Lout,
                                               int n, int npad, const double* scale, int* status, int* active) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int failed;
    __shared__ double rdiag[kNB];
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double* Ab = A + (size_t)b * Astride;
    double* Lb = Lout + (size_t)b * n * n;
    const double sc = scale ? scale[b] : 1.0;
    // working matrix W (ld = ldw): the LDS copy is padded to a multiple of 16 with an identity tail
    double* W; int ldw, m;
    if (LDS) { W = smem; ldw = npad; m = npad; } else { W = Lb; ldw = n; m = n; }
    // global-memory variant: the solved panel strip P[m][17] (both operands of the trailing update) and the current diagonal
    // block D11[16][17] (read by every row of the panel solve) live in LDS; only the trailing read-modify-write goes to L2
    constexpr int kPS = kNB + 1;
    double* P = LDS ? nullptr : smem;
    double* D11 = LDS ? nullptr : smem + (size_t)n * kPS;
    if (tid == 0) failed = 0;
    // copy in the lower triangle only (the upper triangle of W is never consumed): columns c and m-1-c together hold m+1
    // entries, so the triangle is the (m/2) x (m+1) rectangle e -> (c, t); batches of 8 unconditional (clamped) loads in
    // flight per thread, select afterwards.  (m even: npad is a multiple of 16; the global-memory variant keeps m = n.)
    if (LDS) {
        constexpr int NW = NTHR / 64;
        // column pair c (columns c and m-1-c) holds m+1 triangle entries t: wave -> pairs, lane -> t (no integer divisions);
        // 4 pairs x 2 lane chunks = 8 unconditional (clamped) loads in flight per thread, select afterwards
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
    } else {
        for (int e0 = tid; e0 < m * m; e0 += NTHR * 8) {
            double av[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = min(e0 + u * NTHR, m * m - 1), i = e % m, j = e / m;
                av[u] = Ab[(size_t)min(i, n - 1) + (size_t)min(j, n - 1) * n];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * NTHR;
                if (e < m * m) {
                    const int i = e % m, j = e / m;
                    W[(size_t)i + (size_t)j * ldw] = (i >= j) ? sc * av[u] : 0.0;
                }
            }
        }
    }
    __syncthreads();

    const int li = lane & 15, lk = lane >> 4;
    constexpr int NW = NTHR / 64;
    // (1) diagonal block, one wave, lane = row (rows/cols >= nb behave as identity)
    auto factor_diag = [&](int j0, int nb) {
        double d[kNB];
        const bool rowok = lane < nb;
#pragma unroll
        for (int c = 0; c < kNB; ++c) d[c] = (rowok && c < nb) ? W[(size_t)(j0 + lane) + (size_t)(j0 + c) * ldw] : ((lane == c) ? 1.0 : 0.0);
        bool bad = false;
#pragma unroll
        for (int jj = 0; jj < kNB; ++jj) {
            const double piv = bcast_lane(d[jj], jj);
            if (!(piv > 0.0)) bad = true;
            double rs = __builtin_amdgcn_rsq(piv);              // v_rsq_f64 seed + 2 Newton steps (full precision)
            rs = rs * fma(-0.5 * piv * rs, rs, 1.5);
            rs = rs * fma(-0.5 * piv * rs, rs, 1.5);
            const double ljj = piv * rs;                        // sqrt(piv)
            if (lane == jj) { d[jj] = ljj; rdiag[jj] = rs; } else if (lane > jj) d[jj] = d[jj] * rs;
#pragma unroll
            for (int c = jj + 1; c < kNB; ++c) {
                const double lcj = bcast_lane(d[jj], c);
                if (lane >= c) d[c] = fma(-d[jj], lcj, d[c]);
            }
        }
        if (rowok) {
#pragma unroll
            for (int c = 0; c < kNB; ++c) if (c < nb && c <= lane) {
                W[(size_t)(j0 + lane) + (size_t)(j0 + c) * ldw] = d[c];
                if (!LDS) D11[lane * kPS + c] = d[c];
            }
        }
        if (bad && lane == 0) failed = 1;
    };
    // (3) one 16x16 tile of the rank-16 trailing update on the matrix cores: pair q -> tile (t1 + ta, t1 + tb), ta >= tb
    auto trail_pair = [&](int j0, int t1, int q) {
        int ta = 0, qq = q;
        while (qq >= ta + 1) { qq -= ta + 1; ++ta; }
        const int r0 = (t1 + ta) * 16, c0 = (t1 + qq) * 16;
        v4f64_l acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ra = r0 + li, rb = c0 + li, col = j0 + kk * 4 + lk;
            const double av = (ra < m) ? (LDS ? W[(size_t)ra + (size_t)col * ldw] : P[(size_t)ra * kPS + kk * 4 + lk]) : 0.0;
            const double bv = (rb < m) ? (LDS ? W[(size_t)rb + (size_t)col * ldw] : P[(size_t)rb * kPS + kk * 4 + lk]) : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc, 0, 0, 0);     // transposed tile: lanes run along i
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = r0 + li, c = c0 + lk + 4 * r;                               // D'[c - c0][i - r0]
            if (i < m && c < m && i >= c) W[(size_t)i + (size_t)c * ldw] -= acc[r];
        }
    };
    if (wv == 0) factor_diag(0, min(kNB, m));
    __syncthreads();
    for (int j0 = 0; j0 < m; j0 += kNB) {
        if (failed) break;
        const int nb = min(kNB, m - j0);
        const int i1 = j0 + nb;              // first row below the panel
        // (2) panel solve: row i of L21 = A21[i,:] * L11^-T
        for (int i = i1 + tid; i < m; i += NTHR) {
            double x[kNB];
#pragma unroll
            for (int c = 0; c < kNB; ++c) x[c] = (c < nb) ? W[(size_t)i + (size_t)(j0 + c) * ldw] : 0.0;
#pragma unroll
            for (int c = 0; c < kNB; ++c) {
                if (c < nb) {
                    double v = x[c];
#pragma unroll
                    for (int k = 0; k < c; ++k) v = fma(-x[k], LDS ? W[(size_t)(j0 + c) + (size_t)(j0 + k) * ldw] : D11[c * kPS + k], v);
                    x[c] = v * rdiag[c];
                }
            }
#pragma unroll
            for (int c = 0; c < kNB; ++c) if (c < nb) {
                W[(size_t)i + (size_t)(j0 + c) * ldw] = x[c];
                if (!LDS) P[(size_t)i * kPS + c] = x[c];
            }
        }
        __syncthreads();
        // (3) trailing update with one panel of look-ahead: wave 0 updates the next diagonal tile first and factors it
        // right away, while the other waves update the remaining tiles (rows/cols beyond the panel)
        const int t1 = i1 / 16;                                  // i1 is a multiple of 16 except after the last (partial) panel
        const int ntile = (m + 15) / 16 - t1;
        if (nb == kNB && ntile > 0) {
            const int npair = ntile * (ntile + 1) / 2;
            if (wv == 0) {
                trail_pair(j0, t1, 0);
                if (!LDS) __threadfence_block();
                factor_diag(i1, min(kNB, m - i1));
                if (NW == 1) for (int q = 1; q < npair; ++q) trail_pair(j0, t1, q);
            } else {
                for (int q = wv; q < npair; q += NW - 1) trail_pair(j0, t1, q);
            }
        }
        __syncthreads();
    }
    if (failed) {
        if (tid == 0) { if (status) atomicMin(&status[b], MPOPIS_ERR_NOT_PD); if (active) active[b] = 0; }
        return;
    }
    if (LDS) {
        for (int j = wv; j < n; j += NTHR / 64)
            for (int i = lane; i < n; i += 64) Lb[(size_t)i + (size_t)j * n] = (i >= j) ? W[(size_t)i + (size_t)j * ldw] : 0.0;
    }
}

void launch_potrf(const double* A, size_t Astride, double* L, int B, int n, const double* scale, int* status, int* active, hipStream_t s) {
    const int npad = (n + kNB - 1) / kNB * kNB;
    const size_t bytes = (size_t)npad * npad * sizeof(double);
    if (bytes <= 150 * 1024) {
        static std::atomic<unsigned long long> seen{0};
        ensure_dyn_lds((const void*)k_potrf<true, 512>, 150 * 1024, seen);
        hipLaunchKernelGGL((k_potrf<true, 512>), dim3(B), dim3(512), bytes, s, A, Astride, L, n, npad, scale, status, active);
    } else {
        const size_t strip = ((size_t)n * (kNB + 1) + kNB * (kNB + 1)) * sizeof(double);     // panel strip + diagonal block
        static std::atomic<unsigned long long> seen2{0};
        ensure_dyn_lds((const void*)k_potrf<false, 1024>, 150 * 1024, seen2);
        hipLaunchKernelGGL((k_potrf<false, 1024>), dim3(B), dim3(1024), strip, s, A, Astride, L, n, n, scale, status, active);
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
