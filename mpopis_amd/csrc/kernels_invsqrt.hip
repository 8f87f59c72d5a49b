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

constexpr int kTB = 16;                      // block size of the triangular inverse
constexpr int kInvThreads = 1024, kInvWaves = kInvThreads / 64;

// L(r, c) of the n x n column-major factor, padded with the identity beyond n
__device__ __forceinline__ double ld_L(const double* __restrict__ L, int n, int r, int c) {
    return (r < n && c < n) ? L[r + (size_t)c * n] : ((r == c) ? 1.0 : 0.0);
}

}  // namespace

// part[b][J] = sum of squares of the entries of block column J of L^-1 (rows/cols < n)
__global__ void __launch_bounds__(kInvThreads) k_trtri_fro(const double* __restrict__ Lall, size_t Lstride, int n, int nb, double* __restrict__ part,
                                                           const int* active) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.y, J = blockIdx.x;
    if (active && !active[b]) return;
    extern __shared__ __attribute__((aligned(16))) double sh_inv[];
    double* Xs = sh_inv;                               // [(nb - J)][16][16]   block column of X = L^-1 (row-major blocks)
    double* Ws = Xs + (size_t)(nb - J) * 256;          // [kInvWaves][256]     per-wave partial products
    double* Wsum = Ws + kInvWaves * 256;               // [256]
    double* Dinv = Wsum + 256;                         // [256]  inverse of the current diagonal block
    double* Ld = Dinv + 256;                           // [256]  the diagonal block itself
    double* red = Ld + 256;                            // [kInvWaves]
    const double* L = Lall + (size_t)b * Lstride;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int I = J; I < nb; ++I) {
        // ---- inverse of the diagonal block L(I,I): wave 0, lane c < 16 = column c (forward substitution, unit rhs)
        if (tid < 256) Ld[tid] = ld_L(L, n, 16 * I + (tid >> 4), 16 * I + (tid & 15));          // Ld[i*16 + k] = L(I,I)[i][k]
        // ---- partial products P_wv = sum_{K = J+wv, J+wv+16, ...  < I} L(I,K) X(K)      lane = (row i, column group g)
        {
            const int i = lane & 15, g = lane >> 4;
            double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
            for (int K = J + wv; K < I; K += kInvWaves) {
                const double* xk = Xs + (size_t)(K - J) * 256 + 4 * g;
#pragma unroll 4
                for (int k = 0; k < 16; ++k) {
                    const double l = ld_L(L, n, 16 * I + i, 16 * K + k);
                    p0 = fma(l, xk[k * 16 + 0], p0); p1 = fma(l, xk[k * 16 + 1], p1);
                    p2 = fma(l, xk[k * 16 + 2], p2); p3 = fma(l, xk[k * 16 + 3], p3);
                }
            }
            double* w = Ws + wv * 256 + i * 16 + 4 * g;
            w[0] = p0; w[1] = p1; w[2] = p2; w[3] = p3;
        }
        __syncthreads();
        if (wv == 0 && lane < 16) {                                                              // column c of L(I,I)^-1, rows in order
            const int c = lane;
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                double s = (i == c) ? 1.0 : 0.0;
#pragma unroll 1
                for (int k = 0; k < i; ++k) s = fma(-Ld[i * 16 + k], Dinv[k * 16 + c], s);       // own column only: program order suffices
                Dinv[i * 16 + c] = s / Ld[i * 16 + i];
            }
        }
        if (tid >= 64 && tid < 64 + 256) {                                                        // reduce the partial products (waves 1..4)
            const int e = tid - 64;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kInvWaves; ++w) s += Ws[w * 256 + e];
            Wsum[e] = s;
        }
        __syncthreads();
        if (tid < 256) {                                                                          // X(I) = (I == J) ? Dinv : -Dinv * W
            const int i = tid >> 4, c = tid & 15;
            double v;
            if (I == J) v = Dinv[tid];
            else {
                v = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) if (k <= i) v = fma(-Dinv[i * 16 + k], Wsum[k * 16 + c], v);
            }
            Xs[(size_t)(I - J) * 256 + tid] = v;
        }
        __syncthreads();
    }
    double s = 0.0;
    for (int e = tid; e < (nb - J) * 256; e += kInvThreads) {
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
constexpr int kLanThreads = 1024, kLanWaves = kLanThreads / 64;
constexpr int kLanNQ = 5;                    // rows per lane and pass of the mat-vec (64*5 = 320 >= cs = 300 in one pass)
constexpr double kLanTol = 1e-13;

// y[b] = A[b]^-1/2 bvec[b];  fro[b] = scale[b] * sum_J part[b][J]  (= tr(A^-1) = ||A^-1/2||_F^2)
// status: MPOPIS_ERR_NUMERIC when the spectrum bounds are unusable (non-finite input, M/m beyond 1e14)
__global__ void __launch_bounds__(kLanThreads) k_lanczos_invsqrt(const double* __restrict__ Aall, const double* __restrict__ bvec, size_t bstride,
                                                                 const double* __restrict__ part, int nb, const double* __restrict__ scale,
                                                                 double* __restrict__ Vall, double* __restrict__ yall, double* __restrict__ fro_out,
                                                                 int* __restrict__ msteps, int n, int* status, const int* active) {
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
    __shared__ int sh_flag;
    const double* A = Aall + (size_t)b * n * n;
    const double* bv = bvec + (size_t)b * bstride;
    double* V = Vall + (size_t)b * (size_t)(n + 1 + 64) * n;   // basis vectors v_0 .. v_n
    double* upiv = V + (size_t)(n + 1) * n;                    // [n][64] trailing pivots of the quadrature solves, lane-private columns
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
    auto block_max = [&](double v) -> double {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
        __syncthreads();
        if (lane == 0) red[wv] = v;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kLanWaves; ++w) t = fmax(t, red[w]);
        return t;
    };
    // ---- spectrum bounds: M = ||A||_inf >= λ_max;  λ_min >= 1/tr(A^-1) -------------------------------------------------
    double fro = 0.0;
    for (int J = 0; J < nb; ++J) fro += part[(size_t)b * nb + J];
    fro *= scale ? scale[b] : 1.0;
    double mx = 0.0;
    for (int j = wv; j < n; j += kLanWaves) {
        double s = 0.0;
        for (int i = lane; i < n; i += 64) s += fabs(A[i + (size_t)j * n]);
        mx = fmax(mx, wave_sum(s));
    }
    const double Mhi = block_max(mx);
    const double mlo = fmin(1.0 / fro, 0.5 * Mhi);
    if (tid == 0) fro_out[b] = fro;
    // quadrature node of this lane (wave 0 uses it; identical in every wave)
    double q_shift = 0.0, q_weight = 0.0;
    const bool q_ok = invsqrt_quad_node(mlo, Mhi, lane, 64, &q_shift, &q_weight);
    // ---- v_0 = b / ||b|| --------------------------------------------------------------------------------------------------
    double s2 = 0.0;
    for (int i = tid; i < n; i += kLanThreads) { const double v = bv[i]; s2 = fma(v, v, s2); }
    const double nb2 = block_sum(s2);
    const double nrm_b = sqrt(nb2);
    if (!(nrm_b > 0.0) || !q_ok || !(fro > 0.0)) {                   // δw = 0 -> y = 0; unusable bounds -> numeric error
        for (int i = tid; i < n; i += kLanThreads) y[i] = 0.0;
        if (tid == 0) { msteps[b] = 0; if (nrm_b > 0.0 || !(nrm_b == nrm_b)) status[b] = MPOPIS_ERR_NUMERIC; }
        return;
    }
    for (int i = tid; i < n; i += kLanThreads) { const double v = bv[i] / nrm_b; vcur[i] = v; V[i] = v; }
    __syncthreads();
    const int mcap = n;
    int m = 0;
    for (int j = 0; j < mcap; ++j) {
        // ---- w = A v_j : lanes along rows (coalesced column reads), waves split the columns ----------------------------------
        for (int r0 = 0; r0 < n; r0 += 64 * kLanNQ) {
            double acc[kLanNQ];
#pragma unroll
            for (int q = 0; q < kLanNQ; ++q) acc[q] = 0.0;
            for (int c = wv; c < n; c += kLanWaves) {
                const double vc = vcur[c];
                const double* col = A + (size_t)c * n + r0 + lane;
#pragma unroll
                for (int q = 0; q < kLanNQ; ++q) if (r0 + lane + 64 * q < n) acc[q] = fma(col[64 * q], vc, acc[q]);
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
        __syncthreads();
        // ---- orthogonalise against v_0..v_j twice (classical Gram-Schmidt, CGS2); α_j = the v_j coefficient -------------------
        double a_j = 0.0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int k = wv; k <= j; k += kLanWaves) {
                double s = 0.0;
                for (int i = lane; i < n; i += 64) s = fma(V[(size_t)k * n + i], wv_[i], s);
                s = wave_sum(s);
                if (lane == 0) coef[k] = s;
            }
            __syncthreads();
            a_j += coef[j];
            for (int i = tid; i < n; i += kLanThreads) {
                double s = wv_[i];
                for (int k = 0; k <= j; ++k) s = fma(-coef[k], V[(size_t)k * n + i], s);
                wv_[i] = s;
            }
            __syncthreads();
        }
        double w2 = 0.0;
        for (int i = tid; i < n; i += kLanThreads) { const double v = wv_[i]; w2 = fma(v, v, w2); }
        const double bt = sqrt(block_sum(w2));
        if (tid == 0) { alpha[j] = a_j; beta[j] = bt; }
        __syncthreads();
        m = j + 1;
        // ---- c = T_m^-1/2 e_1 by quadrature + error bound (wave 0, lane = node); O(m) per check, so checked at every step while
        //      m is small and every 4th / 16th step later (a generic dense Σ needs m ~ n steps, CMA's Σ converges long before)
        const bool check = m <= 32 || ((m & 3) == 0 && m <= 128) || (m & 15) == 0 || m >= mcap || bt <= 1e-14 * Mhi;
        if (!check) { if (tid == 0) sh_flag = 0; }
        else if (wv == 0) {
            double u = 0.0;
            for (int i = m - 1; i >= 0; --i) {                              // trailing pivots of T_m + s I (SPD: all > 0)
                const double bi = (i + 1 < m) ? beta[i] : 0.0;
                u = alpha[i] + q_shift - ((i + 1 < m) ? bi * bi / u : 0.0);
                upiv[i * 64 + lane] = u;
            }
            double z = 1.0 / upiv[lane], cn2 = 0.0;
            for (int i = 0; i < m; ++i) {
                if (i > 0) z = -beta[i - 1] * z / upiv[i * 64 + lane];
                const double ci = wave_sum(q_weight * z);
                if (lane == 0) cvec[i] = ci;
                cn2 = fma(ci, ci, cn2);
            }
            const double eb = bt * wave_sum(q_weight * fabs(z) / (mlo + q_shift));   // ||y - y_m|| / ||b||
            const bool conv = (eb <= kLanTol * sqrt(cn2)) || (bt <= 1e-14 * Mhi) || (m >= mcap);
            if (lane == 0) sh_flag = conv ? 1 : 0;
        }
        __syncthreads();
        if (sh_flag) break;
        for (int i = tid; i < n; i += kLanThreads) { const double v = wv_[i] / bt; vcur[i] = v; V[(size_t)(j + 1) * n + i] = v; }
        __syncthreads();
    }
    for (int i = tid; i < n; i += kLanThreads) {
        double s = 0.0;
        for (int k = 0; k < m; ++k) s = fma(cvec[k], V[(size_t)k * n + i], s);
        y[i] = nrm_b * s;
    }
    if (tid == 0) msteps[b] = m;
}

size_t invsqrt_workspace_doubles(int B, int n) { return (size_t)B * (size_t)(n + 1 + 64) * n; }
int invsqrt_max_n() {
    // dynamic LDS of k_lanczos_invsqrt: (kLanWaves + 6) n + ... doubles, and of k_trtri_fro: 16 n + 5376 doubles, both <= 150 KiB
    return (int)((150 * 1024 / 8 - 64) / (kLanWaves + 6));
}

// y = A^-1/2 b and fro = tr(A^-1) per slot, from A (n x n, SPD) and the Cholesky factor L of scale*A (scale: per-slot, nullable)
void launch_invsqrt_vec(const double* A, const double* L, size_t Lstride, const double* scale, const double* bvec, size_t bstride,
                        double* part, double* V, double* y, double* fro, int* msteps, int B, int n, int* status, const int* active, hipStream_t s) {
    const int nb = (n + kTB - 1) / kTB;
    const size_t lds1 = ((size_t)nb * 256 + kInvWaves * 256 + 4 * 256 + kInvWaves) * sizeof(double);
    static std::atomic<unsigned long long> seen1{0}, seen2{0};
    ensure_dyn_lds((const void*)k_trtri_fro, 150 * 1024, seen1);
    hipLaunchKernelGGL(k_trtri_fro, dim3(nb, B), dim3(kInvThreads), lds1, s, L, Lstride, n, nb, part, active);
    const size_t lds2 = ((size_t)(kLanWaves + 6) * n + 1 + kLanWaves + 4) * sizeof(double);
    ensure_dyn_lds((const void*)k_lanczos_invsqrt, 150 * 1024, seen2);
    hipLaunchKernelGGL(k_lanczos_invsqrt, dim3(B), dim3(kLanThreads), lds2, s, A, bvec, bstride, part, nb, scale, V, y, fro, msteps, n, status, active);
}

}  // namespace mpopis
