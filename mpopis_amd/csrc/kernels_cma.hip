// kernels_cma.hip -- device side of CMAMPPI_Policy's adaptation step (src/mppi_mpopi_policies.jl:571-599):
//   C = Σ^-0.5                      :580   (LinearAlgebra: Hermitian eigen path -> principal inverse sqrt)
//   pσ, σ, hσ, pΣ updates            :581-586
//   temp_sum (scalar, quirk)        :588-596
//   Σ update + triu symmetrisation   :598-599
// Σ^-0.5 is computed with the coupled Newton-Schulz iteration (Higham, "Functions of Matrices", eq. 6.35)
//   Y0 = A/c, Z0 = I;  T = (3I - Z Y)/2;  Y <- Y T;  Z <- T Z;   Y -> (A/c)^1/2, Z -> (A/c)^-1/2
// which is all GEMM (batched FP64 on the matrix cores, kernels_mfma.hip; every iterate is a polynomial in A, hence
// symmetric) and converges quadratically for SPD A once c >= λmax (c = ||A||_inf).  It reaches the same matrix the
// eigen path does, to ~cond(A)*eps.
#include "engine.h"

namespace mpopis {

__global__ void k_ns_resid_init(unsigned long long* r, int B, size_t n) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i < n) r[i] = (i < (size_t)B) ? 0x7FF0000000000000ull : 0ull;
}

// c[b] = ||A||_inf (A symmetric: max column abs sum; one wave per column, coalesced)
__global__ void __launch_bounds__(1024) k_ns_norm(const double* __restrict__ A, double* cnorm, int n, const int* active) {
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    __shared__ double sh[16];
    const size_t off = (size_t)b * n * n;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double mx = 0.0;
    for (int j = wv; j < n; j += 16) {
        double s = 0.0;
        for (int i = lane; i < n; i += 64) s += fabs(A[off + i + (size_t)j * n]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        mx = fmax(mx, s);
    }
    if (lane == 0) sh[wv] = mx;
    __syncthreads();
    if (threadIdx.x == 0) { double c = 0.0; for (int q = 0; q < 16; ++q) c = fmax(c, sh[q]); cnorm[b] = c; }
}
// Y0 = A / c ; Z0 = I
__global__ void __launch_bounds__(256) k_ns_init(const double* __restrict__ A, double* __restrict__ Y, double* __restrict__ Z, const double* cnorm,
                                                 int n, const int* active) {
    const int b = blockIdx.z, j = blockIdx.y;
    if (active && !active[b]) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t e = (size_t)b * n * n + i + (size_t)j * n;
    Y[e] = A[e] * (1.0 / cnorm[b]);
    Z[e] = (i == j) ? 1.0 : 0.0;
}

// C = Z / sqrt(c)   (Z ~ (A/c)^-1/2).  Z lives in Z0 (even iterations) or Z1 (odd): the slot froze at the first
// iteration `it` whose incoming residual was below tol; if it never converged the last iterate is used.
__global__ void __launch_bounds__(256) k_ns_finish(const double* __restrict__ Z0, const double* __restrict__ Z1, const double* cnorm,
                                                   const unsigned long long* resid, int iters, int B, double tol,
                                                   double* __restrict__ C, int n, const int* active) {
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    int it = 0;
    while (it < iters && !(__longlong_as_double((long long)resid[(size_t)it * B + b]) < tol)) ++it;
    const double* Z = (it & 1) ? Z1 : Z0;
    const size_t e = blockIdx.x * (size_t)256 + threadIdx.x;
    if (e < (size_t)n * n) C[(size_t)b * n * n + e] = Z[(size_t)b * n * n + e] / sqrt(cnorm[b]);
}

void launch_inv_sqrt_spd(const double* A, double* C, double* Y0, double* Y1, double* Z0, double* Z1, double* Tm, double* cnorm,
                         unsigned long long* resid /* [iters+1][B] zeroed by this call */, int B, int n, int iters,
                         const int* active, hipStream_t s) {
    // resid[it][b]: max|I - ZY| seen by iteration it (ordered uint64 bits); row 0 = +inf ("not converged")
    hipLaunchKernelGGL(k_ns_resid_init, dim3(((size_t)(iters + 1) * B + 255) / 256), dim3(256), 0, s, resid, B, (size_t)(iters + 1) * B);
    hipLaunchKernelGGL(k_ns_norm, dim3(B), dim3(1024), 0, s, A, cnorm, n, active);
    hipLaunchKernelGGL(k_ns_init, dim3((n + 255) / 256, n, B), dim3(256), 0, s, A, Y0, Z0, cnorm, n, active);
    const double tol = 64.0 * n * 1.1e-16;                     // rounding floor of max|I - ZY| grows with n
    double *Yc = Y0, *Yn = Y1, *Zc = Z0, *Zn = Z1;
    for (int it = 0; it < iters; ++it) {
        unsigned long long* rprev = resid + (size_t)it * B;
        unsigned long long* rnew = resid + (size_t)(it + 1) * B;
        // T = 1.5 I - 0.5 Z Y  (+ residual max|I - ZY|)
        launch_gemm_sym_mfma(Zc, Yc, Tm, B, n, -0.5, 1.5, rnew, rprev, tol, active, s);
        // Y' = Y T ; Z' = T Z   (slots freeze once converged)
        // (T is a polynomial in Z Y, the iterates are symmetric and commute: T Z = Z T, so both products share the right operand)
        launch_gemm_sym_mfma_pair(Yc, Zc, Tm, Yn, Zn, B, n, rprev, tol, active, s);
        std::swap(Yc, Yn); std::swap(Zc, Zn);
    }
    hipLaunchKernelGGL(k_ns_finish, dim3(((size_t)n * n + 255) / 256, B), dim3(256), 0, s, Z0, Z1, cnorm, resid, iters, B, tol, C, n, active);
}

// Per-slot CMA scalars (d_cma_scal[b][8]): [0] σ  [1] temp_sum  [2] hσ  [3] ||pσ||  [4] ||C||_F²
// vectors (d_cma_vec[b][3*cs]): pσ | pΣ | δw.   sig2[b] = σ² (scale of the next proposal, :551).
struct CmaConsts { double mu_eff, c_sigma, d_sigma, c_Sigma, c1, c_mu, E_cma; int m_elite; };

__global__ void __launch_bounds__(256) k_cma_begin(double* scal, double* vec, double* sig2, double sigma0, int cs, int B) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < 3 * cs; i += 256) vec[(size_t)b * 3 * cs + i] = 0.0;        // pσ, pΣ = 0 (:545)
    if (threadIdx.x == 0) { scal[b * 8 + 0] = sigma0; sig2[b] = sigma0 * sigma0; }              // σ = pol.σ (:536)
}

// δw given (gather_mean with cw); this kernel: pol.U += σ δw; pσ; σ; hσ; pΣ; temp_sum
constexpr int kCmaThreads = 1024, kCmaWaves = kCmaThreads / 64;
__global__ void __launch_bounds__(kCmaThreads) k_cma_paths(const double* __restrict__ C, const double* __restrict__ E, const int32_t* __restrict__ order,
                                                   const double* __restrict__ ws, double* Ucur, double* scal, double* vec, double* sig2,
                                                   int cs, int K, int n_iter, CmaConsts cc, const int* active) {
    extern __shared__ __attribute__((aligned(16))) double part_v[];          // [kCmaWaves][cs] partial C*δw
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    __shared__ double sh[kCmaWaves];
    auto block_sum = [&](double v) -> double {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kCmaWaves; ++w) t += sh[w];
        return t;
    };
    const size_t nn = (size_t)cs * cs;
    const double* Cb = C + (size_t)b * nn;
    double* ps = vec + (size_t)b * 3 * cs; double* pS = ps + cs; double* dw = pS + cs;
    double* Ub = Ucur + (size_t)b * cs;
    const double sigma_old = scal[b * 8 + 0];
    const double sc = sqrt(cc.c_sigma * (2 - cc.c_sigma) * cc.mu_eff);
    double nps2 = 0.0, fro = 0.0;
    // C*δw and ||C||_F²: lanes run along rows i (coalesced column reads), the 16 waves split the columns j
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int jper = (cs + kCmaWaves - 1) / kCmaWaves, j0 = wv * jper, j1 = min(cs, j0 + jper);
        for (int rc = 0; rc < cs; rc += 64) {
            const int i = rc + lane;
            double v = 0.0;
            if (i < cs) {
                for (int j = j0; j < j1; ++j) { const double cij = Cb[i + (size_t)j * cs]; v = fma(sc * cij, dw[j], v); fro = fma(cij, cij, fro); }
                part_v[(size_t)wv * cs + i] = v;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cs; i += kCmaThreads) {
        Ub[i] += sigma_old * dw[i];                                                        // :577
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kCmaWaves; ++w) v += part_v[(size_t)w * cs + i];
        const double pn = (1 - cc.c_sigma) * ps[i] + v;                                    // :581
        ps[i] = pn; nps2 = fma(pn, pn, nps2);
    }
    nps2 = block_sum(nps2);
    fro = block_sum(fro);
    const double nps = sqrt(nps2);
    const double sigma_new = sigma_old * exp(cc.c_sigma / cc.d_sigma * (nps / cc.E_cma - 1));   // :582
    const int h_sigma = (nps / sqrt(1 - pow(1 - cc.c_sigma, 2.0 * n_iter)) < (1.4 + 2.0 / (cs + 1)) * cc.E_cma) ? 1 : 0;   // :585
    const double sS = h_sigma * sqrt(cc.c_Sigma * (2 - cc.c_Sigma) * cc.mu_eff);
    for (int i = threadIdx.x; i < cs; i += kCmaThreads) pS[i] = (1 - cc.c_Sigma) * pS[i] + sS * dw[i];   // :586
    // temp_sum (:588-596): δs[order[ii]] is LINEAR indexing into δs = elite_E/σ (cs x m_elite), a scalar
    const double* Eb = E + (size_t)b * cs * K;
    const int32_t* ob = order + (size_t)b * K;
    double ts = 0.0;
    for (int ii = threadIdx.x; ii < K; ii += kCmaThreads) {
        const int j = ob[ii];                                    // 0-based linear index, requires j < cs*m_elite
        const double d = Eb[(size_t)(j % cs) * K + ob[j / cs]] / sigma_old;
        const double wi = ws[ii];
        double w0;
        if (wi >= 0) w0 = wi;
        else { const double nc = sqrt((d * d) * fro); w0 = n_iter * wi / (nc * nc); }        // norm(C*δ)^2, n = iteration index
        ts += w0 * d * d;
    }
    ts = block_sum(ts);
    if (threadIdx.x == 0) {
        scal[b * 8 + 0] = sigma_new; scal[b * 8 + 1] = ts; scal[b * 8 + 2] = (double)h_sigma; scal[b * 8 + 3] = nps; scal[b * 8 + 4] = fro;
        sig2[b] = sigma_new * sigma_new;
    }
}

// Σ = (1-c1-cμ)Σ + c1 (pΣ pΣ' + (1-hσ) cΣ (2-cΣ) Σ) .+ cμ temp_sum ; Σ = triu(Σ) + triu(Σ,1)'   (:598-599)
__global__ void __launch_bounds__(256) k_cma_sigma_update(double* Sig, const double* scal, const double* vec, int cs, CmaConsts cc, const int* active) {
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const size_t e = blockIdx.x * (size_t)256 + threadIdx.x;
    if (e >= (size_t)cs * cs) return;
    const int i = e % cs, j = e / cs;
    if (i > j) return;                                           // upper triangle drives both halves
    double* S = Sig + (size_t)b * cs * cs;
    const double* pS = vec + (size_t)b * 3 * cs + cs;
    const double ts = scal[b * 8 + 1];
    const int h = (int)scal[b * 8 + 2];
    const double s_old = S[i + (size_t)j * cs];
    const double v = (1 - cc.c1 - cc.c_mu) * s_old + cc.c1 * (pS[i] * pS[j] + (1 - h) * cc.c_Sigma * (2 - cc.c_Sigma) * s_old) + cc.c_mu * ts;
    S[i + (size_t)j * cs] = v;
    S[j + (size_t)i * cs] = v;
}

void launch_cma_begin(double* scal, double* vec, double* sig2, double sigma0, int cs, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_cma_begin, dim3(B), dim3(256), 0, s, scal, vec, sig2, sigma0, cs, B);
}
void launch_cma_paths(const double* C, const double* E, const int32_t* order, const double* ws, double* Ucur, double* scal, double* vec,
                      double* sig2, int B, int cs, int K, int n_iter, const double* consts7, int m_elite, const int* active, hipStream_t s) {
    CmaConsts cc{consts7[0], consts7[1], consts7[2], consts7[3], consts7[4], consts7[5], consts7[6], m_elite};
    hipLaunchKernelGGL(k_cma_paths, dim3(B), dim3(kCmaThreads), (size_t)kCmaWaves * cs * sizeof(double), s, C, E, order, ws, Ucur, scal, vec, sig2, cs, K, n_iter, cc, active);
}
void launch_cma_sigma_update(double* Sig, const double* scal, const double* vec, int B, int cs, const double* consts7, int m_elite,
                             const int* active, hipStream_t s) {
    CmaConsts cc{consts7[0], consts7[1], consts7[2], consts7[3], consts7[4], consts7[5], consts7[6], m_elite};
    hipLaunchKernelGGL(k_cma_sigma_update, dim3(((size_t)cs * cs + 255) / 256, B), dim3(256), 0, s, Sig, scal, vec, cs, cc, active);
}

}  // namespace mpopis
