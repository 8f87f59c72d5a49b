// kernels_cma.hip -- device side of CMAMPPI_Policy's adaptation step (src/mppi_mpopi_policies.jl:571-599):
//   C = Σ^-0.5                      :580   (LinearAlgebra: Hermitian eigen path -> principal inverse sqrt)
//   pσ, σ, hσ, pΣ updates            :581-586
//   temp_sum (scalar, quirk)        :588-596
//   Σ update + triu symmetrisation   :598-599
// Σ^-0.5 is computed with the coupled Newton-Schulz iteration (Higham, "Functions of Matrices", eq. 6.35)
//   Y0 = A/c, Z0 = I;  T = (3I - Z Y)/2;  Y <- Y T;  Z <- T Z;   Y -> (A/c)^1/2, Z -> (A/c)^-1/2
// which is all GEMM (batched FP64, LDS-tiled) and converges quadratically for SPD A once
// c >= λmax (c = ||A||_inf).  It reaches the same matrix the eigen path does, to ~cond(A)*eps.
#include "engine.h"

namespace mpopis {

constexpr int kGT = 32;     // output tile
constexpr int kGK = 32;     // contraction chunk

// D = alpha * (A * B) + beta * I   for n x n column-major matrices, batched (stride n*n).
// If resid != nullptr: atomically tracks max |I - A*B| per batch entry (as ordered uint64 bits).
// done_prev (nullable): entries whose previous residual is below tol are copied through (D = passthru).
__global__ void __launch_bounds__(256) k_gemm_nn(const double* __restrict__ A, const double* __restrict__ Bm, double* __restrict__ D,
                                                 const double* __restrict__ passthru, int n, double alpha, double beta,
                                                 unsigned long long* resid, const unsigned long long* resid_prev, double tol,
                                                 const int* active) {
    const int b = blockIdx.z;
    if (active && !active[b]) return;
    const size_t off = (size_t)b * n * n;
    const int i0 = blockIdx.x * kGT, j0 = blockIdx.y * kGT;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    if (resid_prev && __longlong_as_double((long long)resid_prev[b]) < tol) {        // converged: pass through
        if (passthru) {
            for (int e = threadIdx.x; e < kGT * kGT; e += 256) {
                const int i = i0 + (e % kGT), j = j0 + (e / kGT);
                if (i < n && j < n) D[off + i + (size_t)j * n] = passthru[off + i + (size_t)j * n];
            }
        }
        if (resid && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) resid[b] = resid_prev[b];
        return;
    }
    __shared__ double sa[kGK][kGT + 1], sb[kGK][kGT + 1];
    double acc[2][2] = {{0, 0}, {0, 0}};
    for (int k0 = 0; k0 < n; k0 += kGK) {
        for (int e = threadIdx.x; e < kGT * kGK; e += 256) {
            const int ii = e % kGT, kk = e / kGT;                 // A tile: rows i0+ii (fast), cols k0+kk
            sa[kk][ii] = (i0 + ii < n && k0 + kk < n) ? A[off + (i0 + ii) + (size_t)(k0 + kk) * n] : 0.0;
            const int kb = e % kGK, jj = e / kGK;                 // B tile: rows k0+kb (fast), cols j0+jj
            sb[kb][jj] = (k0 + kb < n && j0 + jj < n) ? Bm[off + (k0 + kb) + (size_t)(j0 + jj) * n] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < kGK; ++kk) {
            const double a0 = sa[kk][tx], a1 = sa[kk][tx + 16], b0 = sb[kk][ty], b1 = sb[kk][ty + 16];
            acc[0][0] = fma(a0, b0, acc[0][0]); acc[0][1] = fma(a0, b1, acc[0][1]);
            acc[1][0] = fma(a1, b0, acc[1][0]); acc[1][1] = fma(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }
    double rmax = 0.0;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = i0 + tx + 16 * p, j = j0 + ty + 16 * q;
            if (i < n && j < n) {
                const double ab = acc[p][q];
                rmax = fmax(rmax, fabs(((i == j) ? 1.0 : 0.0) - ab));
                D[off + i + (size_t)j * n] = alpha * ab + ((i == j) ? beta : 0.0);
            }
        }
    if (resid) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, o, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(&resid[b], (unsigned long long)__double_as_longlong(rmax));
    }
}

__global__ void k_ns_resid_init(unsigned long long* r, int B, size_t n) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i < n) r[i] = (i < (size_t)B) ? 0x7FF0000000000000ull : 0ull;
}

// c[b] = ||A||_inf ; Y0 = A / c ; Z0 = I
__global__ void __launch_bounds__(256) k_ns_init(const double* __restrict__ A, double* __restrict__ Y, double* __restrict__ Z, double* cnorm,
                                                 int n, const int* active) {
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    __shared__ double sh[4];
    __shared__ double c_sh;
    const size_t off = (size_t)b * n * n;
    double mx = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += fabs(A[off + i + (size_t)j * n]);
        mx = fmax(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) { c_sh = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3])); cnorm[b] = c_sh; }
    __syncthreads();
    const double inv = 1.0 / c_sh;
    for (int e = threadIdx.x; e < n * n; e += 256) {
        Y[off + e] = A[off + e] * inv;
        Z[off + e] = ((e % n) == (e / n)) ? 1.0 : 0.0;
    }
}

// C = Z / sqrt(c)   (Z ~ (A/c)^-1/2)
__global__ void __launch_bounds__(256) k_ns_finish(const double* __restrict__ Z, const double* cnorm, double* __restrict__ C, int n, const int* active) {
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const size_t e = blockIdx.x * (size_t)256 + threadIdx.x;
    if (e < (size_t)n * n) C[(size_t)b * n * n + e] = Z[(size_t)b * n * n + e] / sqrt(cnorm[b]);
}

void launch_inv_sqrt_spd(const double* A, double* C, double* Y0, double* Y1, double* Z0, double* Z1, double* Tm, double* cnorm,
                         unsigned long long* resid /* [iters+1][B] zeroed by this call */, int B, int n, int iters,
                         const int* active, hipStream_t s) {
    // resid[it][b]: max|I - ZY| seen by iteration it (ordered uint64 bits); row 0 = +inf ("not converged")
    hipLaunchKernelGGL(k_ns_resid_init, dim3(((size_t)(iters + 1) * B + 255) / 256), dim3(256), 0, s, resid, B, (size_t)(iters + 1) * B);
    hipLaunchKernelGGL(k_ns_init, dim3(B), dim3(256), 0, s, A, Y0, Z0, cnorm, n, active);
    const dim3 grid((n + kGT - 1) / kGT, (n + kGT - 1) / kGT, B);
    const double tol = 1e-14;
    double *Yc = Y0, *Yn = Y1, *Zc = Z0, *Zn = Z1;
    for (int it = 0; it < iters; ++it) {
        unsigned long long* rprev = resid + (size_t)it * B;
        unsigned long long* rnew = resid + (size_t)(it + 1) * B;
        // T = 1.5 I - 0.5 Z Y  (+ residual max|I - ZY|)
        hipLaunchKernelGGL(k_gemm_nn, grid, dim3(256), 0, s, Zc, Yc, Tm, (const double*)nullptr, n, -0.5, 1.5, rnew, rprev, tol, active);
        // Y' = Y T ; Z' = T Z   (pass-through copies once converged)
        hipLaunchKernelGGL(k_gemm_nn, grid, dim3(256), 0, s, Yc, Tm, Yn, Yc, n, 1.0, 0.0, (unsigned long long*)nullptr, rprev, tol, active);
        hipLaunchKernelGGL(k_gemm_nn, grid, dim3(256), 0, s, Tm, Zc, Zn, Zc, n, 1.0, 0.0, (unsigned long long*)nullptr, rprev, tol, active);
        std::swap(Yc, Yn); std::swap(Zc, Zn);
    }
    hipLaunchKernelGGL(k_ns_finish, dim3(((size_t)n * n + 255) / 256, B), dim3(256), 0, s, Zc, cnorm, C, n, active);
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
__global__ void __launch_bounds__(256) k_cma_paths(const double* __restrict__ C, const double* __restrict__ E, const int32_t* __restrict__ order,
                                                   const double* __restrict__ ws, double* Ucur, double* scal, double* vec, double* sig2,
                                                   int cs, int K, int n_iter, CmaConsts cc, const int* active) {
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    __shared__ double sh[4];
    __shared__ double bc[2];
    auto block_sum = [&](double v) -> double {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        return sh[0] + sh[1] + sh[2] + sh[3];
    };
    const size_t nn = (size_t)cs * cs;
    const double* Cb = C + (size_t)b * nn;
    double* ps = vec + (size_t)b * 3 * cs; double* pS = ps + cs; double* dw = pS + cs;
    double* Ub = Ucur + (size_t)b * cs;
    const double sigma_old = scal[b * 8 + 0];
    const double sc = sqrt(cc.c_sigma * (2 - cc.c_sigma) * cc.mu_eff);
    double nps2 = 0.0, fro = 0.0;
    for (int i = threadIdx.x; i < cs; i += 256) {
        Ub[i] += sigma_old * dw[i];                                                        // :577
        double v = 0.0;
        for (int j = 0; j < cs; ++j) { const double cij = Cb[i + (size_t)j * cs]; v = fma(sc * cij, dw[j], v); fro = fma(cij, cij, fro); }
        const double pn = (1 - cc.c_sigma) * ps[i] + v;                                    // :581
        ps[i] = pn; nps2 = fma(pn, pn, nps2);
    }
    nps2 = block_sum(nps2);
    fro = block_sum(fro);
    const double nps = sqrt(nps2);
    const double sigma_new = sigma_old * exp(cc.c_sigma / cc.d_sigma * (nps / cc.E_cma - 1));   // :582
    const int h_sigma = (nps / sqrt(1 - pow(1 - cc.c_sigma, 2.0 * n_iter)) < (1.4 + 2.0 / (cs + 1)) * cc.E_cma) ? 1 : 0;   // :585
    const double sS = h_sigma * sqrt(cc.c_Sigma * (2 - cc.c_Sigma) * cc.mu_eff);
    for (int i = threadIdx.x; i < cs; i += 256) pS[i] = (1 - cc.c_Sigma) * pS[i] + sS * dw[i];   // :586
    // temp_sum (:588-596): δs[order[ii]] is LINEAR indexing into δs = elite_E/σ (cs x m_elite), a scalar
    const double* Eb = E + (size_t)b * cs * K;
    const int32_t* ob = order + (size_t)b * K;
    double ts = 0.0;
    for (int ii = threadIdx.x; ii < K; ii += 256) {
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
    (void)bc;
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
    hipLaunchKernelGGL(k_cma_paths, dim3(B), dim3(256), 0, s, C, E, order, ws, Ucur, scal, vec, sig2, cs, K, n_iter, cc, active);
}
void launch_cma_sigma_update(double* Sig, const double* scal, const double* vec, int B, int cs, const double* consts7, int m_elite,
                             const int* active, hipStream_t s) {
    CmaConsts cc{consts7[0], consts7[1], consts7[2], consts7[3], consts7[4], consts7[5], consts7[6], m_elite};
    hipLaunchKernelGGL(k_cma_sigma_update, dim3(((size_t)cs * cs + 255) / 256, B), dim3(256), 0, s, Sig, scal, vec, cs, cc, active);
}

}  // namespace mpopis
