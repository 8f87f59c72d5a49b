// kernels_cma.hip -- device side of CMAMPPI_Policy's adaptation step (src/mppi_mpopi_policies.jl:571-599):
//   C = Σ^-0.5                      :580   (LinearAlgebra: Hermitian eigen path -> principal inverse sqrt)
//   pσ, σ, hσ, pΣ updates            :581-586
//   temp_sum (scalar, quirk)        :588-596
//   Σ update + triu symmetrisation   :598-599
// The matrix C itself is never formed: the reference only consumes the vector C*δw and the scalar ||C||_F (see
// kernels_invsqrt.hip, which delivers both); this file holds the path / step-size / covariance updates.
#include "engine.h"

namespace mpopis {

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
__global__ void __launch_bounds__(kCmaThreads) k_cma_paths(const double* __restrict__ Cdw, const double* __restrict__ froC, const double* __restrict__ E, const int32_t* __restrict__ order,
                                                   const double* __restrict__ ws, double* Ucur, double* scal, double* vec, double* sig2,
                                                   int cs, int K, int n_iter, CmaConsts cc, const int* active) {
    MPOPIS_HI_PRIO();
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
    const double* yb = Cdw + (size_t)b * cs;                  // Σ^-0.5 δw
    double* ps = vec + (size_t)b * 3 * cs; double* pS = ps + cs; double* dw = pS + cs;
    double* Ub = Ucur + (size_t)b * cs;
    const double sigma_old = scal[b * 8 + 0];
    const double sc = sqrt(cc.c_sigma * (2 - cc.c_sigma) * cc.mu_eff);
    double nps2 = 0.0;
    const double fro = froC[b];                               // ||C||_F² = tr(Σ^-1)
    for (int i = threadIdx.x; i < cs; i += kCmaThreads) {
        Ub[i] += sigma_old * dw[i];                                                        // :577
        const double pn = (1 - cc.c_sigma) * ps[i] + sc * yb[i];                           // :581
        ps[i] = pn; nps2 = fma(pn, pn, nps2);
    }
    nps2 = block_sum(nps2);
    const double nps = sqrt(nps2);
    const double sigma_new = sigma_old * exp(cc.c_sigma / cc.d_sigma * (nps / cc.E_cma - 1));   // :582
    const int h_sigma = (nps / sqrt(1 - pow(1 - cc.c_sigma, 2.0 * n_iter)) < (1.4 + 2.0 / (cs + 1)) * cc.E_cma) ? 1 : 0;   // :585
    const double sS = h_sigma * sqrt(cc.c_Sigma * (2 - cc.c_Sigma) * cc.mu_eff);
    for (int i = threadIdx.x; i < cs; i += kCmaThreads) pS[i] = (1 - cc.c_Sigma) * pS[i] + sS * dw[i];   // :586
    // temp_sum (:588-596): δs[order[ii]] is LINEAR indexing into δs = elite_E/σ (cs x m_elite), a scalar
    const double* Eb = E + (size_t)b * cs * K;
    const int32_t* ob = order + (size_t)b * K;
    double ts = 0.0;
    // order[ii] -> order[j / cs] -> E[...] are three dependent global round trips per term: issue them four terms at a time (K = 4096 is four
    // terms per thread: 12 serial round trips as a plain loop, ~14 us); same summation order as the plain loop
    for (int i0 = threadIdx.x; i0 < K; i0 += 4 * kCmaThreads) {
        int j[4], col[4]; double ev[4], wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int ii = min(i0 + u * kCmaThreads, K - 1); j[u] = ob[ii]; wv[u] = ws[ii]; }   // 0-based linear index, requires j < cs*m_elite
#pragma unroll
        for (int u = 0; u < 4; ++u) col[u] = ob[j[u] / cs];
#pragma unroll
        for (int u = 0; u < 4; ++u) ev[u] = Eb[(size_t)(j[u] % cs) * K + col[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * kCmaThreads < K) {
                const double d = ev[u] / sigma_old;
                const double wi = wv[u];
                double w0;
                if (wi >= 0) w0 = wi;
                else { const double nc = sqrt((d * d) * fro); w0 = n_iter * wi / (nc * nc); }        // norm(C*δ)^2, n = iteration index
                ts += w0 * d * d;
            }
        }
    }
    ts = block_sum(ts);
    if (threadIdx.x == 0) {
        scal[b * 8 + 0] = sigma_new; scal[b * 8 + 1] = ts; scal[b * 8 + 2] = (double)h_sigma; scal[b * 8 + 3] = nps; scal[b * 8 + 4] = fro;
        sig2[b] = sigma_new * sigma_new;
    }
}

// Σ = (1-c1-cμ)Σ + c1 (pΣ pΣ' + (1-hσ) cΣ (2-cΣ) Σ) .+ cμ temp_sum ; Σ = triu(Σ) + triu(Σ,1)'   (:598-599)
__global__ void __launch_bounds__(256) k_cma_sigma_update(double* Sig, const double* scal, const double* vec, int cs, CmaConsts cc, const int* active) {
    MPOPIS_HI_PRIO();
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
void launch_cma_paths(const double* Cdw, const double* fro, const double* E, const int32_t* order, const double* ws, double* Ucur, double* scal, double* vec,
                      double* sig2, int B, int cs, int K, int n_iter, const double* consts7, int m_elite, const int* active, hipStream_t s) {
    CmaConsts cc{consts7[0], consts7[1], consts7[2], consts7[3], consts7[4], consts7[5], consts7[6], m_elite};
    hipLaunchKernelGGL(k_cma_paths, dim3(B), dim3(kCmaThreads), 0, s, Cdw, fro, E, order, ws, Ucur, scal, vec, sig2, cs, K, n_iter, cc, active);
}
void launch_cma_sigma_update(double* Sig, const double* scal, const double* vec, int B, int cs, const double* consts7, int m_elite,
                             const int* active, hipStream_t s) {
    CmaConsts cc{consts7[0], consts7[1], consts7[2], consts7[3], consts7[4], consts7[5], consts7[6], m_elite};
    hipLaunchKernelGGL(k_cma_sigma_update, dim3(((size_t)cs * cs + 255) / 256, B), dim3(256), 0, s, Sig, scal, vec, cs, cc, active);
}

}  // namespace mpopis
