// engine_ais.hip -- the adaptive-importance-sampling updates between iterations (n < N):
//   :imppi / :μaismppi   μ′ = Σ_k ws_k E_k                      src/mppi_mpopi_policies.jl:361-365, :659-663
//   :μΣaismppi           + Σ′ = Σ_k ws_k (E_k-μ′)(E_k-μ′)' + 10e-9 I     :729-734
//   :pmcmppi             multinomial resampling via alias table, moments of E[:,idx]   :802-809
//   :cemppi              elite selection, early break, Σ′ = cov(est, elite') + 10e-9 I  :453-466
//   :cmamppi             CMA-ES style path/σ/Σ adaptation (with the reference's quirks) :561-600
#include "engine.h"
#include "engine_handle.h"

using namespace mpopis;

namespace mpopis {

__global__ void __launch_bounds__(256) k_scale_rows(double* Z, const double* dsc, int cs, int K) {
    const int b = blockIdx.z, r = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k < K) Z[((size_t)b * cs + r) * K + k] *= dsc[(size_t)b * cs + r];
}
void launch_scale_rows(double* Z, const double* dsc, int B, int cs, int K, hipStream_t s) {
    hipLaunchKernelGGL(k_scale_rows, dim3((K + 255) / 256, cs, B), dim3(256), 0, s, Z, dsc, cs, K);
}

__global__ void __launch_bounds__(256) k_add_active(const double* x, double* y, int n, const int* active) {   // y[b] += x[b]
    const int b = blockIdx.y; if (active && !active[b]) return;
    const int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    y[(size_t)b * n + i] = y[(size_t)b * n + i] + x[(size_t)b * n + i];
}

}  // namespace mpopis

void mpopis_handle::init_cma_constants() {}
void mpopis_handle::cma_begin() {}
const double* mpopis_handle::cma_sigma2() { return nullptr; }

int mpopis_handle::ais_update(int n, bool injected) {
    (void)n; (void)injected;
    const int pol = cfg.policy;
    if (pol == MPOPIS_POL_IMPPI || pol == MPOPIS_POL_MUAISMPPI || pol == MPOPIS_POL_MUSIGMAAISMPPI) {
        const double lam = (pol == MPOPIS_POL_IMPPI) ? cfg.lambda : cfg.lambda_ais;          // :362 / :647,:712
        time_begin(3);
        launch_weights(d_cost, d_w, B, K, lam, d_active, d_status, stream);
        launch_wmean(d_E, d_w, nullptr, nullptr, d_mu, B, cs, K, 1, d_active, stream);     // μ′ (mean(E, pw, dims=2))
        time_end();
        if (pol == MPOPIS_POL_MUSIGMAAISMPPI) {
            time_begin(4);
            launch_wcov(d_E, d_w, nullptr, K, d_mu, d_Sig, d_part, B, cs, K, ksplit, 0.0, 10e-9, d_active, stream);
            time_end();
        }
        hipLaunchKernelGGL(k_add_active, dim3((cs + 255) / 256, B), dim3(256), 0, stream, d_mu, d_Ucur, cs, d_active);   // pol.U += μ′
        return MPOPIS_OK;
    }
    err = "policy update not implemented yet";
    return MPOPIS_ERR_ARG;
}

int mpopis_handle::run_trials(int, int, double*, double*) { err = "run_trials not implemented yet"; return MPOPIS_ERR_ARG; }
