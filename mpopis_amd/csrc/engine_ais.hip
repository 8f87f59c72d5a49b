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
    MPOPIS_HI_PRIO();
    const int b = blockIdx.z, r = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k < K) Z[((size_t)b * cs + r) * K + k] *= dsc[(size_t)b * cs + r];
}
void launch_scale_rows(double* Z, const double* dsc, int B, int cs, int K, hipStream_t s) {
    hipLaunchKernelGGL(k_scale_rows, dim3((K + 255) / 256, cs, B), dim3(256), 0, s, Z, dsc, cs, K);
}

__global__ void __launch_bounds__(256) k_add_active(const double* x, double* y, int n, const int* active) {
    MPOPIS_HI_PRIO();   // y[b] += x[b]
    const int b = blockIdx.y; if (active && !active[b]) return;
    const int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    y[(size_t)b * n + i] = y[(size_t)b * n + i] + x[(size_t)b * n + i];
}

}  // namespace mpopis

// CMAMPPI_Policy constructor constants: src/mppi_mpopi_policies.jl:513-525
void mpopis_handle::init_cma_constants() {
    const int m = K; const double n = (double)cs;
    m_elite = (int)nearbyint((1.0 - cfg.elite_threshold) * m);
    cma_ws_host.resize(m);
    for (int i = 1; i <= m; ++i) cma_ws_host[i - 1] = log((m + 1) / 2.0) - log((double)i);
    double s = 0.0;
    for (int i = 0; i < m_elite; ++i) s += cma_ws_host[i];
    for (int i = 0; i < m_elite; ++i) cma_ws_host[i] /= s;
    double s2 = 0.0;
    for (int i = 0; i < m_elite; ++i) s2 += cma_ws_host[i] * cma_ws_host[i];
    const double mu_eff = 1 / s2;
    const double c_sigma = (mu_eff + 2) / (n + mu_eff + 5);
    const double d_sigma = 1 + 2 * fmax(0.0, sqrt((mu_eff - 1) / (n + 1)) - 1) + c_sigma;
    const double c_Sigma = (4 + mu_eff / n) / (n + 4 + 2 * mu_eff / n);
    const double c1 = 2 / ((n + 1.3) * (n + 1.3) + mu_eff);
    const double c_mu = fmin(1 - c1, 2 * (mu_eff - 2 + 1 / mu_eff) / ((n + 2) * (n + 2) + mu_eff));
    double st = 0.0;
    for (int i = m_elite; i < m; ++i) st += cma_ws_host[i];
    const double f = -(1 + c1 / c_mu) / st;
    for (int i = m_elite; i < m; ++i) cma_ws_host[i] *= f;
    const double E_cma = sqrt(n) * (1 - 1 / (4 * n) + 1 / (21 * (n * n)));
    const double c[7] = {mu_eff, c_sigma, d_sigma, c_Sigma, c1, c_mu, E_cma};
    for (int i = 0; i < 7; ++i) cma_consts[i] = c[i];
}
void mpopis_handle::cma_begin() { launch_cma_begin(d_cma_scal, d_cma_vec, d_sig2, cfg.cma_sigma, cs, B, stream); }
const double* mpopis_handle::cma_sigma2() { return d_sig2; }

int mpopis_handle::ais_update(int n, bool injected) {
    const int pol = cfg.policy;
    if (pol == MPOPIS_POL_IMPPI || pol == MPOPIS_POL_MUAISMPPI || pol == MPOPIS_POL_MUSIGMAAISMPPI) {
        const double lam = (pol == MPOPIS_POL_IMPPI) ? cfg.lambda : cfg.lambda_ais;          // :362 / :647,:712
        // μ′, Σ′ = mean_and_cov(E, pw, 2) (:730-733).  μΣ-AIS: one pass over E yields both (ones row in the MFMA scatter)
        const bool one_pass = pol == MPOPIS_POL_MUSIGMAAISMPPI && wcov_mfma_can_emit_mean(cs);
        // :μΣaismppi on a car env: ws = compute_weights(IT(λ_ais), cost) (:712) is evaluated inside the moments kernel from the costs and the
        // minimum the rollout kernel accumulated -- no reweighting launch between rollout and moments in the AIS iterations
        const bool fold = one_pass && weights_in_moments;
        if (!fold) {
            time_begin(3);
            launch_weights(d_cost, d_w, B, K, lam, d_active, d_status, stream, d_wsum);
            if (!one_pass) launch_wmean(d_E, d_w, nullptr, nullptr, d_mu, B, cs, K, 1, d_active, stream);     // μ′ (mean(E, pw, dims=2))
            time_end();
        }
        if (pol == MPOPIS_POL_MUSIGMAAISMPPI) {
            time_begin(4);
            launch_wcov_mfma(d_E, d_w, nullptr, K, d_mu, d_Sig, d_part, B, cs, K, ksplit, wcov_sel_batch(), 0.0, 10e-9, d_active, stream, nullptr,
                             one_pass ? d_mu : nullptr, one_pass ? d_Ucur : nullptr, d_wsum,
                             fold ? d_cost : nullptr, fold ? d_cmin : nullptr, -1 / lam);        // one pass: also pol.U += μ′
            time_end();
        }
        if (!one_pass) hipLaunchKernelGGL(k_add_active, dim3((cs + 255) / 256, B), dim3(256), 0, stream, d_mu, d_Ucur, cs, d_active);   // pol.U += μ′
        return MPOPIS_OK;
    }
    if (pol == MPOPIS_POL_PMCMPPI) {                                                          // :802-809
        time_begin(3);
        launch_weights(d_cost, d_w, B, K, cfg.lambda_ais, d_active, d_status, stream);
        time_end();
        time_begin(5);
        launch_alias_build(d_w, d_accept, d_alias, B, K, d_active, stream, d_alias_need, d_alias_stack);                   // Categorical(ws) -> AliasTable
        const int32_t* di; const double* du; size_t stride;
        if (injected) { di = d_resi_in + (size_t)(n - 1) * K; du = d_resu_in + (size_t)(n - 1) * K; stride = (size_t)(N - 1) * K; }
        else {
            launch_sample_resample_draws(d_resi, d_resu, B, K, d_seeds, (uint32_t)mpc_step, (uint32_t)(n - 1) | 0x80000000u, d_active, stream);
            di = d_resi; du = d_resu; stride = (size_t)K;
        }
        launch_alias_sample(d_accept, d_alias, di, stride, du, d_order, d_residx_log + (size_t)(n - 1) * K, (size_t)(N - 1) * K, B, K, d_active, stream);
        time_end();
        time_begin(4);
        if (wcov_mfma_can_emit_mean(cs)) {
            // E′ = E[:, idx] materialised once (d_Z is free between two sampling phases; the Z prefetch is off for this policy), then
            // (μ′, Σ′) = mean_and_cov(E′, 2) (corrected, :807) in ONE pass over the contiguous E′: ones row for the mean, no per-element gather
            // (shifted by its first column -- d_gvec is free here: γ's row is rebuilt per iteration -- so that the one-pass moments do not cancel
            // when the resampled set collapses onto a few columns; the finish kernel adds the shift back to μ′)
            launch_gather_cols(d_E, d_order, d_Z, B, cs, K, d_active, stream, d_gvec);
            launch_wcov_mfma(d_Z, nullptr, nullptr, K, d_mu, d_Sig, d_part, B, cs, K, ksplit, wcov_sel_batch(), (double)(K - 1), 10e-9, d_active, stream, nullptr,
                             d_mu, d_Ucur, nullptr, nullptr, nullptr, 0.0, d_gvec);            // also pol.U += μ′ (:809)
            time_end();
            return MPOPIS_OK;
        }
        launch_gather_mean(d_E, d_order, nullptr, d_mu, B, cs, K, K, 1, d_active, stream);   // mean_and_cov(E[:,idx], 2): corrected
        launch_wcov_mfma(d_E, nullptr, d_order, K, d_mu, d_Sig, d_part, B, cs, K, ksplit, wcov_sel_batch(), (double)(K - 1), 10e-9, d_active, stream);
        time_end();
        hipLaunchKernelGGL(k_add_active, dim3((cs + 255) / 256, B), dim3(256), 0, stream, d_mu, d_Ucur, cs, d_active);
        return MPOPIS_OK;
    }
    if (pol == MPOPIS_POL_CEMPPI || pol == MPOPIS_POL_CMAMPPI) {
        const bool side = side_free && pol == MPOPIS_POL_CMAMPPI;
        if (side && !trtri_early) {
            // tr(Σ^-1) = σ² ||L^-1||_F² needs only L = chol(σ²Σ): a latency-bound kernel of a few workgroups, run beside the equally
            // latency-bound sort / elite mean on the free second stream (large batches; small ones start it right behind the Cholesky)
            if (!fork_recorded) (void)hipEventRecord(ev_fork, stream);          // the Z prefetch may already have marked the point behind the rollout
            (void)hipStreamWaitEvent(xstream[0], ev_fork, 0);
            // (no `active` predicate on the side stream: launch_sortperm below clears active[b] for the early break concurrently; an
            // inactive slot's trace is never consumed, so computing it is merely wasted, deterministic work)
            launch_trtri_fro(cur_L, cur_Lstride, d_fro_part, B, cs, nullptr, xstream[0], d_tri_dinv, true, d_Sig, cur_L_scaled ? d_sig2 : nullptr, d_lan_prep, d_tri_cnt);
            (void)hipEventRecord(ev_join[0], xstream[0]);
        }
        static const int env_small = [] { const char* e = getenv("MPOPIS_CE_SMALL"); return e ? atoi(e) : 1; }();      // 0: always the general path; 2: small kernel, sort in its own launch (A/B, tests)
        const bool ce_small = pol == MPOPIS_POL_CEMPPI && env_small && ce_cov_small_ok(cs, m_elite, cfg.sigma_est);
        const bool ce_sorts = ce_small && env_small != 2 && ce_sort_fusable(K);           // the CE kernel sorts (and breaks) itself
        if (!ce_sorts) {
            time_begin(5);
            // (:pmcmppi's alias-table buffers are free under :cemppi / :cmamppi: scratch of the chip-wide rank sort)
            launch_sortperm(d_cost, d_order, B, K, m_elite, d_active, stream, d_accept, d_alias_need);   // :455 / :563 and the early break :458-461 / :566-569
            time_end();
        }
        if (pol == MPOPIS_POL_CEMPPI) {                                                       // :464-465
            if (ce_small) {
                time_begin(4);
                launch_ce_cov_small(d_E, d_order, d_mu, d_Sig, d_Ucur, B, cs, K, m_elite, cfg.sigma_est, 10e-9, d_active, stream, ce_sorts ? d_cost : nullptr);
                time_end();
                return MPOPIS_OK;
            }
            time_begin(4);
            launch_gather_mean(d_E, d_order, nullptr, d_mu, B, cs, K, m_elite, 1, d_active, stream);
            if (cfg.sigma_est == MPOPIS_SIGMA_EST_RBLW || cfg.sigma_est == MPOPIS_SIGMA_EST_OAS) {   // DiagonalCommonVariance :421-423
                launch_wcov_mfma(d_E, nullptr, d_order, m_elite, d_mu, d_Sig, d_part, B, cs, K, ksplit, wcov_sel_batch(), (double)m_elite, 0.0, d_active, stream);
                launch_common_shrink(d_Sig, B, cs, m_elite, cfg.sigma_est == MPOPIS_SIGMA_EST_OAS, 10e-9, d_active, stream);
            } else if (cfg.sigma_est == MPOPIS_SIGMA_EST_SS || cfg.sigma_est == MPOPIS_SIGMA_EST_LW) {   // DiagonalUnequalVariance :417-419
                launch_wcov_mfma(d_E, nullptr, d_order, m_elite, d_mu, d_Sig, d_part, B, cs, K, ksplit, wcov_sel_batch(), (double)m_elite, 0.0, d_active, stream);
                if (cfg.sigma_est == MPOPIS_SIGMA_EST_SS) launch_inv_sd(d_Sig, d_gvec, B, cs, d_active, stream);   // d_gvec is free here (γ row is rebuilt per iteration)
                else launch_fill_f64(d_gvec, 1.0, (size_t)B * cs, stream);                       // :lw = same intensity on the unstandardised data
                launch_wcov_mfma(d_E, nullptr, d_order, m_elite, d_mu, d_tmpS, d_part, B, cs, K, ksplit, wcov_sel_batch(), 1.0, 0.0, d_active, stream, d_gvec);
                launch_ss_shrink(d_Sig, d_tmpS, d_gvec, B, cs, m_elite, 10e-9, d_active, stream);
            } else {
                launch_wcov_mfma(d_E, nullptr, d_order, m_elite, d_mu, d_Sig, d_part, B, cs, K, ksplit, wcov_sel_batch(), (double)m_elite, 10e-9, d_active, stream);
            }
            time_end();
            hipLaunchKernelGGL(k_add_active, dim3((cs + 255) / 256, B), dim3(256), 0, stream, d_mu, d_Ucur, cs, d_active);
            return MPOPIS_OK;
        }
        time_begin(4);
        double* dw = d_cma_vec + 2 * (size_t)cs;                                              // δw slot of slot 0; stride 3cs
        launch_gather_mean_strided(d_E, d_order, d_cma_ws, dw, (size_t)3 * cs, B, cs, K, m_elite, d_active, stream);   // :573-576
        // C = Σ^-0.5 (:580) is only consumed as C*δw (:581) and ||C||_F (:593): neither needs the matrix.  cur_L = chol(Σ) or chol(σ²Σ) is the
        // factor this iteration sampled from (same Σ: the update :598 comes after), so tr(Σ^-1) = ||L^-1||_F² or σ² ||L^-1||_F²
        if (side) (void)hipStreamWaitEvent(stream, ev_join[0], 0);
        else launch_trtri_fro(cur_L, cur_Lstride, d_fro_part, B, cs, d_active, stream, d_tri_dinv, true, d_Sig, cur_L_scaled ? d_sig2 : nullptr, d_lan_prep, d_tri_cnt);
        launch_lanczos_invsqrt(d_Sig, d_lan_prep, dw, (size_t)3 * cs, d_lanV, d_Cdw, d_fro, d_lan_m, B, cs, d_status, d_active, stream,
                               lan_regions, lan_coop());
        launch_cma_paths(d_Cdw, d_fro, d_E, d_order, d_cma_ws, d_Ucur, d_cma_scal, d_cma_vec, d_sig2, B, cs, K, n, cma_consts, m_elite, d_active, stream);
        launch_cma_sigma_update(d_Sig, d_cma_scal, d_cma_vec, B, cs, cma_consts, m_elite, d_active, stream);
        time_end();
        return MPOPIS_OK;
    }
    err = "policy update not implemented";
    return MPOPIS_ERR_ARG;
}

