// engine_handle.h -- the opaque handle behind include/mpopis.h (see engine.h for the HBM layout).
#pragma once
#include "engine.h"

struct mpopis_handle {
    mpopis_config cfg{};
    int B = 0, K = 0, T = 0, as = 0, ss = 0, cs = 0, N = 1;
    double gamma = 0.0;
    hipStream_t stream = nullptr;
    // Extra streams.  Part-chain schedule (auto_parts / mpopis_set_overlap, policy_step_enqueue): the batch as 2..4 part-chains, the latency-bound
    // links of one chain (Cholesky, weights, finish kernels: tens of workgroups on 256 CUs) under the others' throughput kernels.  When the batch
    // runs as one chain the same streams carry side chains: ||L^-1||_F of the CMA update (xstream[0]) and the Z prefetch for cs > 128 (xstream[1])
    static constexpr int kMaxSplit = 4;
    hipStream_t xstream[kMaxSplit - 1] = {nullptr, nullptr, nullptr};          // streams of the 2nd .. 4th part
    bool split_pinned = false;   // MPOPIS_NSPLIT set: the schedule is fixed for the process
    const double* cur_L = nullptr; size_t cur_Lstride = 0;   // the factor the current AIS iteration samples from (shared L0 or per-slot d_L)
    bool cur_L_scaled = false;                               // ... factors σ²Σ′ (:cmamppi from the second iteration on) rather than Σ′
    bool trtri_early = false;    // this iteration's ||L^-1||_F launch already sits on xstream[0] (queued right behind the Cholesky)
    bool fork_recorded = false;  // ev_fork already recorded behind the current iteration's rollout
    bool side_free = false;      // batch not split this step: xstream[0] / ev_fork / ev_join[0] carry the CMA side chain (||L^-1||_F beside sort + elite mean)
    hipEvent_t ev_fork = nullptr, ev_join[kMaxSplit - 1] = {nullptr, nullptr, nullptr}, ev_skew[kMaxSplit - 1] = {nullptr, nullptr, nullptr};
    int nsplit = 1;                                                            // parts the batch is split into when split_auto is off
    bool split_auto = true;                                                    // default schedule (one stream); false: nsplit parts (mpopis_set_overlap)
    mpopis::EnvDesc env{};
    std::string err;
    std::vector<void*> allocs;
    // resident env + policy state
    double *d_x = nullptr, *d_xext = nullptr, *d_U = nullptr, *d_Ucur = nullptr, *d_Uin = nullptr;
    int *d_t = nullptr, *d_done = nullptr;
    // proposal
    double *d_Sigma0 = nullptr, *d_L0 = nullptr, *d_dscale0 = nullptr;   // shared pol.Σ, its factor, sqrt(diag)
    double *d_Sig = nullptr, *d_L = nullptr, *d_tmpS = nullptr, *d_dscale = nullptr;
    double *d_L0p = nullptr, *d_Lp = nullptr;     // the factors once more in the fused sampler's staging layout (potrf_panel_doubles(cs) per matrix; cs <= 128)
    bool sigma_diag = false;
    // samples / costs / weights
    double *d_Z = nullptr, *d_E = nullptr, *d_Zin = nullptr, *d_cost = nullptr, *d_w = nullptr;
    unsigned long long* d_cmin = nullptr;   // [B] running minimum cost of the last rollout launch (cost_key), when the AIS reweighting is folded into the moments kernel
    bool weights_in_moments = false;        // :μΣaismppi on a car env with a shape that allows it (engine_api.hip, create)
    double* d_wsum = nullptr;          // [B] Σ_k w_k of the last k_weights launch of the AIS loop
    double *d_wn = nullptr, *d_mu = nullptr, *d_gvec = nullptr, *d_control = nullptr, *d_reward = nullptr, *d_traj = nullptr;
    int *d_status = nullptr, *d_active = nullptr, *d_iters = nullptr;
    unsigned long long* d_iters_acc = nullptr;   // per slot: AIS iterations executed by the policy steps BEFORE the last one (k_step_begin folds iters in before clearing it)
    uint64_t* d_seeds = nullptr;
    double* d_rng_tab = nullptr;            // Box-Muller tables (philox.h), filled at creation
    // elite selection / resampling
    int32_t *d_order = nullptr, *d_resi = nullptr, *d_alias = nullptr, *d_residx_log = nullptr, *d_resi_in = nullptr;
    double *d_resu = nullptr, *d_accept = nullptr, *d_resu_in = nullptr;
    // moments workspace
    double* d_part = nullptr; int ksplit = 1;
    // CMA (src/mppi_mpopi_policies.jl:513-525 constants; :536-545 per-call state)
    int m_elite = 0;
    double cma_consts[7] = {0, 0, 0, 0, 0, 0, 0};    // mu_eff, cσ, dσ, cΣ, c1, cμ, E
    std::vector<double> cma_ws_host;
    double *d_cma_scal = nullptr, *d_cma_vec = nullptr, *d_sig2 = nullptr, *d_cma_ws = nullptr;
    double* d_tri_dinv = nullptr;                                                          // [B][ceil(cs/16)][256] diagonal-block inverses of L (k_trtri_diag)
    double *d_lanV = nullptr, *d_Cdw = nullptr, *d_fro_part = nullptr, *d_fro = nullptr, *d_lan_prep = nullptr;   // Σ^-½ δw and tr(Σ^-1) (kernels_invsqrt.hip)
    unsigned long long* d_coop_flags = nullptr; unsigned long long coop_epoch = 0;         // cooperative Cholesky (cs > 128): panel flags [B][ceil(cs/16)], launch counter
    unsigned long long* d_lan_x = nullptr; int lan_regions = 1;                            // cooperative Lanczos: exchange granules, basis spill regions per slot
    int *d_potrf_redo = nullptr, *d_lan_redo = nullptr, *d_coop_timeouts = nullptr;         // cooperative kernels that gave up (CoopCtx, engine.h)
    int* h_coop_timeouts = nullptr; bool coop_disabled = false;                            // pinned mirror of the counter; set after the first time-out
    int coop_share = 1;                                                                    // multi-stream schedule: that many cluster launches may be in flight at once
    mpopis::CoopCtx potrf_coop() { mpopis::CoopCtx c; if (!coop_disabled) { c.flags = d_coop_flags; c.epoch = &coop_epoch; c.redo = d_potrf_redo; c.timeouts = d_coop_timeouts; c.share = coop_share; } return c; }
    mpopis::CoopCtx lan_coop() { mpopis::CoopCtx c; if (!coop_disabled) { c.flags = d_lan_x; c.epoch = &coop_epoch; c.redo = d_lan_redo; c.timeouts = d_coop_timeouts; c.share = coop_share; } return c; }
    int32_t* d_alias_stack = nullptr;                                                      // :pmcmppi with K beyond the LDS-resident alias construction: the two stacks (B x 2K ints)
    int* d_alias_need = nullptr;                                                           // :pmcmppi: slots whose alias table the parallel construction could not certify
    int* d_lan_m = nullptr;                                                                // Lanczos steps taken per slot (diagnostic)
    unsigned long long* d_tri_cnt = nullptr;                                               // [B][2]: arrival counter of a slot's trace workgroups (the last one prepares the Lanczos run) and their ||Σ||_inf; zero between launches
    double *d_qdist = nullptr, *d_qbeta = nullptr; int* d_qwithin = nullptr;
    // Level-3 harness
    double* d_hs = nullptr; int* d_alive = nullptr; const int* alive_gate = nullptr; bool status_sticky = false;
    double noise_sx = 0.0, noise_sy = 0.0, noise_spsi = 0.0;   // simulate_car_racing state noise (car_example.jl:224-236)
    // RCCL communicator for the summary gather (engine_comm.hip); world == 1 needs none
    void* comm = nullptr; int comm_rank = 0, comm_world = 1;
    // bookkeeping
    uint64_t mpc_step = 0;
    std::vector<int> h_status;
    double* h_pin = nullptr;          // pinned staging for the small per-step outputs (control, iters)
    // mpopis_policy_call mailbox: ONE pinned, device-mapped, coherent host block.  The first kernel of the call reads (x, U, t, done) straight
    // from it, the last kernel writes (control, rolled U, status, iters, cooperative time-outs) straight into it: no copy commands, one wait.
    // doubles: in_x[B*ss] in_U[B*cs] out_control[B*as] out_U[B*cs]; then ints: in_t[B] in_done[B] out_status[B] out_iters[B] out_coop[1]
    double* h_call = nullptr; double* d_call = nullptr;   // host pointer / the same block as the device sees it
    double wait_est_ms = 0.0;                             // how long the last synchronous per-step wait took (0: unknown): long steps block instead of spinning
    unsigned call_seq = 0;                                // sequence number the last k_call_out publishes (the host may spin on it; wraps)
    // timing
    bool timing = false, ev_open = false; int timing_mask = ~0;
    // MPOPIS_DEBUG_LAUNCH=1: after every kernel class of a policy step, hipGetLastError + stream sync, so that a failing
    // launch / faulting kernel is reported with the class it belongs to (release runs check once per call)
    bool debug_launch = false; int cur_class = 0; std::string launch_err;
    std::vector<hipEvent_t> events; std::vector<int> ev_slot; int ev_used = 0;

    void time_begin(int slot);
    void time_end();
    void prepare_state();
    void rollout(const double* Ucur, const double* Uorig, const double* gvec, const int* act, int* iters = nullptr, int iter_n = 0);
    int B_full = 0;                                       // cfg.batch (B is narrowed to a part's slots while a part-chain is enqueued)
    // What the scatter kernel's form goes by (launch_wcov_mfma): the whole batch -- or -1 = "the compact form" for shapes whose DEFAULT schedule is
    // part-chains: the row form's 8-wave workgroups take a CU's whole LDS (136 KB), which a lone stream does not mind (74 vs 86 us at 64 trials) and
    // concurrent chains do (no rollout / sampler workgroup fits beside one: the 64-trial step 5.64 instead of 5.35 ms).  A function of the shape
    // only, never of the schedule actually running, so that a slot's bits do not depend on mpopis_set_overlap.
    int wcov_sel_batch() const { return auto_parts() > 1 ? -1 : B_full; }
    // The part-chain streams must sit on DIFFERENT hardware queues: HIP deals its (by default four) queues to streams in creation order, whatever else
    // the process has created before, and two chains that share a queue serialise (the same 64-trial step 6.8 instead of 5.3 ms when torch had
    // initialised the device first).  Checked once, at the first multi-part step, with a 200-us spin kernel per stream (verify_part_streams); streams
    // that share a queue with an earlier one are replaced, and max_parts says how many independent ones there are.
    bool part_streams_checked = false;
    int max_parts = kMaxSplit;
    std::vector<hipStream_t> rejected_streams;            // (kept until the handle goes: destroying one would hand its queue to the next candidate)
    void verify_part_streams();
    int auto_parts() const;                               // part-chains of the default schedule for this handle's shape
    int policy_step_enqueue(bool injected);
    int step_enqueue_view(bool injected, hipEvent_t wait_first, hipEvent_t record_after_first_sampler);
    void shift_slots(ptrdiff_t db);                           // move every per-slot device pointer by db slots (slot views)
    int ais_update(int n, bool injected);
    int run_trials(int num_steps, int laps, double* records, double* actions);
    void init_cma_constants();
    void cma_begin();
    const double* cma_sigma2();
};

namespace mpopis {
// Error precedence when several slots (or several kernels of one slot) failed in one call: HIP (-4) > ACTION (-3) > NOT_PD (-2) > NUMERIC (-5),
// i.e. the numeric minimum among the codes of ABI version 1, which the newer MPOPIS_ERR_NUMERIC never hides (include/mpopis.h).
// (status_rank lives in engine.h: the device-side writers merge by the same rule, status_raise)
inline int worse_status(int a, int b) { return status_rank(b) > status_rank(a) ? b : a; }
void launch_scale_rows(double* Z, const double* dsc, int B, int cs, int K, hipStream_t s);
inline void launch_mppi_Z_in(const double* src, double* dst, int B, int T, int K, int as, hipStream_t s) { launch_transpose_in(src, dst, B * T, as, K, s); }
inline void launch_mppi_E_out(const double* src, double* dst, int B, int T, int K, int as, hipStream_t s) { launch_transpose_out(src, nullptr, nullptr, dst, B * T, as, K, s); }
}
