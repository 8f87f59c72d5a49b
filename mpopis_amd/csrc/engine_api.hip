// engine_api.hip -- handle, device buffers, launch sequences and the C ABI of include/mpopis.h.
//
// One handle = B resident independent trials (env + policy pairs).  Every policy step is a fixed
// sequence of kernel launches on the handle's stream; data-dependent control flow of the
// reference (CE/CMA early break :459-461,:567-569, PosDefException) is turned into per-slot
// `active` predicates evaluated on the device, so a step never needs a host round trip.
#include "engine.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

using namespace mpopis;

static std::string g_create_error;

#define HIPCHK(h, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            char buf_[512];                                                                     \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            if (h) (h)->err = buf_; else g_create_error = buf_;                                 \
            return MPOPIS_ERR_HIP;                                                              \
        }                                                                                       \
    } while (0)

#include "engine_handle.h"
#include "philox.h"
#include <chrono>
#include <atomic>

// ---- tiny utility kernels -----------------------------------------------------------------------
__global__ void k_fill_i32(int* p, int v, size_t n) { size_t i = blockIdx.x * (size_t)256 + threadIdx.x; if (i < n) p[i] = v; }
__global__ void k_copy_f64(const double* s, double* d, size_t n) { size_t i = blockIdx.x * (size_t)256 + threadIdx.x; if (i < n) d[i] = s[i]; }
__global__ void k_bcast_f64(const double* s, double* d, size_t n, int B) {     // d[b][i] = s[i]
    size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i < n) for (int b = 0; b < B; ++b) d[(size_t)b * n + i] = s[i];
}

__global__ void k_scaled_copy_f64(const double* s, size_t sstride, const double* scale, double* d, size_t n) {   // d[b][i] = scale[b]*s[b][i]
    const int b = blockIdx.y; const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i < n) d[(size_t)b * n + i] = (scale ? scale[b] : 1.0) * s[(size_t)b * sstride + i];
}

// mpopis_policy_call: the hand-over kernels between the pinned mailbox (device-mapped host memory) and the resident buffers.
struct CallBox { double *x, *U, *control, *Uout; int *t, *done, *status, *iters, *coop; volatile int* seq; };
static CallBox call_box(double* base, int B, int ss, int cs, int as) {
    CallBox m;
    m.x = base; m.U = m.x + (size_t)B * ss; m.control = m.U + (size_t)B * cs; m.Uout = m.control + (size_t)B * as;
    m.t = (int*)(m.Uout + (size_t)B * cs); m.done = m.t + B; m.status = m.done + B; m.iters = m.status + B; m.coop = m.iters + B; m.seq = m.coop + 1;
    return m;
}
static size_t call_box_bytes(int B, int ss, int cs, int as) { return sizeof(double) * ((size_t)B * (ss + 2 * cs + as)) + sizeof(int) * ((size_t)4 * B + 4); }
__global__ void k_call_in(CallBox m, double* x, double* U, int* t, int* done, int nx, int nu, int B, int flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if ((flags & 1) && i < nx) x[i] = m.x[i];
    if ((flags & 2) && i < nu) U[i] = m.U[i];
    if (i < B) { if (flags & 4) t[i] = m.t[i]; if (flags & 8) done[i] = m.done[i]; }
}
// ONE workgroup: every thread stores its share and fences at system scope, the workgroup meets, then thread 0 publishes the call's sequence
// number -- the host may spin on that word instead of going through the runtime's completion signal.
__global__ void k_call_out(CallBox m, const double* control, const double* U, const int* status, const int* iters, const int* coop, int nc, int nu, int B, int seq) {
    for (int i = threadIdx.x; i < nc; i += 256) m.control[i] = control[i];
    for (int i = threadIdx.x; i < nu; i += 256) m.Uout[i] = U[i];
    for (int i = threadIdx.x; i < B; i += 256) { m.status[i] = status[i]; m.iters[i] = iters[i]; }
    if (threadIdx.x == 0) *m.coop = coop ? *coop : 0;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { *m.seq = seq; __threadfence_system(); }
}

static void fill_i32(int* p, int v, size_t n, hipStream_t s) { hipLaunchKernelGGL(k_fill_i32, dim3((n + 255) / 256), dim3(256), 0, s, p, v, n); }
static void copy_f64(const double* s_, double* d, size_t n, hipStream_t s) { hipLaunchKernelGGL(k_copy_f64, dim3((n + 255) / 256), dim3(256), 0, s, s_, d, n); }

template <class T> static int dalloc(mpopis_handle* h, T** p, size_t n) {
    HIPCHK(h, hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
    HIPCHK(h, hipMemsetAsync(*p, 0, std::max<size_t>(n, 1) * sizeof(T), h->stream));
    h->allocs.push_back((void*)*p);
    return 0;
}

static const char* kClassNames[] = {"rollout", "sample", "potrf", "reweight", "moments", "select", "other"};

void mpopis_handle::time_begin(int slot) {
    cur_class = slot;
    ev_open = timing && (timing_mask & (1 << slot)) && ev_used + 2 <= (int)events.size();
    if (!ev_open) return;
    (void)hipEventRecord(events[ev_used], stream);
    ev_slot.push_back(slot);
}
void mpopis_handle::time_end() {
    if (debug_launch && launch_err.empty()) {
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) launch_err = std::string("kernel class '") + kClassNames[cur_class] + "': " + hipGetErrorString(e);
    }
    if (!ev_open) return;
    (void)hipEventRecord(events[ev_used + 1], stream);
    ev_used += 2;
    ev_open = false;
}

// Host wait for the handle's stream.  A synchronous pol(env) of one trial is 0.2-2 ms of GPU work; hipStreamSynchronize's blocking wake-up
// adds 20-50 us to that, so poll the stream for the first 3 ms (one host thread spinning, as a CPU implementation of the call would) and
// only then block.
static hipError_t wait_stream(hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(3)) break;
    }
    return hipStreamSynchronize(s);
}
// The per-MPC-step waits (mpopis_policy_step / mpopis_policy_call) know how long the previous step took: the spin only pays for sub-millisecond
// steps (C2 0.15 ms, where the blocking wake-up is a quarter of the call); a handle whose last wait exceeded kSpinWorthMs (C3 1.8 ms, C4 5-25 ms, the
// 64-trial batches) blocks right away instead of burning a host core for 3 ms first -- the 20-50 us wake-up is then below 3 % of the step.
constexpr double kSpinWorthMs = 1.0;
static bool wait_should_spin(const mpopis_handle* h) { return h->wait_est_ms <= kSpinWorthMs; }
static void wait_note(mpopis_handle* h, std::chrono::steady_clock::time_point t0) {
    h->wait_est_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
static hipError_t wait_step(mpopis_handle* h) {
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = wait_should_spin(h) ? wait_stream(h->stream) : hipStreamSynchronize(h->stream);
    wait_note(h, t0);
    return e;
}

static int fold_status(mpopis_handle* h, const int* per_slot) {
    int st = 0;
    for (int b = 0; b < h->B; ++b) st = mpopis::worse_status(st, per_slot[b]);
    if (st == MPOPIS_ERR_NOT_PD) h->err = "PosDefException: proposal covariance is not positive definite";
    else if (st == MPOPIS_ERR_ACTION) h->err = "Action is not in action space (non-finite control/cost)";
    else if (st == MPOPIS_ERR_NUMERIC) h->err = "cmamppi: Σ^-0.5 δw could not be formed (non-finite covariance, trace or δw)";
    return st;
}

static int sync_status(mpopis_handle* h) {
    HIPCHK(h, hipMemcpyAsync(h->h_status.data(), h->d_status, sizeof(int) * h->B, hipMemcpyDeviceToHost, h->stream));
    if (h->h_coop_timeouts && !h->coop_disabled) HIPCHK(h, hipMemcpyAsync(h->h_coop_timeouts, h->d_coop_timeouts, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, wait_step(h));
    // a cluster gave up waiting for a partner (its slot was recomputed by the one-workgroup kernel in the same step, so the results are
    // complete): this device cannot keep the clusters co-resident right now -- stop using them for this handle instead of paying the wait again
    if (h->h_coop_timeouts && *h->h_coop_timeouts > 0) h->coop_disabled = true;
    return fold_status(h, h->h_status.data());
}

// ---- ABI ---------------------------------------------------------------------------------------
extern "C" {

int mpopis_abi_version(void) { return MPOPIS_ABI_VERSION; }

const char* mpopis_last_error(const mpopis_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static void default_car_params(double* p) {      // src/envs/car_racing.jl:68-93
    const double d2r = M_PI / 180.0;
    const double v[20] = {2000.0, 3764.0, 0.3, 1.53, 1.23, 241.0, 25.1, 150000.0, 280000.0, 0.9, 0.9, 18.0 * d2r, 90.0 * d2r,
                          7200.0, 22500.0, 0.6, 0.0, 45.0 * d2r, 0.1, 0.01};
    memcpy(p, v, sizeof v);
}
static void default_mc_params(double* p) {       // RL.jl MountainCarEnvParams, continuous=true
    const double v[8] = {-1.2, 0.6, 0.07, 0.45, 0.0, 0.0015, 0.0025, 200.0};
    memcpy(p, v, sizeof v);
}

static void default_cp_params(double* p) {       // RL.jl CartPoleEnvParams
    const double v[11] = {9.8, 1.0, 0.1, 1.1, 0.5, 0.05, 10.0, 0.02, 12.0 * 2.0 * M_PI / 360.0, 2.4, 200.0};
    memcpy(p, v, sizeof v);
}

int mpopis_create(const mpopis_config* cfg, mpopis_handle** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return MPOPIS_ERR_ARG; }
    *out = nullptr;
    const bool car = cfg->env_kind == MPOPIS_ENV_CAR;
    if (!(car || cfg->env_kind == MPOPIS_ENV_MOUNTAINCAR || cfg->env_kind == MPOPIS_ENV_CARTPOLE)) { g_create_error = "unknown env kind"; return MPOPIS_ERR_ARG; }
    if (car && (cfg->num_cars < 1 || cfg->num_cars > kMaxCars)) { g_create_error = "num_cars must be 1..4"; return MPOPIS_ERR_ARG; }
    if (cfg->policy < MPOPIS_POL_MPPI || cfg->policy > MPOPIS_POL_PMCMPPI) { g_create_error = "No policy_type of that kind"; return MPOPIS_ERR_ARG; }
    if (cfg->num_samples < 1 || cfg->horizon < 1 || cfg->batch < 1) { g_create_error = "num_samples, horizon, batch must be >= 1"; return MPOPIS_ERR_ARG; }
    if (cfg->policy == MPOPIS_POL_CMAMPPI && (car ? 2 * cfg->num_cars : 1) * cfg->horizon > invsqrt_max_n()) { g_create_error = "cmamppi: control space too large for the on-chip Σ^-0.5 δw kernel"; return MPOPIS_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_error = "no HIP device available (the engine has no CPU fallback)"; return MPOPIS_ERR_HIP; }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "bad device ordinal"; return MPOPIS_ERR_ARG; }
    mpopis_handle* h = new mpopis_handle();
    h->cfg = *cfg;
    // a failure half-way (hipStreamCreate under many handles is a realistic one) must not leak the handle and what it already owns
#define CREATECHK(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            char buf_[512];                                                                     \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            g_create_error = buf_;                                                              \
            mpopis_destroy(h);                                                                  \
            return MPOPIS_ERR_HIP;                                                              \
        }                                                                                       \
    } while (0)
    CREATECHK(hipSetDevice(cfg->device));
    CREATECHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATECHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < mpopis_handle::kMaxSplit - 1; ++i) {
        CREATECHK(hipStreamCreateWithFlags(&h->xstream[i], hipStreamNonBlocking));
        CREATECHK(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
        CREATECHK(hipEventCreateWithFlags(&h->ev_skew[i], hipEventDisableTiming));
    }
#undef CREATECHK
    if (const char* e = getenv("MPOPIS_DEBUG_LAUNCH")) h->debug_launch = atoi(e) != 0;
    if (const char* e = getenv("MPOPIS_NSPLIT")) { h->nsplit = std::max(1, std::min((int)mpopis_handle::kMaxSplit, atoi(e))); h->split_auto = false; h->split_pinned = true; }   // experiments / profiling: pins the schedule, mpopis_set_overlap is then ignored
    h->B = cfg->batch; h->B_full = cfg->batch; h->K = cfg->num_samples; h->T = cfg->horizon;
    h->as = car ? 2 * cfg->num_cars : 1;
    h->ss = car ? 8 * cfg->num_cars : (cfg->env_kind == MPOPIS_ENV_CARTPOLE ? 4 : 2);
    h->cs = h->as * h->T;
    h->N = (cfg->policy == MPOPIS_POL_MPPI || cfg->policy == MPOPIS_POL_GMPPI) ? 1 : std::max(1, cfg->ais_its);
    h->gamma = cfg->lambda * (1 - cfg->alpha);
    h->env.kind = cfg->env_kind; h->env.ncars = car ? cfg->num_cars : 0; h->env.ss = h->ss; h->env.as = h->as;
    double p[20];
    default_car_params(p); h->env.car = make_car_params(p);
    default_mc_params(p); h->env.mc = make_mc_params(p);
    default_cp_params(p); h->env.cp = make_cp_params(p);
    for (int i = 0; i < kMaxAs; ++i) { h->env.lo[i] = -1.0; h->env.hi[i] = 1.0; }
    h->env.track = Track{nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr};
    const int B = h->B, K = h->K, cs = h->cs;
    const size_t nn = (size_t)cs * cs;
    int rc = 0;
    rc |= dalloc(h, &h->d_x, (size_t)B * h->ss); rc |= dalloc(h, &h->d_xext, (size_t)B * kMaxCars * kCarExt); rc |= dalloc(h, &h->d_t, B); rc |= dalloc(h, &h->d_done, B);
    rc |= dalloc(h, &h->d_U, (size_t)B * cs); rc |= dalloc(h, &h->d_Ucur, (size_t)B * cs); rc |= dalloc(h, &h->d_Uin, (size_t)B * cs);
    rc |= dalloc(h, &h->d_Sigma0, nn); rc |= dalloc(h, &h->d_Sig, (size_t)B * nn + kInvsqrtPadDoubles); rc |= dalloc(h, &h->d_L, (size_t)B * nn);
    rc |= dalloc(h, &h->d_L0, nn); rc |= dalloc(h, &h->d_tmpS, (size_t)B * nn);
    if (sample_trmm_fusable(cs)) { rc |= dalloc(h, &h->d_L0p, potrf_panel_doubles(cs)); rc |= dalloc(h, &h->d_Lp, (size_t)B * potrf_panel_doubles(cs)); }
    rc |= dalloc(h, &h->d_coop_flags, potrf_coop_flag_words(B, cs)); rc |= dalloc(h, &h->d_potrf_redo, B); rc |= dalloc(h, &h->d_lan_redo, B); rc |= dalloc(h, &h->d_coop_timeouts, 1);
    rc |= dalloc(h, &h->d_Z, (size_t)B * cs * K); rc |= dalloc(h, &h->d_E, (size_t)B * cs * K);
    rc |= dalloc(h, &h->d_cost, (size_t)B * K); rc |= dalloc(h, &h->d_w, (size_t)B * K);
    rc |= dalloc(h, &h->d_wn, (size_t)B * cs); rc |= dalloc(h, &h->d_mu, (size_t)B * cs); rc |= dalloc(h, &h->d_gvec, (size_t)B * cs);
    rc |= dalloc(h, &h->d_dscale, (size_t)B * cs); rc |= dalloc(h, &h->d_dscale0, (size_t)cs);
    rc |= dalloc(h, &h->d_control, (size_t)B * h->as); rc |= dalloc(h, &h->d_reward, B); rc |= dalloc(h, &h->d_wsum, B); rc |= dalloc(h, &h->d_cmin, B);
    rc |= dalloc(h, &h->d_status, B); rc |= dalloc(h, &h->d_active, B); rc |= dalloc(h, &h->d_iters, B); rc |= dalloc(h, &h->d_iters_acc, B);
    rc |= dalloc(h, &h->d_seeds, B); rc |= dalloc(h, &h->d_rng_tab, mpopis::kRngTabDoubles);      // philox.h
    rc |= dalloc(h, &h->d_order, (size_t)B * K); rc |= dalloc(h, &h->d_resi, (size_t)B * K); rc |= dalloc(h, &h->d_resu, (size_t)B * K);
    rc |= dalloc(h, &h->d_accept, (size_t)B * K); rc |= dalloc(h, &h->d_alias, (size_t)B * K); rc |= dalloc(h, &h->d_alias_need, B);
    if (cfg->policy == MPOPIS_POL_PMCMPPI && K > alias_lds_max_K()) rc |= dalloc(h, &h->d_alias_stack, (size_t)B * 2 * K);      // the global-workspace alias construction
    rc |= dalloc(h, &h->d_residx_log, (size_t)B * std::max(1, h->N - 1) * K);
    h->ksplit = std::max(1, std::min(std::min(32, K / 128), std::max(1, 512 / B)));   // ~2-4 workgroups per CU in the scatter kernel
    if (const char* e = getenv("MPOPIS_KSPLIT")) h->ksplit = std::max(1, std::min(32, atoi(e)));
    rc |= dalloc(h, &h->d_part, wcov_mfma_workspace_doubles(B, cs, h->ksplit));
    {
        static const int env_fold = [] { const char* e = getenv("MPOPIS_FOLD_WEIGHTS"); return e ? atoi(e) : 1; }();      // 0: keep the separate reweighting launch (A/B)
        h->weights_in_moments = env_fold && cfg->policy == MPOPIS_POL_MUSIGMAAISMPPI && cfg->env_kind == MPOPIS_ENV_CAR && wcov_weights_from_cost_ok(cs, K, h->ksplit);
    }
    if (cfg->policy == MPOPIS_POL_CMAMPPI) {
        rc |= dalloc(h, &h->d_cma_scal, (size_t)B * 8); rc |= dalloc(h, &h->d_cma_vec, (size_t)B * 3 * cs); rc |= dalloc(h, &h->d_sig2, B);
        rc |= dalloc(h, &h->d_cma_ws, (size_t)K);
        h->lan_regions = invsqrt_coop_groups(B, cs);
        rc |= dalloc(h, &h->d_lanV, invsqrt_workspace_doubles(B, cs, h->lan_regions)); rc |= dalloc(h, &h->d_lan_x, invsqrt_coop_words(B, cs)); rc |= dalloc(h, &h->d_Cdw, (size_t)B * cs);
        rc |= dalloc(h, &h->d_fro_part, (size_t)B * ((cs + 15) / 16)); rc |= dalloc(h, &h->d_tri_dinv, trtri_dinv_doubles(B, cs)); rc |= dalloc(h, &h->d_fro, B); rc |= dalloc(h, &h->d_lan_m, B); rc |= dalloc(h, &h->d_lan_prep, lanczos_prep_doubles(B)); rc |= dalloc(h, &h->d_tri_cnt, 2 * (size_t)B);
    }
    if (cfg->log_trajectories) rc |= dalloc(h, &h->d_traj, (size_t)B * K * h->T * h->ss);
    if (rc) { g_create_error = h->err; mpopis_destroy(h); return MPOPIS_ERR_HIP; }
    launch_rng_tab_init(h->d_rng_tab, h->stream);
    h->h_status.assign(B, 0);
    if (hipHostMalloc((void**)&h->h_pin, sizeof(double) * B * (h->as + 2)) != hipSuccess) h->h_pin = nullptr;
    if (hipHostMalloc((void**)&h->h_call, call_box_bytes(B, h->ss, cs, h->as), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->d_call, h->h_call, 0) != hipSuccess) { if (h->h_call) (void)hipHostFree(h->h_call); h->h_call = nullptr; h->d_call = nullptr; }
    if (h->h_call) memset(h->h_call, 0, call_box_bytes(B, h->ss, cs, h->as));
    if (hipHostMalloc((void**)&h->h_coop_timeouts, sizeof(int)) != hipSuccess) h->h_coop_timeouts = nullptr; else *h->h_coop_timeouts = 0;
    if (const char* e = getenv("MPOPIS_NO_COOP")) h->coop_disabled = atoi(e) != 0;
    // Σ default = I (cov_mat default [1.0], src/mppi_mpopi_policies.jl:42) ; seeds
    const int n0 = (cfg->policy == MPOPIS_POL_MPPI) ? h->as : cs;
    std::vector<double> eye((size_t)n0 * n0, 0.0);
    for (int i = 0; i < n0; ++i) eye[(size_t)i * n0 + i] = 1.0;
    if (mpopis_set_Sigma(h, eye.data(), n0) != 0) { g_create_error = h->err; mpopis_destroy(h); return MPOPIS_ERR_HIP; }
    if (mpopis_seed(h, cfg->seed) != 0) { g_create_error = h->err; mpopis_destroy(h); return MPOPIS_ERR_HIP; }
    if (mpopis_reset(h) != 0) { g_create_error = h->err; mpopis_destroy(h); return MPOPIS_ERR_HIP; }
    if (cfg->policy == MPOPIS_POL_CMAMPPI) {
        h->init_cma_constants();
        if ((long long)cs * h->m_elite < K) { g_create_error = "BoundsError: δs[order[ii]] needs cs*m_elite >= K (src/mppi_mpopi_policies.jl:593)"; mpopis_destroy(h); return MPOPIS_ERR_ARG; }
        if (hipMemcpy(h->d_cma_ws, h->cma_ws_host.data(), sizeof(double) * K, hipMemcpyHostToDevice) != hipSuccess) { g_create_error = "upload failed"; mpopis_destroy(h); return MPOPIS_ERR_HIP; }
    }
    if (cfg->policy == MPOPIS_POL_CEMPPI) h->m_elite = (int)nearbyint(K * (1 - cfg->elite_threshold));   // :437
    // (no upper limit on K: beyond K = 8192 the sort is the chunked chip-wide rank sort, beyond 7168 the alias table is built on global arrays --
    // the reference's sortperm / Categorical take any K, :455, :804)
    *out = h;
    return MPOPIS_OK;
}

void mpopis_destroy(mpopis_handle* h) {
    if (!h) return;
    if (h->comm) (void)mpopis_comm_destroy(h);
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto st : h->xstream) if (st) (void)hipStreamSynchronize(st);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->h_pin) (void)hipHostFree(h->h_pin);
    if (h->h_call) (void)hipHostFree(h->h_call);
    if (h->h_coop_timeouts) (void)hipHostFree(h->h_coop_timeouts);
    for (auto e : h->events) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    for (auto st : h->xstream) if (st) (void)hipStreamDestroy(st);
    for (auto st : h->rejected_streams) if (st) (void)hipStreamDestroy(st);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (auto e : h->ev_join) if (e) (void)hipEventDestroy(e);
    for (auto e : h->ev_skew) if (e) (void)hipEventDestroy(e);
    delete h;
}

int mpopis_set_env_params(mpopis_handle* h, const double* p, int32_t n) {
    if (!h || !p) return MPOPIS_ERR_ARG;
    if (h->env.kind == MPOPIS_ENV_CAR) {
        if (n != MPOPIS_CAR_NPARAMS) { h->err = "car env expects 20 parameters"; return MPOPIS_ERR_ARG; }
        h->env.car = make_car_params(p);
    } else if (h->env.kind == MPOPIS_ENV_CARTPOLE) {
        if (n != MPOPIS_CARTPOLE_NPARAMS) { h->err = "CartPole expects 11 parameters"; return MPOPIS_ERR_ARG; }
        h->env.cp = make_cp_params(p);
    } else {
        if (n != MPOPIS_MOUNTAINCAR_NPARAMS) { h->err = "MountainCar expects 8 parameters"; return MPOPIS_ERR_ARG; }
        h->env.mc = make_mc_params(p);
    }
    return MPOPIS_OK;
}

int mpopis_set_track(mpopis_handle* h, const double* x, const double* y, const double* w, int32_t P) {
    if (!h || !x || !y || !w || P < 2 || P > mpopis::kMaxTrackPoints) { if (h) h->err = "bad track (need 2 <= P <= 2048 points)"; return MPOPIS_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    double* d = nullptr;
    if (dalloc(h, &d, (size_t)4 * P)) return MPOPIS_ERR_HIP;
    std::vector<double> n2(P);
    for (int i = 0; i < P; ++i) n2[i] = x[i] * x[i] + y[i] * y[i];
    HIPCHK(h, hipMemcpyAsync(d + 3 * P, n2.data(), sizeof(double) * P, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d, x, sizeof(double) * P, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d + P, y, sizeof(double) * P, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d + 2 * P, w, sizeof(double) * P, hipMemcpyHostToDevice, h->stream));
    // neighbour tables of the anchored nearest-point search (car_dynamics.h: build_track_tables)
    const int W = std::min<int>(kTrackNbrW, P);
    std::vector<double> nd; std::vector<int> ni;
    build_track_tables(P, x, y, nd, ni);
    double* dnd = nullptr; int* dni = nullptr;
    if (dalloc(h, &dnd, nd.size()) || dalloc(h, &dni, ni.size())) return MPOPIS_ERR_HIP;
    HIPCHK(h, hipMemcpyAsync(dnd, nd.data(), sizeof(double) * nd.size(), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(dni, ni.data(), sizeof(int) * ni.size(), hipMemcpyHostToDevice, h->stream));
    std::vector<double> ring, cert;
    build_track_ring(P, x, y, w, n2.data(), nd, ring, cert);
    double* dring = nullptr;
    if (dalloc(h, &dring, ring.size() + cert.size())) return MPOPIS_ERR_HIP;
    HIPCHK(h, hipMemcpyAsync(dring, ring.data(), sizeof(double) * ring.size(), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(dring + ring.size(), cert.data(), sizeof(double) * cert.size(), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    h->env.track = Track{d, d + P, d + 2 * P, d + 3 * P, P, dni, dnd, W, dring, dring + ring.size()};
    return MPOPIS_OK;
}

int mpopis_set_action_bounds(mpopis_handle* h, const double* lo, const double* hi) {
    if (!h || !lo || !hi) return MPOPIS_ERR_ARG;
    for (int i = 0; i < h->as; ++i) { h->env.lo[i] = lo[i]; h->env.hi[i] = hi[i]; }
    return MPOPIS_OK;
}

int mpopis_reset(mpopis_handle* h) {
    if (!h) return MPOPIS_ERR_ARG;
    std::vector<double> x((size_t)h->B * h->ss, 0.0);
    for (int b = 0; b < h->B; ++b) {
        double* s = x.data() + (size_t)b * h->ss;
        if (h->env.kind == MPOPIS_ENV_CAR) {
            for (int c = 0; c < h->env.ncars; ++c) {            // car_racing.jl:215-223; multi-car_racing.jl:160-180
                const int ii = c + 1;
                if (ii >= 2) s[8 * c] = (ii % 2 == 0) ? (ii / 2.0 * 5.0) : ((1 - ii) / 2.0 * 5.0);
                s[8 * c + 2] = 90.0 * (M_PI / 180.0);
                s[8 * c + 3] = 10.0;
            }
        } else if (h->env.kind == MPOPIS_ENV_MOUNTAINCAR) { s[0] = -0.5; s[1] = 0.0; }   // reference: x ~ U(-0.6,-0.4) from an unseeded RNG
        // CartPole: reference draws 0.1*rand(4) - 0.05; deterministic centre = all zeros
    }
    std::vector<int> z(h->B, 0);
    return mpopis_set_state(h, x.data(), z.data(), z.data());
}

int mpopis_set_state(mpopis_handle* h, const double* x, const int32_t* t, const int32_t* done) {
    if (!h || !x) return MPOPIS_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(h->d_x, x, sizeof(double) * h->B * h->ss, hipMemcpyHostToDevice, h->stream));
    if (t) HIPCHK(h, hipMemcpyAsync(h->d_t, t, sizeof(int) * h->B, hipMemcpyHostToDevice, h->stream));
    if (done) HIPCHK(h, hipMemcpyAsync(h->d_done, done, sizeof(int) * h->B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    return MPOPIS_OK;
}

int mpopis_get_state(mpopis_handle* h, double* x, int32_t* t, int32_t* done) {
    if (!h) return MPOPIS_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (x) HIPCHK(h, hipMemcpyAsync(x, h->d_x, sizeof(double) * h->B * h->ss, hipMemcpyDeviceToHost, h->stream));
    if (t) HIPCHK(h, hipMemcpyAsync(t, h->d_t, sizeof(int) * h->B, hipMemcpyDeviceToHost, h->stream));
    if (done) HIPCHK(h, hipMemcpyAsync(done, h->d_done, sizeof(int) * h->B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    return MPOPIS_OK;
}

int mpopis_set_U(mpopis_handle* h, const double* U) {
    if (!h || !U) return MPOPIS_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(h->d_U, U, sizeof(double) * h->B * h->cs, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    return MPOPIS_OK;
}
int mpopis_get_U(mpopis_handle* h, double* U) {
    if (!h || !U) return MPOPIS_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(U, h->d_U, sizeof(double) * h->B * h->cs, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    return MPOPIS_OK;
}

// MPPI_Policy_Params Σ handling :66-81 : an as x as cov_mat is block-replicated (block_diagm, utils.jl:9-21);
// for :mppi the as x as Σ is kept by the reference, which is the same distribution as the
// block-diagonal cs x cs one used here (Cholesky of a block-diagonal matrix is block-diagonal).
int mpopis_set_Sigma(mpopis_handle* h, const double* Sigma, int32_t n) {
    if (!h || !Sigma) return MPOPIS_ERR_ARG;
    const int cs = h->cs, as = h->as;
    std::vector<double> full((size_t)cs * cs, 0.0);
    if (n == cs && h->cfg.policy != MPOPIS_POL_MPPI) {
        memcpy(full.data(), Sigma, sizeof(double) * cs * cs);
    } else if (n == as) {
        for (int t = 0; t < h->T; ++t)
            for (int j = 0; j < as; ++j)
                for (int i = 0; i < as; ++i) full[(size_t)(t * as + i) + (size_t)(t * as + j) * cs] = Sigma[i + (size_t)j * as];
    } else { h->err = "Covariance matrix size problem"; return MPOPIS_ERR_ARG; }     // :79
    bool diag = true;
    for (int j = 0; j < cs && diag; ++j) for (int i = 0; i < cs; ++i) if (i != j && full[(size_t)i + (size_t)j * cs] != 0.0) { diag = false; break; }
    std::vector<double> ds(cs, 0.0);
    if (diag) for (int i = 0; i < cs; ++i) { const double v = full[(size_t)i + (size_t)i * cs]; if (!(v > 0.0)) { h->err = "PosDefException: Sigma"; return MPOPIS_ERR_NOT_PD; } ds[i] = sqrt(v); }
    h->sigma_diag = diag;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(h->d_Sigma0, full.data(), sizeof(double) * cs * cs, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_dscale0, ds.data(), sizeof(double) * cs, hipMemcpyHostToDevice, h->stream));
    // per-slot copy for the diagonal-Σ sampler, once per pol.Σ (it used to be re-broadcast by a launch of its own in every AIS iteration)
    hipLaunchKernelGGL(k_bcast_f64, dim3((cs + 255) / 256), dim3(256), 0, h->stream, h->d_dscale0, h->d_dscale, (size_t)cs, h->B);
    // factor once: L0 (shared by all slots; the reference refactors the same Σ every call, :307)
    fill_i32(h->d_status, 0, h->B, h->stream);
    launch_potrf(h->d_Sigma0, 0, h->d_L0, 1, cs, nullptr, h->d_status, nullptr, h->stream, h->potrf_coop(), h->d_L0p, 0);
    HIPCHK(h, hipMemcpyAsync(h->h_status.data(), h->d_status, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    if (h->h_status[0] != 0) { h->err = "PosDefException: Sigma is not positive definite"; return MPOPIS_ERR_NOT_PD; }
    return MPOPIS_OK;
}

int mpopis_seed(mpopis_handle* h, uint64_t seed) {
    if (!h) return MPOPIS_ERR_ARG;
    std::vector<uint64_t> s(h->B);
    for (int b = 0; b < h->B; ++b) s[b] = seed + (uint64_t)b + 1;              // seed!(pol, seed + k), car_example.jl:188
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(h->d_seeds, s.data(), sizeof(uint64_t) * h->B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    h->mpc_step = 0;
    return MPOPIS_OK;
}

int mpopis_seed_slots(mpopis_handle* h, const uint64_t* seeds) {
    if (!h || !seeds) return MPOPIS_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(h->d_seeds, seeds, sizeof(uint64_t) * h->B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    h->mpc_step = 0;
    return MPOPIS_OK;
}

int mpopis_get_Sigma(mpopis_handle* h, double* out) {
    if (!h || !out) return MPOPIS_ERR_ARG;
    if (h->cfg.policy == MPOPIS_POL_MPPI) { h->err = "mpopis_get_Sigma: :mppi keeps the as x as pol.Σ (no adaptation)"; return MPOPIS_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int pol = h->cfg.policy;
    const size_t nn = (size_t)h->cs * h->cs;
    const bool sigma_fixed = (pol == MPOPIS_POL_GMPPI || pol == MPOPIS_POL_IMPPI || pol == MPOPIS_POL_MUAISMPPI);
    // d_tmpS is scratch outside a policy step
    const double* scale = (pol == MPOPIS_POL_CMAMPPI && h->N > 1) ? h->d_sig2 : nullptr;     // MvNormal(σ²Σ′) :550-554
    hipLaunchKernelGGL(k_scaled_copy_f64, dim3((nn + 255) / 256, h->B), dim3(256), 0, h->stream,
                       sigma_fixed ? h->d_Sigma0 : h->d_Sig, sigma_fixed ? (size_t)0 : nn, scale, h->d_tmpS, nn);
    HIPCHK(h, hipMemcpyAsync(out, h->d_tmpS, sizeof(double) * h->B * nn, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    return MPOPIS_OK;
}

int mpopis_rollout_costs(mpopis_handle* h, const double* x0, const double* U, const double* U_orig, const double* E,
                         const double* Sigma_inv, double* cost) {
    if (!h || !U || !E || !cost) return MPOPIS_ERR_ARG;
    if (h->env.kind == MPOPIS_ENV_CAR && h->env.track.P == 0) { h->err = "track not set"; return MPOPIS_ERR_ARG; }
    if (h->gamma != 0.0 && !Sigma_inv) { h->err = "Sigma_inv required when alpha != 1"; return MPOPIS_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int B = h->B, K = h->K, cs = h->cs;
    if (x0) HIPCHK(h, hipMemcpyAsync(h->d_x, x0, sizeof(double) * B * h->ss, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_Ucur, U, sizeof(double) * B * cs, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_Uin, U_orig ? U_orig : U, sizeof(double) * B * cs, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_Z, E, sizeof(double) * B * cs * K, hipMemcpyHostToDevice, h->stream));
    launch_transpose_in(h->d_Z, h->d_E, B, cs, K, h->stream);
    const double* gv = nullptr;
    if (h->gamma != 0.0) {
        HIPCHK(h, hipMemcpyAsync(h->d_tmpS, Sigma_inv, sizeof(double) * cs * cs, hipMemcpyHostToDevice, h->stream));
        launch_gvec_from_inv(h->d_tmpS, h->d_Uin, h->gamma, h->d_gvec, B, cs, h->stream);
        gv = h->d_gvec;
    }
    fill_i32(h->d_status, 0, B, h->stream);
    h->prepare_state();
    h->rollout(h->d_Ucur, h->d_Uin, gv, nullptr);
    HIPCHK(h, hipMemcpyAsync(cost, h->d_cost, sizeof(double) * B * K, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    HIPCHK(h, hipGetLastError());
    return MPOPIS_OK;
}

int mpopis_env_step(mpopis_handle* h, const double* action, double* reward) {
    if (!h || !action) return MPOPIS_ERR_ARG;
    if (h->env.kind == MPOPIS_ENV_CAR && h->env.track.P == 0) { h->err = "track not set"; return MPOPIS_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(h->d_control, action, sizeof(double) * h->B * h->as, hipMemcpyHostToDevice, h->stream));
    fill_i32(h->d_status, 0, h->B, h->stream);
    launch_env_step(h->env, h->d_x, h->d_t, h->d_done, h->d_control, h->d_reward, h->d_status, nullptr, h->B, h->stream);
    if (reward) HIPCHK(h, hipMemcpyAsync(reward, h->d_reward, sizeof(double) * h->B, hipMemcpyDeviceToHost, h->stream));
    return sync_status(h);
}

int mpopis_env_query(mpopis_handle* h, double* reward, int32_t* within, double* dist, double* beta) {
    if (!h) return MPOPIS_ERR_ARG;
    if (h->env.kind == MPOPIS_ENV_CAR && h->env.track.P == 0) { h->err = "track not set"; return MPOPIS_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int B = h->B, NC = std::max(1, h->env.ncars);
    if (!h->d_qdist) { if (dalloc(h, &h->d_qdist, (size_t)B * NC) || dalloc(h, &h->d_qbeta, (size_t)B * NC) || dalloc(h, &h->d_qwithin, B)) return MPOPIS_ERR_HIP; }
    launch_env_query(h->env, h->d_x, h->d_done, h->d_reward, h->d_qwithin, h->d_qdist, h->d_qbeta, B, h->stream);
    if (reward) HIPCHK(h, hipMemcpyAsync(reward, h->d_reward, sizeof(double) * B, hipMemcpyDeviceToHost, h->stream));
    if (within) HIPCHK(h, hipMemcpyAsync(within, h->d_qwithin, sizeof(int) * B, hipMemcpyDeviceToHost, h->stream));
    if (dist) HIPCHK(h, hipMemcpyAsync(dist, h->d_qdist, sizeof(double) * B * NC, hipMemcpyDeviceToHost, h->stream));
    if (beta) HIPCHK(h, hipMemcpyAsync(beta, h->d_qbeta, sizeof(double) * B * NC, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    return MPOPIS_OK;
}

int mpopis_get_trajectories(mpopis_handle* h, double* out) {
    if (!h || !out) return MPOPIS_ERR_ARG;
    if (!h->d_traj) { h->err = "log_trajectories was not enabled"; return MPOPIS_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(out, h->d_traj, sizeof(double) * h->B * h->K * h->T * h->ss, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    return MPOPIS_OK;
}

int mpopis_policy_step(mpopis_handle* h, const mpopis_noise* noise, double* control, double* cost, double* weights,
                       double* E_out, int32_t* resample_idx0, int32_t* iters_run) {
    if (!h) return MPOPIS_ERR_ARG;
    if (h->env.kind == MPOPIS_ENV_CAR && h->env.track.P == 0) { h->err = "track not set"; return MPOPIS_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int B = h->B, K = h->K, cs = h->cs, N = h->N;
    const size_t per = (size_t)cs * K;
    if (noise && noise->Z) {
        if (!h->d_Zin) { if (dalloc(h, &h->d_Zin, (size_t)B * N * per)) return MPOPIS_ERR_HIP; }
        HIPCHK(h, hipMemcpyAsync(h->d_Zin, noise->Z, sizeof(double) * B * N * per, hipMemcpyHostToDevice, h->stream));
        if (h->cfg.policy == MPOPIS_POL_PMCMPPI && N > 1) {
            if (!noise->res_i0 || !noise->res_u) { h->err = "pmcmppi with injected noise needs resampling draws"; return MPOPIS_ERR_ARG; }
            if (!h->d_resi_in) { if (dalloc(h, &h->d_resi_in, (size_t)B * (N - 1) * K) || dalloc(h, &h->d_resu_in, (size_t)B * (N - 1) * K)) return MPOPIS_ERR_HIP; }
            HIPCHK(h, hipMemcpyAsync(h->d_resi_in, noise->res_i0, sizeof(int32_t) * B * (N - 1) * K, hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(h->d_resu_in, noise->res_u, sizeof(double) * B * (N - 1) * K, hipMemcpyHostToDevice, h->stream));
        }
    }
    int rc = h->policy_step_enqueue(noise && noise->Z);
    if (rc) return rc;
    if (control) HIPCHK(h, hipMemcpyAsync(h->h_pin ? h->h_pin : control, h->d_control, sizeof(double) * B * h->as, hipMemcpyDeviceToHost, h->stream));
    if (cost) HIPCHK(h, hipMemcpyAsync(cost, h->d_cost, sizeof(double) * B * K, hipMemcpyDeviceToHost, h->stream));
    if (weights) HIPCHK(h, hipMemcpyAsync(weights, h->d_w, sizeof(double) * B * K, hipMemcpyDeviceToHost, h->stream));
    if (E_out) {
        // E .+= (pol.U - U_orig), returned in the reference layout.  d_Z is free at this point.
        if (h->cfg.policy == MPOPIS_POL_MPPI) {
            // reference E[k,t] (as-vectors, k fastest): element ((t*K+k)*as+a)  <- rows r=t*as+a
            launch_mppi_E_out(h->d_E, h->d_Z, B, h->T, K, h->as, h->stream);
        } else {
            launch_transpose_out(h->d_E, h->d_Ucur, h->d_Uin, h->d_Z, B, cs, K, h->stream);
        }
        HIPCHK(h, hipMemcpyAsync(E_out, h->d_Z, sizeof(double) * B * per, hipMemcpyDeviceToHost, h->stream));
    }
    if (resample_idx0 && N > 1) HIPCHK(h, hipMemcpyAsync(resample_idx0, h->d_residx_log, sizeof(int32_t) * B * (N - 1) * K, hipMemcpyDeviceToHost, h->stream));
    if (iters_run) HIPCHK(h, hipMemcpyAsync(h->h_pin ? (void*)(h->h_pin + (size_t)B * h->as) : (void*)iters_run, h->d_iters, sizeof(int) * B, hipMemcpyDeviceToHost, h->stream));
    rc = sync_status(h);
    if (h->h_pin) {
        if (control) memcpy(control, h->h_pin, sizeof(double) * B * h->as);
        if (iters_run) memcpy(iters_run, h->h_pin + (size_t)B * h->as, sizeof(int) * B);
    }
    HIPCHK(h, hipGetLastError());
    return rc;
}

// control = pol(env) with ONE host wait: the synchronous per-MPC-step call of the reference's harness (src/examples/car_example.jl:203-207,
// mountaincar_example.jl:150-153).  Same results as set_state + set_U + policy_step + get_U (tests/test_gpu_host_api.py); the small inputs and
// outputs travel through the handle's device-mapped mailbox, so the stream carries kernels only.
int mpopis_policy_call(mpopis_handle* h, const double* x, const int32_t* t, const int32_t* done, double* U_inout, const mpopis_noise* noise,
                       double* control, double* cost, double* weights, int32_t* iters_run) {
    if (!h) return MPOPIS_ERR_ARG;
    if (!h->h_call || (noise && noise->Z)) {
        // injected noise is a parity-test path (megabytes of host data): plain composition of the four calls
        int rc = 0;
        if (x || t || done) {
            if (!x) { h->err = "mpopis_policy_call: t / done without x"; return MPOPIS_ERR_ARG; }
            if ((rc = mpopis_set_state(h, x, t, done)) != 0) return rc;
        }
        if (U_inout && (rc = mpopis_set_U(h, U_inout)) != 0) return rc;
        rc = mpopis_policy_step(h, noise, control, cost, weights, nullptr, nullptr, iters_run);
        if (U_inout) { const int r2 = mpopis_get_U(h, U_inout); if (!rc) rc = r2; }
        return rc;
    }
    if ((t || done) && !x) { h->err = "mpopis_policy_call: t / done without x"; return MPOPIS_ERR_ARG; }
    if (h->env.kind == MPOPIS_ENV_CAR && h->env.track.P == 0) { h->err = "track not set"; return MPOPIS_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int B = h->B, K = h->K, cs = h->cs, as = h->as, ss = h->ss;
    const CallBox hb = call_box(h->h_call, B, ss, cs, as), db = call_box(h->d_call, B, ss, cs, as);
    int flags = 0;
    if (x) { memcpy(hb.x, x, sizeof(double) * B * ss); flags |= 1; }
    if (U_inout) { memcpy(hb.U, U_inout, sizeof(double) * B * cs); flags |= 2; }
    if (t) { memcpy(hb.t, t, sizeof(int) * B); flags |= 4; }
    if (done) { memcpy(hb.done, done, sizeof(int) * B); flags |= 8; }
    const int nx = B * ss, nu = B * cs, nmax = std::max(std::max(nx, nu), B);
    if (flags) hipLaunchKernelGGL(k_call_in, dim3((nmax + 255) / 256), dim3(256), 0, h->stream, db, h->d_x, h->d_U, h->d_t, h->d_done, nx, nu, B, flags);
    int rc = h->policy_step_enqueue(false);
    if (rc) return rc;
    const int seq = (int)(++h->call_seq);
    hipLaunchKernelGGL(k_call_out, dim3(1), dim3(256), 0, h->stream, db, h->d_control, h->d_U, h->d_status, h->d_iters,
                       h->coop_disabled ? (const int*)nullptr : h->d_coop_timeouts, B * as, nu, B, seq);
    if (cost) HIPCHK(h, hipMemcpyAsync(cost, h->d_cost, sizeof(double) * B * K, hipMemcpyDeviceToHost, h->stream));
    if (weights) HIPCHK(h, hipMemcpyAsync(weights, h->d_w, sizeof(double) * B * K, hipMemcpyDeviceToHost, h->stream));
    // Host wait.  Without bulk outputs everything the caller gets back sits in the mailbox, published by the sequence word: spin on that word
    // (coherent host memory; the store lands ~1-2 us after the kernel issues it) and leave the stream's completion signal alone -- the next call
    // is ordered behind this one by the stream anyway.  Every ~20 us the stream is queried so that a fault ends the wait; after 3 ms: block.
    // Only when the previous step of this handle was short (wait_should_spin): a long step blocks from the start.
    static const int env_wait = [] { const char* e = getenv("MPOPIS_CALL_WAIT"); return e ? atoi(e) : 1; }();      // 0: runtime wait only (A/B)
    if (env_wait && !cost && !weights && wait_should_spin(h)) {
        const auto t0 = std::chrono::steady_clock::now();
        auto tq = t0;
        for (;;) {
            if (*hb.seq == seq) break;
            const auto now = std::chrono::steady_clock::now();
            if (now - tq > std::chrono::microseconds(20)) {
                tq = now;
                const hipError_t e = hipStreamQuery(h->stream);
                if (e == hipSuccess) break;
                if (e != hipErrorNotReady) HIPCHK(h, e);
                if (now - t0 > std::chrono::milliseconds(3)) { HIPCHK(h, hipStreamSynchronize(h->stream)); break; }
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        wait_note(h, t0);
    } else {
        HIPCHK(h, wait_step(h));
    }
    if (*hb.coop > 0) h->coop_disabled = true;
    if (control) memcpy(control, hb.control, sizeof(double) * B * as);
    if (U_inout) memcpy(U_inout, hb.Uout, sizeof(double) * B * cs);
    if (iters_run) memcpy(iters_run, hb.iters, sizeof(int) * B);
    rc = fold_status(h, hb.status);
    HIPCHK(h, hipGetLastError());
    return rc;
}

int mpopis_set_state_noise(mpopis_handle* h, double sigma_x, double sigma_y, double sigma_psi) {
    if (!h) return MPOPIS_ERR_ARG;
    if (!(sigma_x >= 0.0 && sigma_y >= 0.0 && sigma_psi >= 0.0)) { h->err = "state noise sigmas must be >= 0"; return MPOPIS_ERR_ARG; }
    h->noise_sx = sigma_x; h->noise_sy = sigma_y; h->noise_spsi = sigma_psi;
    return MPOPIS_OK;
}

int mpopis_run_trials(mpopis_handle* h, int32_t num_steps, int32_t laps, double* records, double* actions) {
    if (!h || !records) return MPOPIS_ERR_ARG;
    return h->run_trials(num_steps, laps, records, actions);
}

int mpopis_set_overlap(mpopis_handle* h, int32_t on) {
    if (!h) return MPOPIS_ERR_ARG;
    if (h->split_pinned) return MPOPIS_OK;
    if (on <= 0) { h->nsplit = 1; h->split_auto = true; return MPOPIS_OK; }                   // the default: the engine picks (auto_parts)
    h->nsplit = std::min((int)mpopis_handle::kMaxSplit, (int)on);                            // 1: one stream; 2..4: that many parts
    h->split_auto = false;
    return MPOPIS_OK;
}

int mpopis_timing_enable(mpopis_handle* h, int32_t on) {
    if (!h) return MPOPIS_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (on && h->events.empty()) {
        h->events.resize(2 * 8192);
        // timing events carry no system-scope fence: a default event makes the queue write back / invalidate the caches at every record, which
        // costs the kernels around it (measured in the bench's timed region: two records per rollout launch, ~1 % of the step)
        for (auto& e : h->events) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    }
    h->timing = on != 0;
    h->timing_mask = (on == 1 || on == 0) ? ~0 : (on >> 1);     // 1: every class; otherwise bit (class + 1) selects a class
    return MPOPIS_OK;
}
int mpopis_timing_reset(mpopis_handle* h) { if (!h) return MPOPIS_ERR_ARG; h->ev_used = 0; h->ev_slot.clear(); return MPOPIS_OK; }
int mpopis_timing_read(mpopis_handle* h, char* names, int32_t names_cap, double* ms_total, int64_t* launches, int32_t* n) {
    if (!h || !n) return MPOPIS_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, wait_stream(h->stream));
    const char* const* kNames = kClassNames;
    const int nslots = 7;
    std::vector<double> ms(nslots, 0.0); std::vector<int64_t> cnt(nslots, 0);
    for (size_t i = 0; i < h->ev_slot.size(); ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, h->events[2 * i], h->events[2 * i + 1]) == hipSuccess) { ms[h->ev_slot[i]] += t; cnt[h->ev_slot[i]]++; }
    }
    std::string nm;
    const int m = std::min<int>(*n, nslots);
    for (int i = 0; i < m; ++i) { if (i) nm += ";"; nm += kNames[i]; if (ms_total) ms_total[i] = ms[i]; if (launches) launches[i] = cnt[i]; }
    if (names && names_cap > 0) { strncpy(names, nm.c_str(), names_cap - 1); names[names_cap - 1] = 0; }
    *n = m;
    return MPOPIS_OK;
}

int mpopis_bench_policy_steps(mpopis_handle* h, int32_t steps, double* ms, double* rollouts) {
    if (!h || steps < 0) return MPOPIS_ERR_ARG;
    if (h->env.kind == MPOPIS_ENV_CAR && h->env.track.P == 0) { h->err = "track not set"; return MPOPIS_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    // rollouts EXECUTED, not B * N * K: a CE / CMA iteration loop may break early (src/mppi_mpopi_policies.jl:458-461, :567-569).  k_step_begin folds
    // every step's per-slot iteration count into d_iters_acc before clearing it; the last step's is still in d_iters when the region ends.
    HIPCHK(h, hipMemsetAsync(h->d_iters_acc, 0, sizeof(unsigned long long) * h->B, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_iters, 0, sizeof(int) * h->B, h->stream));
    HIPCHK(h, wait_stream(h->stream));
    HIPCHK(h, hipEventRecord(e0, h->stream));
    for (int s = 0; s < steps; ++s) {
        int rc = h->policy_step_enqueue(false);
        if (rc) return rc;
    }
    HIPCHK(h, hipEventRecord(e1, h->stream));
    HIPCHK(h, hipEventSynchronize(e1));
    float t = 0.f;
    HIPCHK(h, hipEventElapsedTime(&t, e0, e1));
    if (ms) *ms = t;
    if (rollouts) {
        std::vector<unsigned long long> acc(h->B); std::vector<int> last(h->B);
        HIPCHK(h, hipMemcpy(acc.data(), h->d_iters_acc, sizeof(unsigned long long) * h->B, hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(last.data(), h->d_iters, sizeof(int) * h->B, hipMemcpyDeviceToHost));
        double its = 0.0;
        for (int b = 0; b < h->B; ++b) its += (double)acc[b] + (double)last[b];
        *rollouts = its * h->K;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    int rc = sync_status(h);
    HIPCHK(h, hipGetLastError());
    return rc;
}

}  // extern "C"

// ---- launch sequences -----------------------------------------------------------------------------
void mpopis_handle::prepare_state() {
    if (env.kind == MPOPIS_ENV_CAR) launch_extend_state(d_x, d_xext, B, env.ncars, stream, env.track);
}

void mpopis_handle::rollout(const double* Ucur, const double* Uorig, const double* gvec, const int* act, int* iters, int iter_n) {
    RolloutArgs a;
    a.env = env; a.B = B; a.K = K; a.T = T; a.cs = cs;
    a.x0 = d_x; a.x0ext = d_xext; a.t0 = d_t; a.done0 = d_done; a.Ucur = Ucur; a.Uorig = Uorig; a.E = d_E; a.gvec = gvec;
    a.cost = d_cost; a.traj = d_traj; a.active = act; a.iters = iters; a.iter_n = iter_n;
    a.cmin = weights_in_moments ? d_cmin : nullptr; a.status = weights_in_moments ? d_status : nullptr;
    a.share = coop_share;
    time_begin(0);
    launch_rollout(a, stream);
    time_end();
}

// calculate_trajectory_costs(pol, env) for the configured policy + functor tail.  `injected`: noise
// staged in d_Zin / d_resi_in / d_resu_in, else device Philox streams (mpc_step, iteration).
// Slot views: every per-slot buffer is laid out [B][...], so "slots [b0, b0 + nb)" is the same handle with its pointers moved.
void mpopis_handle::shift_slots(ptrdiff_t db) {
    const ptrdiff_t nn = (ptrdiff_t)cs * cs, per = (ptrdiff_t)cs * K;
    auto mv = [db](auto*& p, ptrdiff_t stride) { if (p) p += db * stride; };
    mv(d_x, ss); mv(d_xext, kMaxCars * kCarExt); mv(d_t, 1); mv(d_done, 1);
    mv(d_U, cs); mv(d_Ucur, cs); mv(d_Uin, cs);
    mv(d_Sig, nn); mv(d_L, nn); mv(d_tmpS, nn); mv(d_dscale, cs); mv(d_Lp, (ptrdiff_t)potrf_panel_doubles(cs));
    mv(d_Z, per); mv(d_E, per); mv(d_Zin, (ptrdiff_t)N * per); mv(d_cost, K); mv(d_w, K); mv(d_wsum, 1); mv(d_cmin, 1);
    mv(d_wn, cs); mv(d_mu, cs); mv(d_gvec, cs); mv(d_control, as); mv(d_reward, 1); mv(d_traj, (ptrdiff_t)K * T * ss);
    mv(d_status, 1); mv(d_active, 1); mv(d_iters, 1); mv(d_iters_acc, 1); mv(d_seeds, 1);
    mv(d_order, K); mv(d_resi, K); mv(d_alias, K); mv(d_residx_log, (ptrdiff_t)std::max(1, N - 1) * K); mv(d_resi_in, (ptrdiff_t)(N - 1) * K);
    mv(d_resu, K); mv(d_accept, K); mv(d_alias_need, 1); mv(d_alias_stack, 2 * (ptrdiff_t)K); mv(d_resu_in, (ptrdiff_t)(N - 1) * K);
    mv(d_part, (ptrdiff_t)(wcov_mfma_workspace_doubles(1, cs, ksplit)));
    mv(d_cma_scal, 8); mv(d_cma_vec, 3 * (ptrdiff_t)cs); mv(d_sig2, 1);
    mv(d_lanV, (ptrdiff_t)invsqrt_workspace_doubles(1, cs, lan_regions)); mv(d_lan_x, (ptrdiff_t)invsqrt_coop_words(1, cs)); mv(d_Cdw, cs); mv(d_fro_part, (cs + 15) / 16); mv(d_tri_dinv, (ptrdiff_t)trtri_dinv_doubles(1, cs)); mv(d_fro, 1); mv(d_lan_m, 1); mv(d_lan_prep, (ptrdiff_t)lanczos_prep_doubles(1)); mv(d_tri_cnt, 2);
    mv(d_coop_flags, (ptrdiff_t)potrf_coop_flag_words(1, cs)); mv(d_potrf_redo, 1); mv(d_lan_redo, 1);
    mv(alive_gate, 1);
}

// pol(env) for all slots.  Opt-in (mpopis_set_overlap): with >= 2 slots the batch is split into parts that run as independent chains on
// their own streams, each started one sampler later than the previous: while one part sits in a latency-bound link of its chain
// (Cholesky: nb workgroups on 256 CUs; weights; the scatter's finish), the other half's rollout / sampler fills the chip.
// Per-slot results are bit-identical to the single-stream order (every kernel is slot-independent and deterministic).
// How many part-chains the default schedule uses.  The chain of one AIS iteration (weights -> moments -> Cholesky -> sampler -> rollout) is
// serial per trial, some links are latency-bound (Cholesky: B workgroups on 256 CUs; sort; alias table; finish kernels), and a rollout launch
// that exactly fills the wave slots (64 K=4096-trials = 4096 waves on 4096 slots) leaves them emptying one by one at its tail.  Split into 2-4
// skewed part-chains on their own streams, one part's latency-bound links and launch tails run under another part's throughput kernels.
// Bit-identical per slot (every kernel is slot-independent and deterministic: tests/test_gpu_baseline_shapes.py).  The table is the
// measured optimum (tools/split_sweep.py, one box, K = 4096 and 1024, 16..256 trials, ms per step at 1 / 2 / 3 / 4 parts), by rollout WAVES
// in the batch (B x ceil(K / 64) for one car):
//   :μΣaismppi  32 trials 3.37 / 4.07 / 3.80 / 4.24   48: 4.61 / 4.81 / 4.21 / 4.54   64: 5.72 / 5.75 / 5.63 / 5.36   96: 8.41 / 8.22 / 8.07 / 8.16   128: 10.7 all
//   :cemppi     32: 5.45 / 5.26 / 5.04 / 5.12        48: 7.02 / 6.09 / 5.51 / 5.56   64: 8.78 / 7.79 / 6.94 / 6.66   128: 15.4 / 13.9 / 13.2 / 12.9
//   :pmcmppi    32: 4.69 / 5.46 / 4.71 / 5.28        48: 6.34 / 5.47 / 5.41 / 5.38   64: 7.93 / 6.95 / 6.78 / 6.77   128: 14.7 / 13.3 / 12.8 / 12.6
//   :μaismppi   16: 2.30 / 2.04 / 2.29 / 2.44        32: 3.16 / 2.80 / 2.93 / 3.17   64: 4.96 / 4.37 / 4.57 / 4.49   128: 9.27 / 8.55 / 8.91 / 8.91
//   :gmppi      no gain at any size (one iteration: nothing to hide)
// Below ~2000 waves every kernel of the chain is latency-bound and splitting only adds launches (and takes the two-wave rollout kernels
// out of their regime).  :cmamppi with cs > 128 at >= 48 slots: the per-slot Cholesky / Lanczos kernels (one workgroup per slot: 64 of 256
// CUs busy for ~540 us per iteration) go under the other parts' rollouts: C4 at 64 trials 28.6 -> 26.9 ms (32 trials: no gain).
// mpopis_set_overlap(h, 1..4) overrides (1 = one stream: what a per-kernel profile wants).
int mpopis_handle::auto_parts() const {
    const int pol = cfg.policy;
    const int B = B_full;                                                     // (the member B is a part's slot count while a part-chain is enqueued)
    if (pol == MPOPIS_POL_CMAMPPI) return (cs > 128 && B >= 48) ? 4 : 1;
    if (N <= 1 || cs > 128 || env.kind != MPOPIS_ENV_CAR) return 1;           // one iteration / shapes outside the sweep: one stream
    const long long waves = (long long)B * ((K + 63) / 64);
    int np = 1;
    if (pol == MPOPIS_POL_MUAISMPPI || pol == MPOPIS_POL_IMPPI) np = waves >= 1024 ? 2 : 1;
    else if (pol == MPOPIS_POL_CEMPPI) np = waves >= 4096 ? 4 : (waves >= 2048 ? 3 : 1);
    else if (pol == MPOPIS_POL_PMCMPPI) np = waves >= 3072 ? 4 : 1;
    else if (pol == MPOPIS_POL_MUSIGMAAISMPPI) np = waves >= 4096 ? 4 : (waves >= 3072 ? 3 : 1);
    return std::min(np, B);
}

// spin for `ticks` of the 100 MHz real-time counter (one wave) and leave the kernel's own [start, end] on that clock in ts[0..1]
__global__ void k_spin_ticks(unsigned long long ticks, unsigned long long* ts) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) { ts[0] = t0; ts[1] = __builtin_amdgcn_s_memrealtime(); }
}

void mpopis_handle::verify_part_streams() {
    part_streams_checked = true;
    static const int env_check = [] { const char* e = getenv("MPOPIS_STREAM_CHECK"); return e ? atoi(e) : 1; }();
    if (!env_check) return;
    constexpr double kSpinUs = 200.0;
    // Do one spin kernel per stream run at the same time?  Decided on the DEVICE's clock -- every kernel records its own [start, end] and the
    // intervals must pairwise overlap by more than half a spin -- not on host wall time (until round 5: one host measurement against 1.6 spins, which
    // a descheduled host thread or a busy GPU turns into a false "serialised": extra streams for the handle's lifetime and possibly max_parts = 1).
    // Three probes, majority.
    unsigned long long* d_ts = nullptr;
    if (hipMalloc((void**)&d_ts, sizeof(unsigned long long) * 2 * (kMaxSplit + 1)) != hipSuccess) { (void)hipGetLastError(); return; }
    auto probe = [&](const std::vector<hipStream_t>& ss) {
        for (auto s_ : ss) (void)hipStreamSynchronize(s_);
        for (size_t i = 0; i < ss.size(); ++i) hipLaunchKernelGGL(k_spin_ticks, dim3(1), dim3(64), 0, ss[i], (unsigned long long)(kSpinUs * 100.0), d_ts + 2 * i);
        for (auto s_ : ss) (void)hipStreamSynchronize(s_);
        std::vector<unsigned long long> ts(2 * ss.size());
        if (hipMemcpy(ts.data(), d_ts, sizeof(unsigned long long) * ts.size(), hipMemcpyDeviceToHost) != hipSuccess) return false;
        const long long need = (long long)(0.5 * kSpinUs * 100.0);
        for (size_t i = 0; i < ss.size(); ++i)
            for (size_t j = i + 1; j < ss.size(); ++j) {
                const long long lo = (long long)std::max(ts[2 * i], ts[2 * j]), hi = (long long)std::min(ts[2 * i + 1], ts[2 * j + 1]);
                if (hi - lo < need) return false;
            }
        return true;
    };
    auto concurrent = [&](const std::vector<hipStream_t>& ss) {
        int yes = 0;
        for (int r = 0; r < 3 && yes < 2 && r - yes < 2; ++r) yes += probe(ss) ? 1 : 0;
        return yes >= 2;
    };
    { std::vector<hipStream_t> warm{stream}; (void)concurrent(warm); }   // (first launch of the kernel: code object load)
    std::vector<hipStream_t> chosen{stream};
    int next_x = 0, created = 0;
    while ((int)chosen.size() < kMaxSplit && created < 12) {
        hipStream_t c = nullptr;
        if (next_x < kMaxSplit - 1) c = xstream[next_x++];                 // the streams the handle was created with first
        else { if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) break; ++created; }
        std::vector<hipStream_t> trial = chosen; trial.push_back(c);
        if (concurrent(trial)) chosen.push_back(c); else rejected_streams.push_back(c);
    }
    // the chosen ones first (part-chains), then rejected ones as fillers: side chains (||L^-1||_F, Z prefetch) only need A stream
    size_t r = 0;
    for (int i = 0; i < kMaxSplit - 1; ++i) {
        if (i + 1 < (int)chosen.size()) xstream[i] = chosen[i + 1];
        else if (r < rejected_streams.size()) { xstream[i] = rejected_streams[r]; rejected_streams.erase(rejected_streams.begin() + r); }
    }
    max_parts = std::max(1, (int)chosen.size());
    (void)hipFree(d_ts);
    (void)hipGetLastError();
}

int mpopis_handle::policy_step_enqueue(bool injected) {
    const int B0 = B;
    int np = split_auto ? auto_parts() : std::min(nsplit, B0);
    if (np >= 2 && !part_streams_checked) verify_part_streams();
    np = std::min(np, max_parts);
    if (np < 2) {
        side_free = (xstream[0] != nullptr);
        const int rc = step_enqueue_view(injected, nullptr, nullptr);
        side_free = false;
        mpc_step += 1;
        if (!rc && !launch_err.empty()) { err = launch_err; launch_err.clear(); return MPOPIS_ERR_HIP; }
        return rc;
    }
    (void)hipEventRecord(ev_fork, stream);                      // the other streams start after everything already queued on the main stream
    hipStream_t main_stream = stream;
    coop_share = np;                                            // up to np cluster launches in flight at once: each may take 1/np of the device
    int rc = 0, b0 = 0;
    for (int p = 0; p < np; ++p) {
        const int nbp = B0 / np + (p < B0 % np ? 1 : 0);
        if (p > 0) { stream = xstream[p - 1]; (void)hipStreamWaitEvent(stream, ev_fork, 0); }
        B = nbp;
        // part p starts when part p-1 has entered its first rollout (one sampler later) and tells part p+1 when it gets there itself
        const int r = step_enqueue_view(injected, p > 0 ? ev_skew[p - 1] : nullptr, p + 1 < np ? ev_skew[p] : nullptr);
        if (!rc) rc = r;
        if (p > 0) (void)hipEventRecord(ev_join[p - 1], stream);
        shift_slots(nbp); b0 += nbp;
    }
    shift_slots(-b0); B = B0; stream = main_stream; coop_share = 1;
    for (int p = 1; p < np; ++p) (void)hipStreamWaitEvent(stream, ev_join[p - 1], 0);   // later work on the main stream sees every part
    mpc_step += 1;
    if (!rc && !launch_err.empty()) { err = launch_err; launch_err.clear(); return MPOPIS_ERR_HIP; }
    return rc;
}

int mpopis_handle::step_enqueue_view(bool injected, hipEvent_t wait_first, hipEvent_t record_after_first_sampler) {
    const int pol = cfg.policy;
    if (wait_first) (void)hipStreamWaitEvent(stream, wait_first, 0);
    const size_t nn = (size_t)cs * cs, per = (size_t)cs * K;
    const bool sigma_fixed = (pol == MPOPIS_POL_MPPI || pol == MPOPIS_POL_GMPPI || pol == MPOPIS_POL_IMPPI || pol == MPOPIS_POL_MUAISMPPI);
    // status / active / iters reset, U_orig = pol.U (d_Uin keeps U_orig; d_Ucur is the rebinding pol.U inside the loop), extended start states
    launch_step_begin(status_sticky ? nullptr : d_status, d_active, alive_gate, d_iters, d_U, d_Uin, d_Ucur, B, cs,
                      env.kind == MPOPIS_ENV_CAR ? d_x : nullptr, d_xext, env.ncars, stream, weights_in_moments ? d_cmin : nullptr, env.track, d_iters_acc);
    if (!sigma_fixed) hipLaunchKernelGGL(k_bcast_f64, dim3((nn + 255) / 256), dim3(256), 0, stream, d_Sigma0, d_Sig, nn, B);   // Σ′ = pol.Σ
    if (pol == MPOPIS_POL_CMAMPPI) cma_begin();
    // Shapes the fused sampler does not cover (cs > 128: Z goes through memory anyway) with device RNG, a dense proposal from iteration 2 on
    // and the handle's other streams free: prefetch Z on a side stream (see below).  For fusable shapes the prefetch was measured and LOSES
    // (C5, 64 trials: 6.72 vs 6.45 ms per step: the drawing then competes with the moments kernel instead of filling the MFMA bubbles of
    // the fused kernel); MPOPIS_ZPREFETCH=2 forces it for such A/B runs, 0 disables it.
    static const int env_zpre = [] { const char* e = getenv("MPOPIS_ZPREFETCH"); return e ? atoi(e) : 1; }();
    // (the fork and the wait are two packets of ~6 us each on the main stream: below ~4 M normals per draw -- one or two C4 trials, an 8-12 us kernel -- drawing in line is
    // shorter: 4.20 -> 4.165 ms per C4 step at one trial, 4.55 -> 4.525 at two, neutral at four, +1 % at eight)
    const bool z_prefetch_ok = env_zpre && (env_zpre == 2 || (!sample_trmm_fusable(cs) && (size_t)B * cs * K >= (size_t)4000000)) && side_free && xstream[1] && !injected &&
                               pol != MPOPIS_POL_MPPI && pol != MPOPIS_POL_PMCMPPI /* d_Z holds E[:, idx] there */ && !(sigma_diag && sigma_fixed) && N > 1;
    bool z_prefetched = false;
    for (int n = 1; n <= N; ++n) {
        // ---- P = MvNormal(Σ′): Cholesky; Σ_inv only through gvec --------------------------------
        const double* Lp; size_t Lstride; const double* dsc = nullptr;
        const bool first_diag = sigma_diag && (sigma_fixed || n == 1);
        // :cmamppi draws from MvNormal(σ²Σ′) (:550-554).  In the first iteration Σ′ = pol.Σ and chol(σ²Σ′) = σ chol(pol.Σ): the shared factor of pol.Σ
        // is used like for every other policy and the step size scales the sampler's output (osc2) -- one Cholesky less per MPC step (160 us at
        // cs = 300).  From the second iteration on σ²Σ′ itself is factored, like the reference does: Σ′ drifts towards indefiniteness there
        // (:598 adds negative-weight terms) and whether a last pivot of relative size 1e-16 comes out positive decides between PosDefException now and
        // a numerically singular factor one iteration later -- factoring the unscaled Σ′ flipped exactly that in test_cma_posdef_error_matches_reference_behaviour.
        const bool cma_scaled = pol == MPOPIS_POL_CMAMPPI && N > 1;
        const double* osc2 = (cma_scaled && n == 1) ? cma_sigma2() : nullptr;
        if (sigma_fixed || n == 1) { Lp = d_L0; Lstride = 0; }
        else {
            time_begin(2);
            launch_potrf(d_Sig, nn, d_L, B, cs, cma_scaled ? cma_sigma2() : nullptr, d_status, d_active, stream, potrf_coop(), d_Lp, potrf_panel_doubles(cs));
            time_end();
            Lp = d_L; Lstride = nn;
        }
        cur_L = Lp; cur_Lstride = Lstride; cur_L_scaled = cma_scaled && n > 1;
        if (gamma != 0.0) launch_chol_solve_gvec(Lp, Lstride, d_Uin, gamma, d_gvec, B, cs, d_active, stream, osc2);
        // ---- E = rand(rng, P, K) ----------------------------------------------------------------
        time_begin(1);
        if (first_diag && !(pol == MPOPIS_POL_CMAMPPI)) dsc = d_dscale;        // sqrt(diag Σ) per slot, written by mpopis_set_Sigma
        double* Zdst = dsc ? d_E : d_Z;
        bool fused = false;
        if (injected) {
            // B x N x (cs x K col-major) -> rows; source slot stride N*per
            if (pol == MPOPIS_POL_MPPI) launch_mppi_Z_in(d_Zin, Zdst, B, T, K, as, stream);       // N == 1
            else launch_transpose_in(d_Zin + (size_t)(n - 1) * per, Zdst, B, cs, K, stream, (size_t)N * per);
            if (dsc) launch_scale_rows(Zdst, dsc, B, cs, K, stream);
        } else if (z_prefetched) {
            // this iteration's standard normals were drawn on the side stream while the previous iteration's reweighting / moments / Cholesky
            // kernels (a few workgroups each) left the chip idle: only the matrix product is left on the critical path
            (void)hipStreamWaitEvent(stream, ev_skew[1], 0);
            z_prefetched = false;
        } else if (!dsc && pol != MPOPIS_POL_MPPI) {
            // dense proposal: draw inside the unwhitening kernel when the shape allows it (no Z round trip through HBM)
            fused = launch_sample_trmm_fused(Lp, Lstride, d_E, B, cs, K, d_seeds, (uint32_t)mpc_step, (uint32_t)(n - 1), d_active, stream, d_rng_tab,
                                             Lstride ? d_Lp : d_L0p, Lstride ? potrf_panel_doubles(cs) : (size_t)0, osc2);
            if (!fused) launch_sample_normal(Zdst, B, cs, K, as, 0, d_seeds, (uint32_t)mpc_step, (uint32_t)(n - 1), dsc, d_active, stream, d_rng_tab);
        } else {
            launch_sample_normal(Zdst, B, cs, K, as, pol == MPOPIS_POL_MPPI, d_seeds, (uint32_t)mpc_step, (uint32_t)(n - 1), dsc, d_active, stream, d_rng_tab);
        }
        if (!dsc && !fused) launch_trmm_LZ_mfma(Lp, Lstride, d_Z, d_E, B, cs, K, d_active, stream, osc2);
        time_end();
        // (queued BEHIND the sampler / L.Z launch of this iteration: the event record that forks the side chain costs the main stream ~6-10 us, and there it
        // sits under the L.Z kernel instead of between the Cholesky and L.Z on the critical path; the trace still has ~200 us of slack before the Lanczos kernel needs it)
        {
            // :cmamppi: tr(Σ^-1) (= ||L^-1||_F², times σ² when L factors σ²Σ) of THIS iteration's update needs only the factor this iteration samples from -- start it now on the free
            // second stream (low wave priority: it shares the chip with the sampler and the rollout, which at small batches leave most CUs idle), so
            // that the Lanczos kernel of the update never waits for it.  It used to start behind the rollout, beside the sort, and part of it stayed on
            // the critical path: C4 6.56 -> 5.98 ms per step at one trial, 8.25 -> 7.70 at 8, 11.2 -> 10.8 at 16, neutral from 32 on
            // (MPOPIS_TRTRI_EARLY=0 restores the old placement for A/B runs).
            static const int env_early = [] { const char* e = getenv("MPOPIS_TRTRI_EARLY"); return e ? atoi(e) : 1; }();
            trtri_early = false;
            if (env_early && pol == MPOPIS_POL_CMAMPPI && side_free && n < N) {
                (void)hipEventRecord(ev_skew[2], stream);
                (void)hipStreamWaitEvent(xstream[0], ev_skew[2], 0);
                launch_trtri_fro(Lp, Lstride, d_fro_part, B, cs, nullptr, xstream[0], d_tri_dinv, false, d_Sig, cur_L_scaled ? d_sig2 : nullptr, d_lan_prep, d_tri_cnt);   // + the Lanczos run's spectrum bounds / quadrature nodes (Σ and σ² do not change before the update consumes them)
                (void)hipEventRecord(ev_join[0], xstream[0]);
                trtri_early = true;
            }
        }
        if (n == 1 && record_after_first_sampler) (void)hipEventRecord(record_after_first_sampler, stream);   // the next part starts when this one enters its first rollout
        // ---- trajectory_cost = simulate_model(pol, env, E, Σ_inv, U_orig) -------------------------
        rollout(d_Ucur, d_Uin, gamma != 0.0 ? d_gvec : nullptr, d_active, d_iters, n);   // also records iters_run = n for the active slots
        fork_recorded = false;
        if (z_prefetch_ok && n < N) {
            // Z of iteration n+1 depends on nothing but (seed, MPC step, iteration): draw it now, beside the latency-bound kernels that follow
            (void)hipEventRecord(ev_fork, stream);                      // one fork point behind the rollout for both side chains (an event
            fork_recorded = true;                                       // record on the main stream costs ~6 us of its critical path)
            (void)hipStreamWaitEvent(xstream[1], ev_fork, 0);
            // (no `active` predicate: the main stream's sort may clear active[b] concurrently; Z of a slot that stops is simply not consumed)
            launch_sample_normal(d_Z, B, cs, K, as, 0, d_seeds, (uint32_t)mpc_step, (uint32_t)n, nullptr, nullptr, xstream[1], d_rng_tab);
            (void)hipEventRecord(ev_skew[1], xstream[1]);
            z_prefetched = true;
        }
        // ---- adapt (μ, Σ′) ----------------------------------------------------------------------
        if (n < N) {
            int rc = ais_update(n, injected);
            if (rc) return rc;
        }
    }
    // weights = compute_weights(IT(λ), cost); weighted_noise = Σ_k w_k (E_k + (pol.U - U_orig)); roll
    time_begin(3);
    launch_weights(d_cost, d_w, B, K, cfg.lambda, alive_gate, d_status, stream);
    launch_wmean(d_E, d_w, d_Ucur, d_Uin, d_wn, B, cs, K, 0, alive_gate, stream);
    launch_finalize_env(d_wn, d_U, d_control, B, cs, as, T, env, stream);
    time_end();
    return MPOPIS_OK;
}
