// engine.h -- internal declarations shared by the HIP translation units of libmpopis_hip.so.
//
// Data layout in HBM (per handle, B trial slots, all FP64 unless noted):
//   x      [B][ss]        resident real-env state (+ t[B], done[B] int32)
//   U      [B][cs]        pol.U (nominal control, step-major like the reference)
//   Ucur   [B][cs]        AIS mean inside calculate_trajectory_costs (pol.U rebinding)
//   E      [B][cs][K]     noise, ROW-MAJOR BY CONTROL ROW (K fastest).  The reference stores E as
//                         cs x K column-major (sample k contiguous); here lanes = samples, so the
//                         transpose makes every rollout-kernel load and every reduction over k
//                         coalesced.  The C ABI transposes on upload/download.
//   Sigma  [B][n][n]      proposal covariance, column-major (n = as for :mppi, cs otherwise)
//   Lchol  [B][n][n]      its lower Cholesky factor
//   cost, w [B][K]; status[B] int32; misc per-trial scalars.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <string>
#include <vector>
#include "../../include/mpopis.h"
#include "car_dynamics.h"

namespace mpopis {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: a process that keeps one handle per GPU
//  must configure each large-LDS kernel once on every device it launches on.
// `seen` = the call site's static bit mask of configured device ordinals (atomic: handles may live on different threads).
inline void ensure_dyn_lds(const void* fn, int bytes, std::atomic<unsigned long long>& seen) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (seen.load(std::memory_order_relaxed) & bit) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    seen.fetch_or(bit, std::memory_order_relaxed);
}

// Wave issue priority (s_setprio 0..3, default 0): the latency-bound kernels of the per-slot chain raise it.  Whenever kernels of
// different streams share the chip (the part-chain schedule of policy_step_enqueue, the side chains of the CMA update, several
// handles on one device), the short links then win the SIMD issue arbitration against long-running throughput waves instead of being
// starved by them (measured: k_weights 9 us alone, 155 us next to a rollout kernel at equal priority).
#define MPOPIS_HI_PRIO() __builtin_amdgcn_s_setprio(3)

constexpr int kMaxCars = 4;
constexpr int kMaxAs = 2 * kMaxCars;

struct EnvDesc {
    int kind;        // MPOPIS_ENV_*
    int ncars;
    int ss, as;
    CarParams car;
    McParams mc;
    CpParams cp;
    Track track;     // device pointers
    double lo[kMaxAs], hi[kMaxAs];
};

// the two scalar-action classic-control envs share one code path ("simple" envs): ss = 2 or 4, as = 1
MP_HD void simple_env_step(const EnvDesc& env, double* s, int* t, int* done, double a) {
    if (env.kind == MPOPIS_ENV_CARTPOLE) cp_step(env.cp, s, t, done, a);
    else mc_step(env.mc, s, t, done, a);
}
MP_HD double simple_env_reward(const EnvDesc& env, const double* s, int done) {
    return env.kind == MPOPIS_ENV_CARTPOLE ? cp_reward(done) : mc_reward(env.mc, s, done);
}

// Arguments of the fused rollout kernel (== simulate_model + rollout_model + env step + reward)
struct RolloutArgs {
    EnvDesc env;
    int B, K, T, cs;
    const double* x0;      // [B][ss]   (MountainCar)
    const double* x0ext;   // [B][ncars][kCarExt] car start states + sin/cos + nearest track point (launch_extend_state)
    const int* t0;         // [B] (MountainCar step counter) or nullptr
    const int* done0;      // [B]
    const double* Ucur;    // [B][cs]
    const double* Uorig;   // [B][cs]
    const double* E;       // [B][cs][K]   (G-variants)  /  [B][T*as][K] (:mppi, same thing)
    const double* gvec;    // [B][cs] = (γ U_orig' Σ_inv) or nullptr when γ == 0
    double* cost;          // [B][K]
    double* traj;          // nullptr or [B][K][ss][T] (Julia (T x ss) column-major per sample)
    const int* active;     // nullptr or [B]: slots with active==0 are skipped (AIS early break)
    int* iters;            // nullptr or [B]: iters[b] = iter_n for every slot this launch works on (AIS iterations executed)
    int iter_n;
    unsigned long long* cmin;   // nullptr or [B]: running minimum of the slot's costs as an order-preserving key (cost_key), car kernels only
    int* status;                // with cmin: a non-finite cost sets MPOPIS_ERR_ACTION here (what k_weights reports when it runs)
    int share = 1;              // launches of this size in flight at once (multi-stream schedule): kernel choice goes by the chip's total load
};
// order-preserving map double -> uint64 (unsigned comparison == numeric comparison, NaN sorts last) for atomicMin on costs
__host__ __device__ __forceinline__ unsigned long long cost_key(double v) {
    unsigned long long u; memcpy(&u, &v, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__host__ __device__ __forceinline__ double cost_unkey(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double v; memcpy(&v, &u, 8);
    return v;
}

// Per-slot error codes merge by PRECEDENCE, on the device like on the host (include/mpopis.h: HIP > ACTION > NOT_PD > NUMERIC -- the codes of
// ABI version 1 keep their order and the newer MPOPIS_ERR_NUMERIC never hides one of them).  Every kernel that reports into status[b] goes
// through status_raise: a plain atomicMin would let an earlier NUMERIC (-5) mask a later NOT_PD (-2) / ACTION (-3) of the same slot.
__host__ __device__ __forceinline__ int status_rank(int c) {
    return c == MPOPIS_ERR_HIP ? 4 : c == MPOPIS_ERR_ACTION ? 3 : c == MPOPIS_ERR_NOT_PD ? 2 : c == MPOPIS_ERR_NUMERIC ? 1 : c < 0 ? 5 : 0;
}
__device__ __forceinline__ void status_raise(int* p, int code) {
    int cur = *p;
    while (status_rank(code) > status_rank(cur)) {
        const int seen = atomicCAS(p, cur, code);
        if (seen == cur) break;
        cur = seen;
    }
}

void launch_rollout(const RolloutArgs& a, hipStream_t s);
void launch_extend_state(const double* x, double* xext, int B, int ncars, hipStream_t s, const Track& tk);
// (iters_acc: per-slot running sum of the iteration counts of earlier steps, folded in before iters is cleared; may be null)
void launch_step_begin(int* status, int* active, const int* alive, int* iters, const double* U, double* Uin, double* Ucur, int B, int cs,
                       const double* x, double* xext, int ncars, hipStream_t st, unsigned long long* cmin, const Track& tk, unsigned long long* iters_acc = nullptr);

// compute_weights (utils.jl:79-86) per slot: w = exp(-(1/λ)(c-min c)) / Σ
void launch_weights(const double* cost, double* w, int B, int K, double lambda, const int* active,
                    int* status, hipStream_t s, double* wsum = nullptr);

// out[b][r] = Σ_k w[b][k] * (E[b][r][k] + shift[b][r]) / (norm ? Σ_k w : 1)
void launch_wmean(const double* E, const double* w, const double* shiftA, const double* shiftB,
                  double* out, int B, int cs, int K, int normalize, const int* active, hipStream_t s);

// functor tail: wc = U + wn; control = clamp(wc[1:as]); roll U (utils.jl:88-101)
void launch_finalize_env(const double* wn, double* U, double* control, int B, int cs, int as, int T,
                         const EnvDesc& env, hipStream_t s);

// layout converters between the ABI's cs x K column-major and the engine's [cs][K]
void launch_transpose_in(const double* src_colmajor, double* dst_rows, int B, int cs, int K, hipStream_t s, size_t src_stride = 0 /* 0: cs*K */);
void launch_transpose_out(const double* src_rows, const double* shiftA, const double* shiftB,
                          double* dst_colmajor, int B, int cs, int K, hipStream_t s);

// real env step + reward for the resident envs
void launch_env_step(const EnvDesc& env, double* x, int* t, int* done, const double* action,
                     double* reward, int* status, const int* alive, int B, hipStream_t s);

void launch_env_query(const EnvDesc& env, const double* x, const int* done, double* reward, int* within, double* dist, double* beta, int B, hipStream_t s);

// kernels_sample.hip
void launch_sample_normal(double* Z, int B, int cs, int K, int as, int mppi_order, const uint64_t* seeds,
                          uint32_t slo, uint32_t shi, const double* dscale, const int* active, hipStream_t s, const double* rng_tab);
void launch_rng_tab_init(double* gtab, hipStream_t s);      // philox.h: kRngTabDoubles doubles (Box-Muller tables), once per handle
void launch_sample_resample_draws(int32_t* di, double* du, int B, int K, const uint64_t* seeds, uint32_t slo, uint32_t shi,
                                  const int* active, hipStream_t s);


// Workspace of the cooperative ("cluster") kernels: several workgroups per matrix that exchange through global memory and therefore need
// to be co-resident.  Co-residency is planned (grid <= coop_max_workgroups() of the device) but never assumed: every wait is bounded, a
// cluster that gives up marks its slot in `redo`, counts the event in `timeouts`, and the launch is followed by the one-workgroup kernel
// predicated on `redo` -- a slot is only ever reported failed by the kernel that needs no partner.  The host disables the cooperative
// kernels of a handle after the first recorded time-out (sync_status), so a device that cannot hold the clusters (partitioned GPU, many
// handles / processes) pays the bounded wait once.
struct CoopCtx {
    unsigned long long* flags = nullptr;   // potrf: [B][ceil(n/16)] panel flags; Lanczos: exchange granules (invsqrt_coop_words)
    unsigned long long* epoch = nullptr;   // host-side launch counter (tags)
    int* redo = nullptr;                   // [B] slot must be recomputed by the one-workgroup kernel (set on a time-out, cleared by the fall-back)
    int* timeouts = nullptr;               // [1] time-outs since the handle was created
    int share = 1;                         // cluster launches of this handle that may be in flight at once (multi-stream schedule): each gets 1/share of the device
    bool usable() const { return flags && epoch && redo && timeouts; }
};
int coop_max_workgroups();                 // per current device: its CU count (one 150 KB-LDS workgroup per CU)
// bounded wait of a cluster member for its partners in ticks of the 100 MHz wall clock (default 1 s; MPOPIS_COOP_WAIT_US) and the test hook
// MPOPIS_COOP_TEST_DROP=1: the last workgroup of every cluster leaves at once, i.e. every cluster times out and is redone by the fall-back
unsigned long long coop_wait_ticks();
int coop_test_drop();

// kernels_linalg.hip
size_t potrf_coop_flag_words(int B, int n);
// panel (nullable, n <= kPanelRows): second copy of the factor in the staging layout of the fused sampler (potrf_panel_doubles(n) per slot)
constexpr int kPanelRows = 128;
size_t potrf_panel_doubles(int n);
void launch_potrf(const double* A, size_t Astride, double* L, int B, int n, const double* scale, int* status, int* active, hipStream_t s,
                  const CoopCtx& coop = CoopCtx(), double* panel = nullptr, size_t pstride = 0);
void launch_chol_solve_gvec(const double* L, size_t Lstride, const double* Uorig, double gamma, double* g, int B, int n, const int* active, hipStream_t s,
                            const double* inv_scale2 = nullptr);      // L = chol(Σ) while the proposal is MvNormal(s²Σ) (:cmamppi): g = (s²Σ)^-1 γU = Σ^-1 γU / s²
void launch_gvec_from_inv(const double* Sinv, const double* Uorig, double gamma, double* g, int B, int n, hipStream_t s);
// kernels_mfma.hip
void launch_trmm_LZ_mfma(const double* L, size_t Lstride, const double* Z, double* E, int B, int n, int K, const int* active, hipStream_t s,
                         const double* oscale2 = nullptr);      // oscale2[b] (nullable): E = sqrt(oscale2[b]) L Z
bool sample_trmm_fusable(int n);
bool launch_sample_trmm_fused(const double* L, size_t Lstride, double* E, int B, int n, int K, const uint64_t* seeds, uint32_t slo, uint32_t shi,
                              const int* active, hipStream_t s, const double* rng_tab, const double* panel, size_t pstride, const double* oscale2 = nullptr);
size_t wcov_mfma_workspace_doubles(int B, int cs, int ksplit);
void launch_wcov_mfma(const double* X, const double* w, const int32_t* idx, int m, const double* mu, double* S, double* part,
                      int B, int cs, int K, int ksplit, int sel_batch /* batch the kernel choice goes by: the handle's whole batch when this launch covers one part-chain of it (0: B; -1: compact form, see mpopis_handle::wcov_sel_batch) */,
                      double den, double ridge, const int* active, hipStream_t s,
                      const double* rscale = nullptr, double* mu_out = nullptr, double* u_add = nullptr, const double* wsum = nullptr,
                      const double* cost = nullptr, unsigned long long* cmin = nullptr, double neg_inv_lambda = 0.0, const double* mu_shift = nullptr);
bool wcov_weights_from_cost_ok(int cs, int K, int ksplit);
bool wcov_mfma_can_emit_mean(int cs);
void launch_inv_sd(const double* S, double* rs, int B, int cs, const int* active, hipStream_t s);
void launch_common_shrink(double* S, int B, int cs, int m, int oas, double ridge, const int* active, hipStream_t s);
void launch_fill_f64(double* p, double v, size_t n, hipStream_t s);
void launch_ss_shrink(double* S, const double* Q, double* rs_ws, int B, int cs, int m, double ridge, const int* active, hipStream_t s);
void launch_gather_cols(const double* X, const int32_t* idx, double* Xout, int B, int cs, int K, const int* active, hipStream_t s, double* shift = nullptr);
void launch_gather_mean(const double* X, const int32_t* idx, const double* cw, double* mu, int B, int cs, int K, int m, int divide,
                        const int* active, hipStream_t s);


void launch_gather_mean_strided(const double* X, const int32_t* idx, const double* cw, double* out, size_t out_stride, int B, int cs, int K, int m,
                                const int* active, hipStream_t s);

// kernels_ce.hip: the whole CE proposal update (elite mean, Σ_est covariance + ridge, pol.U += μ′) in one launch for small elite sets
bool ce_cov_small_ok(int cs, int m, int est);
bool ce_sort_fusable(int K);
void launch_ce_cov_small(const double* E, int32_t* order, double* mu, double* S, double* Ucur, int B, int cs, int K, int m, int est, double ridge,
                         int* active, hipStream_t s, const double* cost = nullptr /* non-null (K <= 256): also sortperm(cost) -> order and the elite early break */);

// kernels_select.hip
void launch_sortperm(const double* cost, int32_t* order, int B, int K, int m_elite, int* active, hipStream_t s,
                     double* skey = nullptr, int* done = nullptr);   // + elite early break; skey [B][K] / done [B] (zeroed): workspace of the chip-wide rank sort
void launch_alias_build(const double* w, double* accept, int32_t* alias, int B, int K, const int* active, hipStream_t s, int* need_ws = nullptr,
                        int32_t* stack_ws = nullptr /* B x 2K ints, needed when K > alias_lds_max_K() */);
int alias_lds_max_K();
void launch_alias_sample(const double* accept, const int32_t* alias, const int32_t* di, size_t di_stride, const double* du,
                         int32_t* out, int32_t* log, size_t log_stride, int B, int K, const int* active, hipStream_t s);

// kernels_cma.hip
// kernels_invsqrt.hip: y = A^-1/2 b (Lanczos + quadrature) and fro = tr(A^-1) = scale * ||L^-1||_F^2 with L = chol(scale * A)
constexpr size_t kInvsqrtPadDoubles = 512;
size_t invsqrt_workspace_doubles(int B, int n, int regions_per_slot = 1);
int invsqrt_coop_groups(int B, int n, int share = 1);      // workgroups per matrix the cooperative Lanczos would use for this batch (1: not cooperative)
int invsqrt_max_n();
size_t trtri_dinv_doubles(int B, int n);      // workspace: the inverses of the diagonal 16 x 16 blocks, [B][ceil(n/16)][256]
size_t lanczos_prep_doubles(int B);           // workspace: per slot fro, spectrum bounds, 64 quadrature nodes (left by launch_trtri_fro for launch_lanczos_invsqrt)
void launch_trtri_fro(const double* L, size_t Lstride, double* part, int B, int n, const int* active, hipStream_t s, double* dinv, bool hiprio = true,
                      const double* A = nullptr, const double* scale = nullptr, double* prep = nullptr, unsigned long long* sync2 = nullptr);
void launch_lanczos_invsqrt(const double* A, const double* prep, const double* bvec, size_t bstride,
                            double* V, double* y, double* fro, int* msteps, int B, int n, int* status, const int* active, hipStream_t s,
                            int regions_per_slot = 1, const CoopCtx& coop = CoopCtx());
size_t invsqrt_coop_words(int B, int n);

void launch_cma_begin(double* scal, double* vec, double* sig2, double sigma0, int cs, int B, hipStream_t s);
void launch_cma_paths(const double* Cdw, const double* fro, const double* E, const int32_t* order, const double* ws, double* Ucur, double* scal, double* vec,
                      double* sig2, int B, int cs, int K, int n_iter, const double* consts7, int m_elite, const int* active, hipStream_t s);
void launch_cma_sigma_update(double* Sig, const double* scal, const double* vec, int B, int cs, const double* consts7, int m_elite,
                             const int* active, hipStream_t s);

}  // namespace mpopis
