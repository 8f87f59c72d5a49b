/*
 * mpopis.h -- C ABI of the MI355X-native MPPI/MPOPI sampling engine (libmpopis_hip.so).
 *
 * Drop-in boundary for the rollout-and-reweight hot path of sisl/MPOPIS
 * (src/mppi_mpopi_policies.jl).  The reference has NO FFI; its plug-in seam is Julia dispatch on
 * (policy type, env type) -- it already specialises these three functions for EnvpoolEnv
 * (src/mppi_mpopi_policies.jl:148,240; src/utils.jl:103).  Each entry point below replaces the
 * reference interface cited next to it; julia/MPOPISHip.jl and INTEGRATION.md show the ccall
 * methods a maintainer adds.  All file:line citations are relative to the reference repo root.
 *
 * Conventions
 *   - plain C: pointers + sizes, FP64 everywhere (the reference hard-codes Float64,
 *     src/mppi_mpopi_policies.jl:3-5,13,110-111), Julia column-major layouts, no repacking.
 *   - a handle owns B = cfg.batch independent "trial slots" (independent env + policy pairs =
 *     the `for k in 1:num_trials` axis of src/examples/car_example.jl:170); every array argument
 *     is the concatenation over slots b = 0..B-1 of the per-trial reference array.
 *   - the caller owns all host arrays; the library copies in/out during the call and never
 *     retains host pointers.  Calls are synchronous (return after the stream has drained).
 *   - return 0 on success; negative codes mirror the reference's error()/exception sites:
 *       MPOPIS_ERR_ARG     (-1) size/argument errors     src/mppi_mpopi_policies.jl:55,64,73,79-80
 *       MPOPIS_ERR_NOT_PD  (-2) PosDefException from MvNormal(Sigma')   :447,551,723,796
 *       MPOPIS_ERR_ACTION  (-3) "Action is not in action space" / NaN   src/envs/car_racing.jl:239
 *       MPOPIS_ERR_HIP     (-4) HIP runtime failure / no device
 *       MPOPIS_ERR_NUMERIC (-5) :cmamppi only: Σ^-0.5 δw (:580-581) could not be formed because Σ, tr(Σ^-1) or δw is not finite.
 *                               (Until ABI 4 also raised when cond(Σ) exceeded the 1e14 the Lanczos quadrature resolves; since ABI 5 such a
 *                               slot is answered by a dense symmetric eigen-solve on the device, like the reference's eigen-based Σ^-0.5,
 *                               and a non-positive eigenvalue there is MPOPIS_ERR_NOT_PD.)
 *     mpopis_last_error() returns a human readable message for the last failure.
 *   - a handle is not thread-safe; distinct handles are independent (own HIP stream).
 */
#ifndef MPOPIS_H
#define MPOPIS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  1: round-1 surface.  2: + mpopis_seed_slots, mpopis_get_Sigma, mpopis_set_overlap, mpopis_comm_unique_id / _init / _destroy,
 * mpopis_gather_summary, and the error code MPOPIS_ERR_NUMERIC.  A version-1 caller keeps working against a version-2 library (nothing was
 * removed or changed in meaning); a caller that needs the newer entry points checks mpopis_abi_version() >= 2.
 * Error precedence when several slots / kernels fail in one call: MPOPIS_ERR_HIP > MPOPIS_ERR_ACTION > MPOPIS_ERR_NOT_PD > MPOPIS_ERR_NUMERIC --
 * the version-1 codes keep their order (numeric minimum) and are never hidden by MPOPIS_ERR_NUMERIC.
 * 3: + mpopis_policy_call (control = pol(env) with a single host wait).  Nothing removed or changed in meaning.
 * 4: no new entry point; the execution knob mpopis_set_overlap changed: the default (on <= 0) is now the engine's own choice of schedule for the
 *    handle's shape instead of one stream, and on = 1 means "one stream" instead of "two halves".  Results never depended on the knob (bit-identical
 *    per slot in every schedule), so a version-3 caller sees the same numbers, sooner.
 * 5: + mpopis_comm_count (the number of ranks RCCL itself reports for the handle's communicator).  Limits lifted, nothing changed in meaning: any K for
 *    :cemppi / :cmamppi / :pmcmppi (were <= 8192 / 7168), and cond(Σ) beyond 1e14 under :cmamppi is computed instead of MPOPIS_ERR_NUMERIC. */
#define MPOPIS_ABI_VERSION 5

enum { MPOPIS_OK = 0, MPOPIS_ERR_ARG = -1, MPOPIS_ERR_NOT_PD = -2, MPOPIS_ERR_ACTION = -3, MPOPIS_ERR_HIP = -4, MPOPIS_ERR_NUMERIC = -5 };

/* env kinds: RL.jl MountainCarEnv(continuous=true) + src/examples/mountaincar_example.jl:4-22;
 * CarRacingEnv src/envs/car_racing.jl (num_cars==1) / MultiCarRacingEnv src/envs/multi-car_racing.jl;
 * RL.jl CartPoleEnv(continuous=true) + src/examples/cartpole_example.jl:3-6 (state [x,xdot,theta,thetadot]) */
enum { MPOPIS_ENV_MOUNTAINCAR = 0, MPOPIS_ENV_CAR = 1, MPOPIS_ENV_CARTPOLE = 2 };

/* policy kinds = get_policy symbols, src/examples/example_utils.jl:20-128 */
enum { MPOPIS_POL_MPPI = 0,            /* :mppi       MPPI_Policy      :107-216 */
       MPOPIS_POL_GMPPI = 1,           /* :gmppi      GMPPI_Policy     :284-315 */
       MPOPIS_POL_IMPPI = 2,           /* :imppi      IMPPI_Policy     :321-373 */
       MPOPIS_POL_CEMPPI = 3,          /* :cemppi     CEMPPI_Policy    :379-472 */
       MPOPIS_POL_CMAMPPI = 4,         /* :cmamppi    CMAMPPI_Policy   :478-606 */
       MPOPIS_POL_MUAISMPPI = 5,       /* :μaismppi   μAISMPPI_Policy  :612-671 */
       MPOPIS_POL_MUSIGMAAISMPPI = 6,  /* :μΣaismppi  μΣAISMPPI_Policy :677-742 */
       MPOPIS_POL_PMCMPPI = 7 };       /* :pmcmppi    PMCMPPI_Policy   :748-817 */

enum { MPOPIS_SIGMA_EST_MLE = 0, MPOPIS_SIGMA_EST_SS = 1, MPOPIS_SIGMA_EST_LW = 2, MPOPIS_SIGMA_EST_RBLW = 3,
       MPOPIS_SIGMA_EST_OAS = 4 };   /* CEMPPI Σ_est :414-426 (:mle exact; shrinkage estimators restate CovarianceEstimation.jl, unpinned) */

/* Car parameter vector (20 doubles) = CarRacingEnvParams fields in declaration order
 * (src/envs/car_racing.jl:2-21) followed by dt, δt (:33-34).  MountainCar: 8 doubles
 * {min_pos,max_pos,max_speed,goal_pos,goal_velocity,power,gravity,max_steps}.  CartPole: 11 doubles
 * {gravity,masscart,masspole,totalmass,halflength,polemasslength,forcemag,dt,thetathreshold,xthreshold,max_steps}. */
#define MPOPIS_CAR_NPARAMS 20
#define MPOPIS_MOUNTAINCAR_NPARAMS 8
#define MPOPIS_CARTPOLE_NPARAMS 11

typedef struct mpopis_handle mpopis_handle;

/* Mirrors the keyword arguments of MPPI_Policy_Params (:36-47) and of the per-policy
 * constructors (:337-340,:407-412,:506-511,:630-634,:695-699,:766-770). */
typedef struct {
    int32_t device;            /* HIP device ordinal                                              */
    int32_t env_kind;          /* MPOPIS_ENV_*                                                    */
    int32_t num_cars;          /* car env: 1 => CarRacingEnv, >1 => MultiCarRacingEnv(N)          */
    int32_t policy;            /* MPOPIS_POL_*                                                    */
    int32_t num_samples;       /* K                                                               */
    int32_t horizon;           /* H (T in the reference code)                                     */
    int32_t batch;             /* B resident independent trials                                   */
    int32_t ais_its;           /* opt_its / ais_its N (ignored for :mppi/:gmppi)                  */
    int32_t sigma_est;         /* MPOPIS_SIGMA_EST_* (:cemppi)                                    */
    int32_t log_trajectories;  /* params.log: keep K x H x ss model trajectories (MPPI_Logger)   */
    double lambda;             /* λ                                                               */
    double alpha;              /* α ; γ = λ(1-α)                                                  */
    double lambda_ais;         /* λ_ais (:μaismppi/:μΣaismppi/:pmcmppi)                           */
    double elite_threshold;    /* ce_elite_threshold / elite_perc_threshold                       */
    double cma_sigma;          /* σ (:cmamppi)                                                    */
    uint64_t seed;             /* trial slot b draws from seed+b+1 (seed!(pol, seed+k), car_example.jl:188) */
} mpopis_config;

/* Injected randomness for results-parity runs (NULL => device Philox4x32-10 streams).
 * The reference draws from Julia's MersenneTwister, which is not reproduced; parity is defined
 * on identical standard normals / resampling draws. */
typedef struct {
    const double *Z;        /* G-variants: B x N x (cs x K col-major randn! matrix, :308,:448,:556,:657,:724,:797)
                               :mppi: B x (K*T*as), element ((t*K+k)*as+a)  (rand(rng,P,K,T), :193)  */
    const int32_t *res_i0;  /* :pmcmppi: B x (N-1) x K uniform ints in [0,K) (0-based)  (:805)   */
    const double *res_u;    /* :pmcmppi: B x (N-1) x K uniforms in [0,1)                          */
} mpopis_noise;

/* ---- lifetime ---------------------------------------------------------------------------- */
int  mpopis_abi_version(void);
const char *mpopis_last_error(const mpopis_handle *h);     /* h may be NULL: last create() error  */

/* XPolicy(env; kwargs...) + env construction: src/mppi_mpopi_policies.jl:116-119,298-301,...;
 * src/examples/car_example.jl:172-185.  Defaults after create: env params = reference
 * defaults, track = none (must be set for car envs), bounds = [-1,1], U = 0, Sigma = I. */
int  mpopis_create(const mpopis_config *cfg, mpopis_handle **out);
void mpopis_destroy(mpopis_handle *h);

/* ---- env description (the env protocol the path consumes) ------------------------ */
int  mpopis_set_env_params(mpopis_handle *h, const double *p, int32_t n);   /* env.params, env.dt, env.δt */
int  mpopis_set_track(mpopis_handle *h, const double *x, const double *y, const double *w, int32_t P);
                                                                  /* env.track.{x′,y′,lane_width′} car_racing_tracks.jl:21-23 */
int  mpopis_set_action_bounds(mpopis_handle *h, const double *lo, const double *hi); /* action_space(env) */
int  mpopis_reset(mpopis_handle *h);                       /* reset!(env) per slot: car_racing.jl:215-223, multi :160-180 */
int  mpopis_set_state(mpopis_handle *h, const double *x /* B*ss */, const int32_t *t, const int32_t *done);
int  mpopis_get_state(mpopis_handle *h, double *x /* B*ss */, int32_t *t, int32_t *done);

/* ---- policy state ---------------------------------------------------------------------------- */
int  mpopis_set_U(mpopis_handle *h, const double *U /* B*cs */);            /* pol.U                */
int  mpopis_get_U(mpopis_handle *h, double *U /* B*cs */);
int  mpopis_set_Sigma(mpopis_handle *h, const double *Sigma, int32_t n);    /* pol.Σ : n = as (mppi, or block-replicated :76-78) or cs; col-major, shared by all slots */
int  mpopis_seed(mpopis_handle *h, uint64_t seed);                          /* seed!(pol, seed) src/MPOPIS.jl:54 */
/* Per-slot seeds: slot b draws from seeds[b] (B values).  The reference seeds trial k with seed!(pol, seed + k),
 * src/examples/car_example.jl:187-188; a rank that holds the trials k0, k0+G, k0+2G, ... of a sharded run
 *  passes {seed + k0, seed + k0 + G, ...} so that all of them stay in ONE resident batch. */
int  mpopis_seed_slots(mpopis_handle *h, const uint64_t *seeds /* B */);
/* Σ′ of the proposal MvNormal the LAST executed AIS iteration of the last policy step drew from, per slot:
 * B x (cs x cs col-major).  (:447,:723,:796 `P = MvNormal(Σ′)`; :cmamppi with N > 1: σ²·Σ′, :550-554; policies
 * that keep pol.Σ fixed return pol.Σ.)  Lets a caller compare the adapted covariance, which the reference
 * only exposes indirectly through the next draw. */
int  mpopis_get_Sigma(mpopis_handle *h, double *Sigma_out /* B*cs*cs */);

/* ---- Level 1: simulate_model(pol, env, E, Σ_inv, U_orig) -> trajectory_cost  (:261-278) ------
 * x0: B*ss (NULL: resident env state); U: B*cs current AIS mean (pol.U); U_orig: B*cs (NULL => U);
 * E: B x (cs x K col-major); Sigma_inv: cs x cs col-major or NULL (required only when γ != 0);
 * cost: B*K out. */
int  mpopis_rollout_costs(mpopis_handle *h, const double *x0, const double *U, const double *U_orig,
                          const double *E, const double *Sigma_inv, double *cost);

/* ---- Level 2: control = pol(env)  (:121-146 for :mppi, :221-238 for the G-variants) ------------
 * Runs calculate_trajectory_costs for the configured policy, the weighted control update and
 * get_controls_roll_U! (src/utils.jl:88-101) for every slot.  Resident env state is used (set it
 * with mpopis_set_state) and pol.U is rolled in place on the device.
 * Outputs (any may be NULL): control B*as; cost B*K; weights B*K; E_out B x (cs x K) after the
 * final shift (:468,:602,:668,:739,:814) [:mppi: B x K*T*as]; resample_idx0 B x (N-1) x K 0-based;
 * iters_run B (AIS iterations executed, CE/CMA may break early :459-461,:567-569). */
int  mpopis_policy_step(mpopis_handle *h, const mpopis_noise *noise,
                        double *control, double *cost, double *weights, double *E_out,
                        int32_t *resample_idx0, int32_t *iters_run);

/* control = pol(env) exactly as the reference's harness calls it once per MPC step, synchronously, for one env the HOST owns
 * (src/examples/car_example.jl:203-207 `act = pol(env); env(act)`; mountaincar_example.jl:150-153): in ONE call and ONE host wait
 *   x, t, done   -> state(env), env.t, env.done of every slot (B*ss, B, B; x NULL: keep the resident state; t / done may be NULL)
 *   U_inout      -> in: pol.U (B*cs) before the call; out: the rolled pol.U (get_controls_roll_U!, src/utils.jl:88-101).  NULL: the
 *                   resident pol.U is used and rolled on the device only
 *   noise        -> NULL: device Philox streams (the production path); non-NULL: as mpopis_policy_step
 *   control B*as, cost B*K, weights B*K, iters_run B: outputs, any may be NULL
 * Equivalent to mpopis_set_state + mpopis_set_U + mpopis_policy_step + mpopis_get_U (four host waits); the small arrays travel through a
 * pinned, device-mapped mailbox owned by the handle instead of copy commands. */
int  mpopis_policy_call(mpopis_handle *h, const double *x, const int32_t *t, const int32_t *done, double *U_inout,
                        const mpopis_noise *noise, double *control, double *cost, double *weights, int32_t *iters_run);

/* env(action) for the resident real envs + reward(env): src/envs/car_racing.jl:238-250,201-213;
 * multi-car_racing.jl:200-207,145-158; mountaincar_example.jl:4-22.  action B*as, reward B out. */
int  mpopis_env_step(mpopis_handle *h, const double *action, double *reward);

/* reward(env), within_track(env), calculate_β(env) for the resident envs WITHOUT stepping
 * (src/envs/car_racing.jl:178-213; car_racing_tracks.jl:68-92; multi-car_racing.jl:122-158).
 * reward B; within B (car: all cars inside the lane, multi-car :122-128; MountainCar: 1);
 * dist / beta: B x num_cars (distance to the centre line, atan(Vy,Vx)); any pointer may be NULL. */
int  mpopis_env_query(mpopis_handle *h, double *reward, int32_t *within, double *dist, double *beta);

/* pol.logger.trajectories (src/mppi_mpopi_policies.jl:92-95, src/utils.jl:139-141): B x K x (H x ss),
 * state after each model step of the LAST simulate_model call; needs cfg.log_trajectories. */
int  mpopis_get_trajectories(mpopis_handle *h, double *out);

/* ---- Level 3: the closed-loop trial harness, device resident ------------------------------------
 * simulate_car_racing / simulate_mountaincar trial loop (src/examples/car_example.jl:170-326,
 * mountaincar_example.jl:125-180) for all B slots at once, device RNG, no host round trips per
 * MPC step.  One record per slot (MPOPIS_RECORD_LEN doubles):
 *   [0] rew [1] steps [2] rew/step [3..6] lap_t[1..4] [7] mean_v [8] max_v [9] mean_β [10] max_β
 *   [11] β_viol [12] T_viol [13] C_viol [14] rollouts executed [15] status            (:144-155,287-302) */
#define MPOPIS_RECORD_LEN 16
/* state_x_sigma, state_y_sigma, state_ψ_sigma of simulate_car_racing (src/examples/car_example.jl:38-40,224-236): after
 * every real-env step of mpopis_run_trials, single-car slots get x += σx z0, y += σy z1, ψ += δψ (δψ = σψ z2) and (Vx, Vy)
 * rotated passively by δψ.  z from the slot's Philox stream (mpc_step, 0x40000000).  Default 0 (no draws).  Multi-car and
 * non-car envs: ignored, like the reference (`sim_type == :cr` only). */
int  mpopis_set_state_noise(mpopis_handle *h, double sigma_x, double sigma_y, double sigma_psi);
int  mpopis_run_trials(mpopis_handle *h, int32_t num_steps, int32_t laps, double *records /* B*16 */,
                       double *actions /* NULL or B x (num_steps+1) x as */);

/* ---- the one collective: summary records to rank 0 over RCCL/xGMI ---------------------
 * Trials shard over GPUs at trial granularity: trial k -> rank (k-1) mod G (`for k in 1:num_trials`,
 * src/examples/car_example.jl:170, has no cross-trial dependency), each rank runs its trials as one resident batch
 * (mpopis_seed_slots + mpopis_run_trials) and nothing is exchanged until the per-trial records
 * (src/examples/car_example.jl:144-155,287-302) are gathered for the AVE/STD/MED/... table (:328-410).
 *   mpopis_comm_unique_id : rank 0 creates the 128-byte RCCL id; the host distributes it (MPI / Distributed.jl / a file)
 *   mpopis_comm_init      : every rank, same id; binds an RCCL communicator to the handle's device and stream
 *                           (world == 1 with id == NULL: no RCCL is loaded, the gather is a copy;
 *                           world == 1 with an id: a real one-rank RCCL communicator)
 *   mpopis_gather_summary : every rank passes its n_local records (rows of MPOPIS_RECORD_LEN doubles); n_max = the largest
 *                           n_local over ranks (ceil(num_trials / world)).  Rank 0 receives out[world][n_max][RECORD_LEN]
 *                           and counts[world] (rows valid per rank); other ranks may pass NULL for both.
 *   mpopis_comm_count     : *ranks = what ncclCommCount reports for the handle's communicator, 0 when no RCCL communicator is bound
 *                           (lets a launcher prove the gather really spans its N ranks; replaces nothing in the reference, whose
 *                           trial loop is serial: src/examples/car_example.jl:170)
 * librccl is bound with dlopen at the first comm call; single-GPU users never load it. */
#define MPOPIS_COMM_ID_BYTES 128
int  mpopis_comm_unique_id(char *id128);
int  mpopis_comm_init(mpopis_handle *h, const char *id128, int32_t rank, int32_t world);
int  mpopis_gather_summary(mpopis_handle *h, const double *records, int32_t n_local, int32_t n_max,
                           double *out, int32_t *counts);
int  mpopis_comm_count(mpopis_handle *h, int32_t *ranks);
int  mpopis_comm_destroy(mpopis_handle *h);

/* Execution knob.  A handle can split its batch into parts that run as independent chains on their own HIP streams, so that
 * the latency-bound links of one chain (Cholesky, sort, alias table) and the tail of its rollout launch hide under another chain's
 * throughput kernels: 5-30 % of an MPC step for the adaptive policies from ~48 resident K = 4096 trials on, nothing below.
 * Default (on <= 0): the engine picks the measured optimum for the handle's shape (one stream for small batches and for :mppi / :gmppi).
 * on = 1: one stream (what a per-kernel profile wants); 2..4: that many parts.  (Until round 4 the default was one stream and on = 1
 * meant two halves.)  Results do not depend on it (bit-identical per slot). */
int  mpopis_set_overlap(mpopis_handle *h, int32_t on);

/* ---- measurement hooks (bench.py; HIP events on the engine's own stream) ----------------------- */
int  mpopis_timing_enable(mpopis_handle *h, int32_t on);   /* 0 off; 1 every kernel class; else a mask: bit (i + 1) = class i of
                                                              * mpopis_timing_read's name list (2 = "rollout" only: 2 events per launch) */
/* accumulated since enable/reset: per kernel class average launch duration.
 * names: semicolon separated list; ms_total[i], launches[i] for i < *n (in: capacity). */
int  mpopis_timing_read(mpopis_handle *h, char *names, int32_t names_cap, double *ms_total,
                        int64_t *launches, int32_t *n);
int  mpopis_timing_reset(mpopis_handle *h);

/* Synthetic-workload entry used by bench.py: run `steps` policy steps (device RNG, resident
 * state, env not advanced) back to back on the stream; returns wall ms measured with HIP events
 * around the whole region and the number of model rollouts executed. */
int  mpopis_bench_policy_steps(mpopis_handle *h, int32_t steps, double *ms, double *rollouts);

#ifdef __cplusplus
}
#endif
#endif
