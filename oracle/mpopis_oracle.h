/*
 * mpopis_oracle.h -- CPU restatement (plain C, FP64) of the MPOPIS rollout-and-reweight path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mpopis_amd/ (the product) may include, link or call
 * this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
 * as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (sisl/MPOPIS v0.2.0, Julia) ships no tests, golden vectors or
 * fixtures for this path, and Julia is not installed in the build image, so the reference could
 * not be executed to generate vectors.  This file is a literal restatement of the reference
 * source (each function cites the file:line it follows, relative to /root/reference) and is
 * anchored by hand-derivable known-answer tests (tests/test_oracle_kat.py) and by an independent
 * NumPy re-derivation of the closed-form pieces (oracle/np_rederive.py).  Third-party semantics
 * (Distributions / StatsBase / PDMats / CovarianceEstimation / ReinforcementLearning.jl) are
 * restated from their published algorithms as recalled; each such spot is marked [3P] and names the
 * upstream package (version from the reference's Project.toml [compat]), file and function it stands for.
 *
 * PINNING IS A ONE-COMMAND JOB once Julia is available: `julia --project=<MPOPIS> tools/gen_golden.jl` writes
 * tests/golden/julia_*.json (inputs, the normals / resampling draws the reference consumed, its outputs), and
 * tests/test_julia_golden.py then checks THIS file against them (it skips, loudly, while they are absent).
 */
#ifndef MPOPIS_ORACLE_H
#define MPOPIS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- environments ------------------------------------------------------------------- */
enum { ORC_ENV_MOUNTAINCAR = 0, ORC_ENV_CAR = 1 /* ncars>=1; ncars>1 == MultiCarRacingEnv */, ORC_ENV_CARTPOLE = 2 };

/* Car parameter vector, 20 doubles: src/envs/car_racing.jl:2-21,68-93 (+ dt, δt :33-34) */
enum { ORC_CP_M = 0, ORC_CP_IZZ, ORC_CP_HCM, ORC_CP_LF, ORC_CP_LR, ORC_CP_CD0, ORC_CP_CD1,
       ORC_CP_CAF, ORC_CP_CAR, ORC_CP_MUF, ORC_CP_MUR, ORC_CP_DMAX, ORC_CP_DDOTMAX,
       ORC_CP_FXMAX, ORC_CP_FXMIN, ORC_CP_LBRAKE, ORC_CP_LDRIVE, ORC_CP_BETALIM,
       ORC_CP_DT, ORC_CP_DDT, ORC_CP_N };
/* MountainCar parameter vector, 8 doubles [3P: RL.jl MountainCarEnvParams, continuous=true] */
enum { ORC_MP_MINPOS = 0, ORC_MP_MAXPOS, ORC_MP_MAXSPEED, ORC_MP_GOALPOS, ORC_MP_GOALVEL,
       ORC_MP_POWER, ORC_MP_GRAVITY, ORC_MP_MAXSTEPS, ORC_MP_N };
/* CartPole parameter vector, 11 doubles [3P: RL.jl CartPoleEnvParams] */
enum { ORC_XP_GRAVITY = 0, ORC_XP_MASSCART, ORC_XP_MASSPOLE, ORC_XP_TOTALMASS, ORC_XP_HALFLENGTH,
       ORC_XP_POLEMASSLENGTH, ORC_XP_FORCEMAG, ORC_XP_DT, ORC_XP_THETATHR, ORC_XP_XTHR, ORC_XP_MAXSTEPS, ORC_XP_N };

typedef struct {
    int kind;            /* ORC_ENV_* */
    int ncars;           /* car: number of cars (1 => CarRacingEnv) */
    int ss, as;          /* state size, action size */
    double params[ORC_CP_N];
    int P;               /* track points (sub-sampled centre line) */
    const double *tx, *ty, *tw; /* borrowed */
    double state[64];    /* ss doubles (8 per car; [x,v] for MountainCar) */
    int t;               /* step counter */
    int done;
} orc_env;

void   orc_car_default_params(double *p20);
void   orc_mountaincar_default_params(double *p8);
void   orc_cartpole_default_params(double *p11);
void   orc_env_init(orc_env *e, int kind, int ncars, const double *params,
                    int P, const double *tx, const double *ty, const double *tw);
void   orc_env_reset(orc_env *e);                       /* deterministic resets; MountainCar: x=-0.5 */
int    orc_env_step(orc_env *e, const double *a);       /* env(a); returns 0, or -3 if action not in space */
double orc_env_reward(const orc_env *e);
void   orc_action_bounds(const orc_env *e, double *lo, double *hi);

double orc_calc_tire_fy(double alpha, double mu, double C_alpha, double fzt, double fxt);
double orc_calc_tire_fz(const double *p, double fx, char tire);
void   orc_car_step(const double *p, double *s8, const double *a2);
int    orc_within_track(int P, const double *tx, const double *ty, const double *tw,
                        const double *pos2, double *dist_out);
double orc_car_reward(const double *p, int P, const double *tx, const double *ty, const double *tw,
                      const double *s8);
double orc_calculate_beta(const double *s8);

/* ---- utils.jl ------------------------------------------------------------------------- */
void   orc_block_diagm(const double *A, int r, int rep, double *B /* (r*rep)^2 col-major */);
void   orc_get_model_controls(const double *lo, const double *hi, int as, double *V, int T);
void   orc_compute_weights(double lambda, const double *cost, int K, double *w);
double orc_rollout_model(orc_env *e, int T, const double *controls /* as x T col-major */,
                         double *traj_log /* NULL or T*ss, row t = state after step t */);
int    orc_m_elite(int K, double threshold);            /* round(Int, K*(1-thr)), half-even */

/* ---- dense linear algebra (stand-ins for LAPACK calls made by PDMats/LinearAlgebra) [3P] -- */
int    orc_cholesky_lower(int n, const double *A, double *L);  /* 0 ok, -2 not PD; col-major */
void   orc_inv_from_chol(int n, const double *L, double *Ainv);
int    orc_sym_eig(int n, const double *A, double *evals, double *V); /* cyclic Jacobi */
int    orc_sym_pow(int n, const double *A, double p, double *out);    /* V diag(l^p) V' */

/* cov(Σ_est, elite') as CEMPPI calls it (src/mppi_mpopi_policies.jl:464; X = cs x m col-major, est = ORC_SIGMA_EST_*): the estimator
 * alone, exported so that tests can check each third-party block against an independent implementation (tests/test_third_party_blocks.py) */
int    orc_cov_estimate(int cs, int m, const double *X, int est, double *mean, double *S);

/* ---- RNG used by BOTH the oracle and the engine for synthetic runs (Philox4x32-10) ------ */
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void   orc_philox_normals(uint64_t seed, uint32_t stream_lo, uint32_t stream_hi,
                          int64_t n, double *out);      /* out[i], i in [0,n) */
void   orc_philox_resample_draws(uint64_t seed, uint32_t stream_lo, uint32_t stream_hi,
                                 int K, int32_t *idx0 /* 0-based uniform in [0,K) */, double *u);

/* ---- alias table [3P: StatsBase.make_alias_table!, Distributions.AliasTable] ------------ */
void   orc_make_alias_table(const double *w, double wsum, int n, double *accept, int32_t *alias0);
void   orc_alias_sample(const double *accept, const int32_t *alias0, int n,
                        const int32_t *draw_i0, const double *draw_u, int m, int32_t *out0);

/* ---- policies -------------------------------------------------------------------------- */
enum { ORC_POL_MPPI = 0, ORC_POL_GMPPI, ORC_POL_IMPPI, ORC_POL_CEMPPI, ORC_POL_CMAMPPI,
       ORC_POL_MUAISMPPI, ORC_POL_MUSIGMAAISMPPI, ORC_POL_PMCMPPI };
enum { ORC_SIGMA_EST_MLE = 0, ORC_SIGMA_EST_SS = 1, ORC_SIGMA_EST_LW = 2, ORC_SIGMA_EST_RBLW = 3, ORC_SIGMA_EST_OAS = 4 };

typedef struct {
    int kind, K, T, as, cs, ss, N;
    double lambda, alpha, lambda_ais, elite_threshold, cma_sigma;
    int sigma_est;
    int nthreads;        /* OpenMP threads over k in simulate_model (reference: Threads.@threads) */
    double *U;           /* cs, owned; aliased U0 (src/utils.jl:96 is a no-op) */
    double *Sigma;       /* mppi: as*as; else cs*cs col-major; owned */
    double lo[16], hi[16];
    /* CMA constants: src/mppi_mpopi_policies.jl:513-525 */
    int m_elite;
    double *ws; double mu_eff, c_sigma, d_sigma, c_Sigma, c1, c_mu, E_cma;
} orc_policy;

typedef struct {
    const double *Z;        /* injected standard normals.  G-variants: iteration n uses
                               Z[n*cs*K ...] as the cs x K column-major randn! matrix;
                               mppi: K*T*as, element ((t*K+k)*as+a) (k fastest, :193)      */
    const int32_t *res_i0;  /* pmc: (N-1)*K 0-based uniform ints   */
    const double *res_u;    /* pmc: (N-1)*K uniforms in [0,1)      */
} orc_noise;

typedef struct {
    double *control;     /* as */
    double *cost;        /* K  */
    double *weights;     /* K  */
    double *E;           /* G: cs*K col-major (after the final shift); mppi: K*T*as like Z */
    int32_t *res_idx0;   /* optional (N-1)*K, 0-based */
    double *Sigma_last;  /* optional cs*cs: proposal covariance used by the LAST executed iteration */
    double *U_last;      /* optional cs: AIS mean (pol.U inside the loop) at exit */
    int iters_run;
    int status;          /* 0, -2 (not PD), -3 (bad action) */
} orc_step_out;

int  orc_policy_create(orc_policy *pol, int kind, const orc_env *env, int K, int T,
                       double lambda, double alpha, const double *U0, int nU0,
                       const double *cov, int ncov /* rows of cov: as or cs */, int cov_is_vector,
                       int N, double lambda_ais, double elite_threshold, int sigma_est,
                       double cma_sigma);
void orc_policy_free(orc_policy *pol);
void orc_simulate_model(const orc_policy *pol, const double *Ucur, const orc_env *env,
                        const double *E, const double *Sigma_inv, const double *U_orig,
                        double *cost, double *traj_log /* NULL or K*T*ss */);
int  orc_policy_call(orc_policy *pol, const orc_env *env, const orc_noise *nz, orc_step_out *out);
void orc_roll_U(orc_policy *pol, const double *wc, double *control);

/* closed loop: src/examples/car_example.jl:170-326, mountaincar_example.jl:125-180 */
typedef struct {
    double rew, steps, rew_per_step, lap_t[4], mean_v, max_v, mean_beta, max_beta,
           beta_viol, trk_viol, crash_viol;
    double rollouts;     /* number of model rollouts executed (for throughput) */
} orc_trial_record;
int  orc_run_trial(orc_policy *pol, orc_env *env, uint64_t seed, int num_steps, int laps,
                   orc_trial_record *rec, double *act_log /* NULL or (num_steps+1)*as */);

int  orc_run_trial_noise(orc_policy *pol, orc_env *env, uint64_t seed, int num_steps, int laps,
                         double sx, double sy, double spsi, orc_trial_record *rec, double *act_log);

void orc_quantile_ci(const double *x, int n, double *lo, double *med, double *hi);

#ifdef __cplusplus
}
#endif
#endif
