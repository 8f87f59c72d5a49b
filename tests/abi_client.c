/* abi_client.c -- a plain C99 consumer of include/mpopis.h: what a non-Python, non-Julia host links against.
 * Part 1 runs a short MountainCar closed loop (env defaults live in the library, no track needed) with :cemppi on the device RNG and prints
 * every control and reward with full precision; tests/test_abi.py compiles it (gcc -std=c99 -pedantic: the header is valid C) and
 * tests/test_gpu_host_api.py runs it on the GPU and compares the text with the same calls made through the Python mirror.
 * Part 2 is the reference harness' call order against a C mirror of julia/MPOPISHip.jl's handle logic (`binding` below): construct the policy,
 * seed!(pol, seed + k), THEN the first pol(env) -- src/examples/car_example.jl:172-188 precede :205 -- with the engine handle created lazily
 * inside that first call, one mpopis_policy_call per MPC step, the host owning state and pol.U.  The whole harness runs twice and the
 * controls must repeat bit for bit; a seed parked before the handle exists must reach the device streams.  Compiled with
 * -DBINDING_DROPS_EARLY_SEED the mirror behaves like the round-3 binding (early seed dropped, creation seed from the clock) and this
 * program must FAIL -- tests/test_gpu_host_api.py builds both.
 *   build: gcc -std=c99 -pedantic -Wall -Iinclude tests/abi_client.c -Lmpopis_amd/lib -lmpopis_hip -Wl,-rpath,$PWD/mpopis_amd/lib -o abi_client */
#include <stdio.h>
#include <string.h>
#include <time.h>
#include "mpopis.h"

/* ---- C mirror of julia/MPOPISHip.jl: one `binding` = one policy object ------------------------------------------------------------- */
typedef struct {
    mpopis_handle *h;            /* HANDLES[pol]; NULL until the first pol(env)                     */
    int have_pending;            /* haskey(PENDING_SEED, pol)                                       */
    uint64_t pending;            /* PENDING_SEED[pol]                                               */
    uint64_t rng_state;          /* stands for pol.rng (default_seed draws from a copy of it)       */
    double U[15];                /* pol.U (cs = 15), owned by the host like the Julia array         */
} binding;

static void binding_construct(binding *b, uint64_t rng_state) { memset(b, 0, sizeof *b); b->rng_state = rng_state; }

/* MPOPIS.seed!(pol, seed): seeds pol.rng and either forwards to the device or parks the seed until the handle exists */
static int binding_seed(binding *b, uint64_t seed) {
    b->rng_state = seed;
#ifdef BINDING_DROPS_EARLY_SEED
    if (b->h) return mpopis_seed(b->h, seed - 1);          /* round-3 logic: `haskey(HANDLES, pol) && ...`, nothing kept otherwise */
    return 0;
#else
    if (b->h) return mpopis_seed(b->h, seed - 1);
    b->pending = seed - 1; b->have_pending = 1;
    return 0;
#endif
}

static int binding_handle(binding *b) {
    mpopis_config cfg;
    double Sigma[1] = {1.0};
    int rc;
    if (b->h) return 0;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0; cfg.env_kind = MPOPIS_ENV_MOUNTAINCAR; cfg.policy = MPOPIS_POL_CEMPPI; cfg.num_samples = 64; cfg.horizon = 15; cfg.batch = 1;
    cfg.ais_its = 4; cfg.sigma_est = MPOPIS_SIGMA_EST_MLE; cfg.lambda = 0.1; cfg.alpha = 1.0; cfg.elite_threshold = 0.8; cfg.cma_sigma = 1.0;
#ifdef BINDING_DROPS_EARLY_SEED
    cfg.seed = (uint64_t)clock() * 2654435761u + (uint64_t)time(NULL) + (uint64_t)(size_t)b;     /* rand(UInt64) */
#else
    cfg.seed = b->have_pending ? b->pending : b->rng_state * 6364136223846793005ULL + 1442695040888963407ULL;   /* default_seed(pol) */
    b->have_pending = 0;
#endif
    if ((rc = mpopis_create(&cfg, &b->h)) != MPOPIS_OK) return rc;
    return mpopis_set_Sigma(b->h, Sigma, 1);
}

/* act = pol(env): ONE ABI call with one host wait */
static int binding_call(binding *b, const double *x, int32_t t, int32_t done, double *control) {
    int rc = binding_handle(b);
    if (rc) return rc;
    return mpopis_policy_call(b->h, x, &t, &done, b->U, NULL, control, NULL, NULL, NULL);
}

static void binding_finalize(binding *b) { if (b->h) mpopis_destroy(b->h); b->h = NULL; }

/* the trial loop of the reference harness (src/examples/car_example.jl:170-207; mountaincar_example.jl:125-153), 2 trials x 3 MPC steps.
 * The env lives in a second handle (`envh`, policy unused) that plays the host's Julia env: state out, env(act) in. */
#define HARNESS_TRIALS 2
#define HARNESS_STEPS 3
static int harness(uint64_t seed, int late_reseed, double *controls /* TRIALS*STEPS */, int verbose) {
    int k, s, rc;
    for (k = 1; k <= HARNESS_TRIALS; ++k) {
        mpopis_config ecfg;
        mpopis_handle *envh = NULL;
        binding pol;
        memset(&ecfg, 0, sizeof ecfg);
        ecfg.env_kind = MPOPIS_ENV_MOUNTAINCAR; ecfg.policy = MPOPIS_POL_GMPPI; ecfg.num_samples = 1; ecfg.horizon = 1; ecfg.batch = 1; ecfg.lambda = 1.0; ecfg.alpha = 1.0;
        if ((rc = mpopis_create(&ecfg, &envh)) != MPOPIS_OK) return rc;                /* env = MountainCarEnv(...) */
        binding_construct(&pol, 12345u);                                               /* pol = get_policy(...)     */
        if ((rc = binding_seed(&pol, seed + (uint64_t)k)) != 0) return rc;             /* seed!(pol, seed + k)      */
        for (s = 0; s < HARNESS_STEPS; ++s) {
            double x[2], act[1], rew[1];
            int32_t t[1], done[1];
            if ((rc = mpopis_get_state(envh, x, t, done)) != 0) return rc;             /* state(env)                */
            if ((rc = binding_call(&pol, x, t[0], done[0], act)) != 0) return rc;      /* act = pol(env)            */
            if (late_reseed && s == 0) {                                               /* seed! AFTER the handle exists must also work */
                if ((rc = binding_seed(&pol, seed + (uint64_t)k)) != 0) return rc;
                memset(pol.U, 0, sizeof pol.U);
                if ((rc = binding_call(&pol, x, t[0], done[0], act)) != 0) return rc;
            }
            if ((rc = mpopis_env_step(envh, act, rew)) != 0) return rc;                /* env(act)                  */
            controls[(k - 1) * HARNESS_STEPS + s] = act[0];
            if (verbose) printf("harness trial %d step %d x %.17g %.17g control %.17g U0 %.17g\n", k, s, x[0], x[1], act[0], pol.U[0]);
        }
        binding_finalize(&pol);
        mpopis_destroy(envh);
    }
    return 0;
}

static int fail(const mpopis_handle *h, const char *what, int rc) {
    fprintf(stderr, "%s failed: %d (%s)\n", what, rc, mpopis_last_error(h));
    return 1;
}

int main(void) {
    mpopis_config cfg;
    mpopis_handle *h = NULL;
    double Sigma[1] = {1.0};
    double control[2], reward[2], cost[2 * 64];
    int32_t iters[2];
    int rc, step;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0; cfg.env_kind = MPOPIS_ENV_MOUNTAINCAR; cfg.num_cars = 0; cfg.policy = MPOPIS_POL_CEMPPI;
    cfg.num_samples = 64; cfg.horizon = 15; cfg.batch = 2; cfg.ais_its = 4; cfg.sigma_est = MPOPIS_SIGMA_EST_MLE;
    cfg.lambda = 0.1; cfg.alpha = 1.0; cfg.lambda_ais = 0.0; cfg.elite_threshold = 0.8; cfg.cma_sigma = 1.0; cfg.seed = 1234;
    printf("abi %d\n", mpopis_abi_version());
    if ((rc = mpopis_create(&cfg, &h)) != MPOPIS_OK) return fail(NULL, "mpopis_create", rc);
    if ((rc = mpopis_set_Sigma(h, Sigma, 1)) != MPOPIS_OK) return fail(h, "mpopis_set_Sigma", rc);
    if ((rc = mpopis_reset(h)) != MPOPIS_OK) return fail(h, "mpopis_reset", rc);
    for (step = 0; step < 5; ++step) {
        if ((rc = mpopis_policy_step(h, NULL, control, cost, NULL, NULL, NULL, iters)) != MPOPIS_OK) return fail(h, "mpopis_policy_step", rc);
        if ((rc = mpopis_env_step(h, control, reward)) != MPOPIS_OK) return fail(h, "mpopis_env_step", rc);
        printf("step %d control %.17g %.17g reward %.17g %.17g cost0 %.17g iters %d %d\n", step, control[0], control[1], reward[0], reward[1], cost[0],
               (int)iters[0], (int)iters[1]);
    }
    mpopis_destroy(h);
    /* ---- part 2: construct, seed!, then pol(env) -- twice; then with the seed re-issued after the handle exists ---- */
    {
        double a[HARNESS_TRIALS * HARNESS_STEPS], b[HARNESS_TRIALS * HARNESS_STEPS], c[HARNESS_TRIALS * HARNESS_STEPS];
        if ((rc = harness(777, 0, a, 1)) != 0 || (rc = harness(777, 0, b, 0)) != 0 || (rc = harness(777, 1, c, 0)) != 0) return fail(NULL, "harness", rc);
        if (memcmp(a, b, sizeof a) != 0) { fprintf(stderr, "harness not reproducible: seed!(pol, seed + k) before the first pol(env) did not reach the device streams\n"); return 2; }
        if (memcmp(a, c, sizeof a) != 0) { fprintf(stderr, "seed! after the handle exists gives a different stream than the parked seed\n"); return 3; }
        if (a[0] == a[HARNESS_STEPS]) { fprintf(stderr, "trials 1 and 2 drew the same noise\n"); return 4; }
        printf("harness reproducible 1\n");
    }
    return 0;
}
