/*
 * mpopis_oracle.c -- CPU restatement of the MPOPIS rollout-and-reweight path (see header).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (no reference vectors exist, Julia unavailable).
 * Citations "file:line" are relative to /root/reference.  Compile with -ffp-contract=off so the
 * arithmetic is evaluated operation-by-operation as the Julia source is (Julia does not fuse).
 */
#include "mpopis_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Julia Base.sign: sign(0)=0, sign(-0.0)=-0.0, NaN->NaN */
static inline double jl_sign(double x) { return (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : x); }
static inline double jl_min(double a, double b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }
static inline double jl_max(double a, double b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }
static inline double jl_clamp(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }
static inline int mod1(int x, int n) { int r = ((x - 1) % n + n) % n; return r + 1; }

/* ======================================================================================
 * Car racing: src/envs/car_racing.jl
 * ====================================================================================== */

/* defaults: src/envs/car_racing.jl:68-93 */
void orc_car_default_params(double *p) {
    p[ORC_CP_M] = 2000.0;  p[ORC_CP_IZZ] = 3764.0; p[ORC_CP_HCM] = 0.3;
    p[ORC_CP_LF] = 1.53;   p[ORC_CP_LR] = 1.23;
    p[ORC_CP_CD0] = 241.0; p[ORC_CP_CD1] = 25.1;
    p[ORC_CP_CAF] = 150000.0; p[ORC_CP_CAR] = 280000.0;
    p[ORC_CP_MUF] = 0.9;   p[ORC_CP_MUR] = 0.9;
    p[ORC_CP_FXMAX] = 7200.0; p[ORC_CP_FXMIN] = 22500.0;
    p[ORC_CP_LBRAKE] = 0.6; p[ORC_CP_LDRIVE] = 0.0;
    p[ORC_CP_DT] = 0.1; p[ORC_CP_DDT] = 0.01;
    /* Julia: deg2rad(z::AbstractFloat) = z * (oftype(z,pi)/180)  -> z*(M_PI/180.0), this order */
    p[ORC_CP_DMAX] = 18.0 * (M_PI / 180.0);
    p[ORC_CP_DDOTMAX] = 90.0 * (M_PI / 180.0);
    p[ORC_CP_BETALIM] = 45.0 * (M_PI / 180.0);
}

/* src/envs/car_racing.jl:252-260 */
double orc_calc_tire_fy(double alpha, double mu, double C_alpha, double fzt, double fxt) {
    double fy_max = sqrt(jl_max((mu * fzt) * (mu * fzt) - fxt * fxt, 1e-8));
    double ta = tan(alpha);
    if (fabs(alpha) < atan(3 * fy_max / C_alpha)) {
        return -C_alpha * ta + ((C_alpha * C_alpha) / (3 * fy_max)) * fabs(ta) * ta
               - ((C_alpha * C_alpha * C_alpha) / (27 * (fy_max * fy_max))) * (ta * ta * ta);
    } else {
        return -fy_max * jl_sign(alpha);
    }
}

/* src/envs/car_racing.jl:262-272 */
double orc_calc_tire_fz(const double *p, double fx, char tire) {
    double mass = p[ORC_CP_M], l_t = p[ORC_CP_LF], h_cm = p[ORC_CP_HCM];
    double L = p[ORC_CP_LR] + p[ORC_CP_LF];
    if (tire == 'f') { l_t = p[ORC_CP_LR]; h_cm *= -1; }
    return (mass * l_t * 9.81 + h_cm * fx) / L;
}

/* src/envs/car_racing.jl:282-344 (state [x,y,psi,Vx,Vy,psi_dot,delta,pedal]) */
void orc_car_step(const double *p, double *s, const double *a) {
    double x = s[0], y = s[1], psi = s[2], Vx = s[3], Vy = s[4], psid = s[5], delta = s[6];
    const double dt = p[ORC_CP_DT], ddt = p[ORC_CP_DDT];
    const double l_f = p[ORC_CP_LF], l_r = p[ORC_CP_LR];

    double tgt = a[0] * p[ORC_CP_DMAX] - delta;
    double commanded = fabs(tgt) / dt;                                             /* :295 */
    double rate = jl_min(commanded, p[ORC_CP_DDOTMAX]) * jl_sign(tgt);             /* :296 */
    double pedal = a[1];                                                           /* :297 */

    int steps = (int)nearbyint(dt / ddt);                                          /* :299 round(Int,..) */
    for (int it = 0; it < steps; ++it) {
        delta += rate * ddt;                                                       /* :301 */
        double alpha_f = atan2(Vy + l_f * psid, Vx) - delta;                       /* :304 */
        double alpha_r = atan2(Vy - l_r * psid, Vx);                               /* :305 */
        double fx_aero = (p[ORC_CP_CD0] + p[ORC_CP_CD1] * fabs(Vx)) * jl_sign(Vx); /* :308 */
        double accel = p[ORC_CP_FXMAX] * jl_max(pedal, 0.0);                       /* :310 */
        double brake = p[ORC_CP_FXMIN] * jl_min(pedal, 0.0) * jl_sign(Vx);         /* :311 */
        double fx = accel + brake;
        double lam = (pedal <= 0) ? p[ORC_CP_LBRAKE] : p[ORC_CP_LDRIVE];
        double fxf = lam * fx;                                                     /* :315 */
        double fxr = (1 - lam) * fx;                                               /* :316 */
        double fzf = orc_calc_tire_fz(p, fx, 'f');
        double fzr = orc_calc_tire_fz(p, fx, 'r');
        double fyf = orc_calc_tire_fy(alpha_f, p[ORC_CP_MUF], p[ORC_CP_CAF], fzf, fxf);
        double fyr = orc_calc_tire_fy(alpha_r, p[ORC_CP_MUR], p[ORC_CP_CAR], fzr, fxr);

        double sd = sin(delta), cd = cos(delta);
        double psidd = (1 / p[ORC_CP_IZZ]) * (l_f * (fxf * sd + fyf * cd) - l_r * fyr);      /* :322 */
        double Vyd = (1 / p[ORC_CP_M]) * (fyf * cd + fxf * sd + fyr) - psid * Vx;           /* :323 */
        double Vxd = (1 / p[ORC_CP_M]) * (fxf * cd - fyf * sd + fxr - fx_aero) + psid * Vy; /* :324 */

        psid += psidd * ddt;                                                       /* :326 */
        Vx += Vxd * ddt;
        Vy += Vyd * ddt;
        psi += psid * ddt;                                                         /* :329 */
        psi = atan2(sin(psi), cos(psi));                                           /* :330 */
        x += (Vx * cos(psi) - Vy * sin(psi)) * ddt;                                /* :331 */
        y += (Vx * sin(psi) + Vy * cos(psi)) * ddt;                                /* :332 */
    }
    s[0] = x; s[1] = y; s[2] = psi; s[3] = Vx; s[4] = Vy; s[5] = psid; s[6] = delta; s[7] = pedal;
}

/* src/envs/car_racing_tracks/car_racing_tracks.jl:68-92 */
int orc_within_track(int P, const double *tx, const double *ty, const double *tw,
                     const double *pos, double *dist_out) {
    int min_idx = 1; double best = 0.0;
    for (int i = 1; i <= P; ++i) {                                                 /* :71-73 findmin: first min */
        double dx = tx[i - 1] - pos[0], dy = ty[i - 1] - pos[1];
        double d = dx * dx + dy * dy;
        if (i == 1 || d < best) { best = d; min_idx = i; }
    }
    int m1 = mod1(min_idx - 1, P), p1i = mod1(min_idx + 1, P);                     /* :75-76 */
    double ax = tx[m1 - 1] - pos[0], ay = ty[m1 - 1] - pos[1];
    double bx = tx[p1i - 1] - pos[0], by = ty[p1i - 1] - pos[1];
    double dist_m1 = sqrt(ax * ax + ay * ay);                                      /* :77 */
    double dist_p1 = sqrt(bx * bx + by * by);                                      /* :78 */
    int idx2 = (dist_m1 <= dist_p1) ? m1 : p1i;                                    /* :79 */
    double p1x = tx[min_idx - 1], p1y = ty[min_idx - 1];
    double p2x = tx[idx2 - 1], p2y = ty[idx2 - 1];
    double ux = pos[0] - p1x, uy = pos[1] - p1y;
    double vx = p2x - p1x, vy = p2y - p1y;
    double t = (ux * vx + uy * vy) / (vx * vx + vy * vy);                          /* :87 */
    double qx = p1x + t * vx, qy = p1y + t * vy;                                   /* :88 */
    double ex = qx - pos[0], ey = qy - pos[1];
    double dist = sqrt(ex * ex + ey * ey);                                         /* :89 */
    *dist_out = dist;
    return dist < tw[min_idx - 1];                                                 /* :90 */
}

double orc_calculate_beta(const double *s) { return atan2(s[4], s[3]); }           /* car_racing.jl:181-183 */

/* src/envs/car_racing.jl:201-213 */
double orc_car_reward(const double *p, int P, const double *tx, const double *ty, const double *tw,
                      const double *s) {
    double rew = 0.0, dist;
    int within = orc_within_track(P, tx, ty, tw, s, &dist);
    if (!within) rew += -1000000.0;
    if (fabs(orc_calculate_beta(s)) > p[ORC_CP_BETALIM]) rew += -5000.0;           /* :184-189,207 */
    rew += -dist;
    rew += 2.0 * sqrt(s[3] * s[3] + s[4] * s[4]);
    return rew;
}

/* ======================================================================================
 * MountainCar [3P: ReinforcementLearning.jl 0.11 -> ReinforcementLearningEnvironments,
 * src/environments/examples/MountainCarEnv.jl: MountainCarEnvParams(continuous = true), _step!(env, force)],
 * functor + reward override: src/examples/mountaincar_example.jl:4-22
 * ====================================================================================== */
void orc_mountaincar_default_params(double *p) {
    p[ORC_MP_MINPOS] = -1.2; p[ORC_MP_MAXPOS] = 0.6; p[ORC_MP_MAXSPEED] = 0.07;
    p[ORC_MP_GOALPOS] = 0.45; p[ORC_MP_GOALVEL] = 0.0; p[ORC_MP_POWER] = 0.0015;
    p[ORC_MP_GRAVITY] = 0.0025; p[ORC_MP_MAXSTEPS] = 200.0;
}

static void mc_step(const double *p, double *s, int *t, int *done, double force) {
    *t += 1;
    double x = s[0], v = s[1];
    v += force * p[ORC_MP_POWER] + cos(3 * x) * (-p[ORC_MP_GRAVITY]);
    v = jl_clamp(v, -p[ORC_MP_MAXSPEED], p[ORC_MP_MAXSPEED]);
    x += v;
    x = jl_clamp(x, p[ORC_MP_MINPOS], p[ORC_MP_MAXPOS]);
    if (x == p[ORC_MP_MINPOS] && v < 0) v = 0;
    *done = (x >= p[ORC_MP_GOALPOS] && v >= p[ORC_MP_GOALVEL]) || (*t >= (int)p[ORC_MP_MAXSTEPS]);
    s[0] = x; s[1] = v;
}

/* src/examples/mountaincar_example.jl:10-22 */
static double mc_reward(const double *p, const double *s, int done) {
    double rew = 0.0;
    if (s[0] >= p[ORC_MP_GOALPOS] && s[1] >= p[ORC_MP_GOALVEL]) rew += 100000;
    rew += fabs(s[1]);
    rew += done ? 0.0 : -1.0;
    return rew;
}

/* ======================================================================================
 * CartPole [3P: ReinforcementLearning.jl 0.11 -> ReinforcementLearningEnvironments,
 * src/environments/examples/CartPoleEnv.jl: CartPoleEnvParams, _step!(env, a); recalled, unpinned], functor: src/examples/cartpole_example.jl:3-6; reward = RL.jl's (done ? 0 : 1)
 * ====================================================================================== */
void orc_cartpole_default_params(double *p) {
    p[ORC_XP_GRAVITY] = 9.8; p[ORC_XP_MASSCART] = 1.0; p[ORC_XP_MASSPOLE] = 0.1;
    p[ORC_XP_TOTALMASS] = p[ORC_XP_MASSPOLE] + p[ORC_XP_MASSCART];
    p[ORC_XP_HALFLENGTH] = 0.5; p[ORC_XP_POLEMASSLENGTH] = p[ORC_XP_MASSPOLE] * p[ORC_XP_HALFLENGTH];
    p[ORC_XP_FORCEMAG] = 10.0; p[ORC_XP_DT] = 0.02;
    p[ORC_XP_THETATHR] = 12 * 2 * M_PI / 360; p[ORC_XP_XTHR] = 2.4; p[ORC_XP_MAXSTEPS] = 200.0;
}

static void cp_step(const double *p, double *s, int *t, int *done, double a) {
    *t += 1;
    double force = a * p[ORC_XP_FORCEMAG];
    double xdot = s[1], theta = s[2], thetadot = s[3];
    double costheta = cos(theta), sintheta = sin(theta);
    double tmp = (force + p[ORC_XP_POLEMASSLENGTH] * (thetadot * thetadot) * sintheta) / p[ORC_XP_TOTALMASS];
    double thetaacc = (p[ORC_XP_GRAVITY] * sintheta - costheta * tmp) /
                      (p[ORC_XP_HALFLENGTH] * (4.0 / 3.0 - p[ORC_XP_MASSPOLE] * (costheta * costheta) / p[ORC_XP_TOTALMASS]));
    double xacc = tmp - p[ORC_XP_POLEMASSLENGTH] * thetaacc * costheta / p[ORC_XP_TOTALMASS];
    s[0] += p[ORC_XP_DT] * xdot;
    s[1] += p[ORC_XP_DT] * xacc;
    s[2] += p[ORC_XP_DT] * thetadot;
    s[3] += p[ORC_XP_DT] * thetaacc;
    *done = fabs(s[0]) > p[ORC_XP_XTHR] || fabs(s[2]) > p[ORC_XP_THETATHR] || *t > (int)p[ORC_XP_MAXSTEPS];
}

/* ======================================================================================
 * env protocol
 * ====================================================================================== */
void orc_env_init(orc_env *e, int kind, int ncars, const double *params,
                  int P, const double *tx, const double *ty, const double *tw) {
    memset(e, 0, sizeof(*e));
    e->kind = kind; e->ncars = ncars;
    if (kind == ORC_ENV_CAR) {
        e->ss = 8 * ncars; e->as = 2 * ncars;
        if (params) memcpy(e->params, params, sizeof(double) * ORC_CP_N); else orc_car_default_params(e->params);
    } else if (kind == ORC_ENV_CARTPOLE) {
        e->ss = 4; e->as = 1; e->ncars = 0;
        if (params) memcpy(e->params, params, sizeof(double) * ORC_XP_N); else orc_cartpole_default_params(e->params);
    } else {
        e->ss = 2; e->as = 1; e->ncars = 0;
        if (params) memcpy(e->params, params, sizeof(double) * ORC_MP_N); else orc_mountaincar_default_params(e->params);
    }
    e->P = P; e->tx = tx; e->ty = ty; e->tw = tw;
    orc_env_reset(e);
}

/* car_racing.jl:215-223; multi-car_racing.jl:160-180.  MountainCar's reset draws x~U(-0.6,-0.4)
 * from an unseeded RNG in the reference -- here: midpoint; callers set state. */
void orc_env_reset(orc_env *e) {
    memset(e->state, 0, sizeof(e->state));
    e->t = 0; e->done = 0;
    if (e->kind == ORC_ENV_CAR) {
        for (int c = 0; c < e->ncars; ++c) {
            double *s = e->state + 8 * c;
            int ii = c + 1;
            if (ii >= 2) s[0] = (ii % 2 == 0) ? (ii / 2.0 * 5.0) : ((1 - ii) / 2.0 * 5.0);
            s[2] = 90.0 * (M_PI / 180.0);
            s[3] = 10.0;
        }
    } else if (e->kind == ORC_ENV_MOUNTAINCAR) {
        e->state[0] = -0.5; e->state[1] = 0.0;
    }   /* CartPole: reference draws 0.1*rand(4)-0.05; here the centre (zeros); callers set state */
}

void orc_action_bounds(const orc_env *e, double *lo, double *hi) {
    for (int i = 0; i < e->as; ++i) { lo[i] = -1.0; hi[i] = 1.0; }  /* car_racing.jl:156-159; multi :75-84; RL.jl -1..1 */
}

int orc_env_step(orc_env *e, const double *a) {
    for (int i = 0; i < e->as; ++i)
        if (!(a[i] >= -1.0 && a[i] <= 1.0)) {
            /* car_racing.jl:239 "Action is not in action space"; the multi-car functor calls _step!
             * directly (:204) and does not check -- but NaN would poison everything; flag it. */
            if (e->kind != ORC_ENV_CAR || e->ncars == 1) return -3;
        }
    if (e->kind == ORC_ENV_CAR) {
        for (int c = 0; c < e->ncars; ++c) orc_car_step(e->params, e->state + 8 * c, a + 2 * c); /* multi :200-207 */
        e->t += 1;
    } else if (e->kind == ORC_ENV_CARTPOLE) {
        cp_step(e->params, e->state, &e->t, &e->done, a[0]);
    } else {
        mc_step(e->params, e->state, &e->t, &e->done, a[0]);
    }
    return 0;
}

double orc_env_reward(const orc_env *e) {
    if (e->kind == ORC_ENV_CARTPOLE) return e->done ? 0.0 : 1.0;
    if (e->kind != ORC_ENV_CAR) return mc_reward(e->params, e->state, e->done);
    if (e->ncars == 1) return orc_car_reward(e->params, e->P, e->tx, e->ty, e->tw, e->state);
    double rew = 0.0;                                                              /* multi-car_racing.jl:145-158 */
    for (int i = 0; i < e->ncars; ++i) {
        const double *si = e->state + 8 * i;
        rew += orc_car_reward(e->params, e->P, e->tx, e->ty, e->tw, si);
        for (int j = i + 1; j < e->ncars; ++j) {
            const double *sj = e->state + 8 * j;
            double dx = sj[0] - si[0], dy = sj[1] - si[1];
            double dd = sqrt(dx * dx + dy * dy);
            rew += -dd;
            if (dd <= 4.0) rew += -11000.0;
        }
    }
    return rew;
}

/* ======================================================================================
 * src/utils.jl
 * ====================================================================================== */
/* :9-21 (Vector variant = diagonal; Matrix variant = dense block diagonal) */
void orc_block_diagm(const double *A, int r, int rep, double *B) {
    int n = r * rep;
    memset(B, 0, sizeof(double) * n * n);
    for (int b = 0; b < rep; ++b)
        for (int j = 0; j < r; ++j)
            for (int i = 0; i < r; ++i)
                B[(b * r + i) + (size_t)(b * r + j) * n] = A[i + j * r];
}

/* :55-67 reshape(V, as, T) and clamp row r to [lo_r, hi_r] (in place, V aliased) */
void orc_get_model_controls(const double *lo, const double *hi, int as, double *V, int T) {
    for (int t = 0; t < T; ++t)
        for (int r = 0; r < as; ++r)
            V[t * as + r] = jl_clamp(V[t * as + r], lo[r], hi[r]);
}

/* :79-86 */
void orc_compute_weights(double lambda, const double *cost, int K, double *w) {
    double rho = cost[0];
    for (int k = 1; k < K; ++k) rho = jl_min(rho, cost[k]);
    double eta = 0.0;
    for (int k = 0; k < K; ++k) { w[k] = exp(-1 / lambda * (cost[k] - rho)); eta += w[k]; }
    for (int k = 0; k < K; ++k) w[k] = w[k] / eta;
}

/* :129-144 */
double orc_rollout_model(orc_env *e, int T, const double *controls, double *traj_log) {
    double traj_cost = 0.0;
    int thrown = 0;
    for (int t = 0; t < T; ++t) {
        /* env(controls) throws "Action is not in action space" for a NaN control (car_racing.jl:239; clamp passes NaN through): the whole pol(env)
         * call dies there.  A restated rollout cannot throw out of an OpenMP loop, so the rollout's cost is poisoned instead and the caller turns a
         * non-finite cost into status -3 (round 6: the return code used to be dropped here, a NaN control gave status 0 and a finite cost). */
        if (orc_env_step(e, controls + (size_t)t * e->as)) thrown = 1;
        traj_cost -= orc_env_reward(e);
        if (traj_log) memcpy(traj_log + (size_t)t * e->ss, e->state, sizeof(double) * e->ss);
    }
    return thrown ? NAN : traj_cost;
}

int orc_m_elite(int K, double thr) { return (int)nearbyint(K * (1 - thr)); }       /* policies.jl:437,515 */

/* ======================================================================================
 * dense linear algebra [3P stand-ins for the LAPACK calls behind: LinearAlgebra stdlib cholesky.jl
 * (cholesky(::Symmetric) -> potrf, PosDefException on a non-positive pivot), PDMats 0.11 src/pdmat.jl
 * (PDMat(Σ) inside Distributions.MvNormal(Σ); inv(::PDMat) for invcov), symmetric.jl (^, eigen)]
 * ====================================================================================== */
int orc_cholesky_lower(int n, const double *A, double *L) {
    memset(L, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; ++j) {
        double d = A[j + (size_t)j * n];
        for (int k = 0; k < j; ++k) d -= L[j + (size_t)k * n] * L[j + (size_t)k * n];
        if (!(d > 0.0)) return -2;
        double ljj = sqrt(d);
        L[j + (size_t)j * n] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double v = A[i + (size_t)j * n];
            for (int k = 0; k < j; ++k) v -= L[i + (size_t)k * n] * L[j + (size_t)k * n];
            L[i + (size_t)j * n] = v / ljj;
        }
    }
    return 0;
}

void orc_inv_from_chol(int n, const double *L, double *Ainv) {
    /* Linv (lower) then Ainv = Linv' * Linv */
    double *Li = (double *)calloc((size_t)n * n, sizeof(double));
    for (int j = 0; j < n; ++j) {
        Li[j + (size_t)j * n] = 1.0 / L[j + (size_t)j * n];
        for (int i = j + 1; i < n; ++i) {
            double v = 0.0;
            for (int k = j; k < i; ++k) v -= L[i + (size_t)k * n] * Li[k + (size_t)j * n];
            Li[i + (size_t)j * n] = v / L[i + (size_t)i * n];
        }
    }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            double v = 0.0;
            int k0 = i > j ? i : j;
            for (int k = k0; k < n; ++k) v += Li[k + (size_t)i * n] * Li[k + (size_t)j * n];
            Ainv[i + (size_t)j * n] = v;
        }
    free(Li);
}

/* cyclic Jacobi for symmetric A; V columns = eigenvectors */
int orc_sym_eig(int n, const double *A, double *ev, double *V) {
    double *M = (double *)malloc(sizeof(double) * n * n);
    memcpy(M, A, sizeof(double) * n * n);
    for (int i = 0; i < n * n; ++i) V[i] = 0.0;
    for (int i = 0; i < n; ++i) V[i + (size_t)i * n] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
            double v = M[i + (size_t)j * n];
            if (i == j) diag += v * v; else off += v * v;
        }
        if (off <= 1e-30 * (diag > 0 ? diag : 1.0)) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = M[p + (size_t)q * n];
                if (apq == 0.0) continue;
                double app = M[p + (size_t)p * n], aqq = M[q + (size_t)q * n];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {   /* columns p,q */
                    double kp = M[k + (size_t)p * n], kq = M[k + (size_t)q * n];
                    M[k + (size_t)p * n] = c * kp - s * kq;
                    M[k + (size_t)q * n] = s * kp + c * kq;
                }
                for (int k = 0; k < n; ++k) {   /* rows p,q */
                    double pk = M[p + (size_t)k * n], qk = M[q + (size_t)k * n];
                    M[p + (size_t)k * n] = c * pk - s * qk;
                    M[q + (size_t)k * n] = s * pk + c * qk;
                }
                for (int k = 0; k < n; ++k) {
                    double kp = V[k + (size_t)p * n], kq = V[k + (size_t)q * n];
                    V[k + (size_t)p * n] = c * kp - s * kq;
                    V[k + (size_t)q * n] = s * kp + c * kq;
                }
            }
    }
    for (int i = 0; i < n; ++i) ev[i] = M[i + (size_t)i * n];
    free(M);
    return 0;
}

/* [3P] LinearAlgebra stdlib, symmetric.jl: ^(A::Symmetric{<:Real}, p::Real): F = eigen(A); all λ >= 0 ->
 * Symmetric((F.vectors * Diagonal(F.values .^ p)) * F.vectors')   (policies.jl:580 calls it as Σ^-0.5) */
int orc_sym_pow(int n, const double *A, double p, double *out) {
    double *ev = (double *)malloc(sizeof(double) * n);
    double *V = (double *)malloc(sizeof(double) * n * n);
    orc_sym_eig(n, A, ev, V);
    int bad = 0;
    for (int i = 0; i < n; ++i) { if (!(ev[i] > 0.0)) bad = 1; ev[i] = pow(ev[i], p); }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            double v = 0.0;
            for (int k = 0; k < n; ++k) v += V[i + (size_t)k * n] * ev[k] * V[j + (size_t)k * n];
            out[i + (size_t)j * n] = v;
        }
    free(ev); free(V);
    return bad ? -2 : 0;
}

/* ======================================================================================
 * Philox4x32-10 (Salmon et al., SC'11) + Box-Muller.  The reference uses Julia's
 * MersenneTwister + ziggurat randn!, which is not reproduced: parity is defined
 * on injected noise; this generator is the shared synthetic-noise source of oracle and engine.
 * ====================================================================================== */
void orc_philox4x32_10(const uint32_t c_in[4], const uint32_t k_in[2], uint32_t out[4]) {
    uint32_t c0 = c_in[0], c1 = c_in[1], c2 = c_in[2], c3 = c_in[3];
    uint32_t k0 = k_in[0], k1 = k_in[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline void philox_pair(uint64_t seed, uint32_t slo, uint32_t shi, uint64_t j, uint64_t *a, uint64_t *b) {
    uint32_t ctr[4] = { (uint32_t)j, (uint32_t)(j >> 32), slo, shi };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t r[4];
    orc_philox4x32_10(ctr, key, r);
    *a = ((uint64_t)r[1] << 32) | r[0];
    *b = ((uint64_t)r[3] << 32) | r[2];
}

/* The normal stream (the engine's definition, mpopis_amd/csrc/philox.h): call q (counter = q, stream words, key = seed) yields normals
 * 4q .. 4q+3 -- Box-Muller on (w0, w1) and on (w2, w3), radius word first, uniforms u = (w + 0.5) 2^-32. */
static inline void box_muller_u32(uint32_t wr, uint32_t wa, double *z0, double *z1) {
    const double two_m32 = 1.0 / 4294967296.0;
    double u1 = ((double)wr + 0.5) * two_m32, u2 = ((double)wa + 0.5) * two_m32;
    double R = sqrt(-2.0 * log(u1));
    double th = 6.283185307179586476925286766559 * u2;
    *z0 = R * cos(th); *z1 = R * sin(th);
}
void orc_philox_normals(uint64_t seed, uint32_t slo, uint32_t shi, int64_t n, double *out) {
    for (int64_t q = 0; 4 * q < n; ++q) {
        uint32_t ctr[4] = { (uint32_t)q, (uint32_t)((uint64_t)q >> 32), slo, shi };
        uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
        uint32_t r[4];
        double z[4];
        orc_philox4x32_10(ctr, key, r);
        box_muller_u32(r[0], r[1], &z[0], &z[1]);
        box_muller_u32(r[2], r[3], &z[2], &z[3]);
        for (int i = 0; i < 4 && 4 * q + i < n; ++i) out[4 * q + i] = z[i];
    }
}

void orc_philox_resample_draws(uint64_t seed, uint32_t slo, uint32_t shi, int K, int32_t *idx0, double *u) {
    const double two_m53 = 1.0 / 9007199254740992.0;
    for (int k = 0; k < K; ++k) {
        uint64_t a, b;
        philox_pair(seed, slo, shi, (uint64_t)k, &a, &b);
        double ua = (double)(a >> 11) * two_m53;
        int32_t i = (int32_t)(ua * K);
        if (i >= K) i = K - 1;
        idx0[k] = i;
        u[k] = (double)(b >> 11) * two_m53;
    }
}

/* ======================================================================================
 * alias table [3P]: StatsBase 0.34 src/sampling.jl make_alias_table!(w, wsum, a, alias) as used by
 * Distributions 0.25 src/samplers/aliastable.jl AliasTable(probs) -- the sampler of Categorical(ws)
 * (policies.jl:804-805); rand(rng, s::AliasTable): i = rand(rng, 1:n); rand(rng) < s.accept[i] ? i : s.alias[i]
 * ====================================================================================== */
void orc_make_alias_table(const double *w, double wsum, int n, double *a, int32_t *alias) {
    double ac = n / wsum;
    for (int i = 0; i < n; ++i) { a[i] = w[i] * ac; alias[i] = i; }
    int32_t *larges = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *smalls = (int32_t *)malloc(sizeof(int32_t) * n);
    int kl = 0, ks = 0;
    for (int i = 0; i < n; ++i) {
        double ai = a[i];
        if (ai > 1.0) larges[kl++] = i;
        else if (ai < 1.0) smalls[ks++] = i;
    }
    while (kl > 0 && ks > 0) {
        int s = smalls[--ks];
        int l = larges[--kl];
        alias[s] = l;
        double al = a[l] = (a[l] - 1.0) + a[s];
        if (al > 1.0) larges[kl++] = l; else smalls[ks++] = l;
    }
    for (int i = 0; i < ks; ++i) a[smalls[i]] = 1.0;
    free(larges); free(smalls);
}

/* Distributions rand(rng, s::AliasTable): i = rand(1:n); u = rand(); u < accept[i] ? i : alias[i] */
void orc_alias_sample(const double *accept, const int32_t *alias, int n,
                      const int32_t *di, const double *du, int m, int32_t *out) {
    (void)n;
    for (int k = 0; k < m; ++k) { int i = di[k]; out[k] = (du[k] < accept[i]) ? i : alias[i]; }
}

/* ======================================================================================
 * policies: src/mppi_mpopi_policies.jl
 * ====================================================================================== */
/* MPPI_Policy_Params :36-102 + per-policy constructors */
int orc_policy_create(orc_policy *pol, int kind, const orc_env *env, int K, int T,
                      double lambda, double alpha, const double *U0, int nU0,
                      const double *cov, int ncov, int cov_is_vector,
                      int N, double lambda_ais, double elite_threshold, int sigma_est,
                      double cma_sigma) {
    memset(pol, 0, sizeof(*pol));
    pol->kind = kind; pol->K = K; pol->T = T; pol->ss = env->ss; pol->as = env->as;
    pol->cs = env->as * T;                                                         /* :58-59 */
    pol->lambda = lambda; pol->alpha = alpha; pol->N = N; pol->lambda_ais = lambda_ais;
    pol->elite_threshold = elite_threshold; pol->sigma_est = sigma_est; pol->cma_sigma = cma_sigma;
    pol->nthreads = 1;
    int as = pol->as, cs = pol->cs;
    pol->U = (double *)malloc(sizeof(double) * cs);
    if (nU0 == as) { for (int t = 0; t < T; ++t) memcpy(pol->U + t * as, U0, sizeof(double) * as); } /* :61-63 */
    else if (nU0 == cs) memcpy(pol->U, U0, sizeof(double) * cs);
    else return -1;                                                                /* :64 */
    int repeat_num = (kind == ORC_POL_MPPI) ? 1 : T;                               /* :66-74 */
    int check = (kind == ORC_POL_MPPI) ? as : cs;
    int n = check;
    pol->Sigma = (double *)calloc((size_t)n * n, sizeof(double));
    double *full = NULL; int nfull = ncov;
    /* expand a Vector cov to a diagonal matrix (block_diagm Vector variant, utils.jl:9-11) */
    double *covm = (double *)calloc((size_t)ncov * ncov, sizeof(double));
    if (cov_is_vector) for (int i = 0; i < ncov; ++i) covm[i + (size_t)i * ncov] = cov[i];
    else memcpy(covm, cov, sizeof(double) * ncov * ncov);
    if (ncov == as) {                                                              /* :76-78 */
        if (cov_is_vector || 1) {
            full = (double *)malloc(sizeof(double) * (size_t)(as * repeat_num) * (as * repeat_num));
            orc_block_diagm(covm, as, repeat_num, full);
            nfull = as * repeat_num;
        }
    } else { full = covm; covm = NULL; }
    if (nfull != check) { free(full); free(covm); return -1; }                     /* :79 */
    memcpy(pol->Sigma, full, sizeof(double) * (size_t)n * n);
    free(full); free(covm);
    orc_action_bounds(env, pol->lo, pol->hi);

    if (kind == ORC_POL_CMAMPPI) {                                                 /* :513-525 */
        int m = K; double nn = (double)cs;
        pol->m_elite = (int)nearbyint((1.0 - elite_threshold) * m);
        pol->ws = (double *)malloc(sizeof(double) * m);
        for (int i = 1; i <= m; ++i) pol->ws[i - 1] = log((m + 1) / 2.0) - log((double)i);
        double s = 0.0;
        for (int i = 0; i < pol->m_elite; ++i) s += pol->ws[i];
        for (int i = 0; i < pol->m_elite; ++i) pol->ws[i] /= s;
        double s2 = 0.0;
        for (int i = 0; i < pol->m_elite; ++i) s2 += pol->ws[i] * pol->ws[i];
        double mu_eff = 1 / s2;
        pol->mu_eff = mu_eff;
        pol->c_sigma = (mu_eff + 2) / (nn + mu_eff + 5);
        pol->d_sigma = 1 + 2 * jl_max(0, sqrt((mu_eff - 1) / (nn + 1)) - 1) + pol->c_sigma;
        pol->c_Sigma = (4 + mu_eff / nn) / (nn + 4 + 2 * mu_eff / nn);
        pol->c1 = 2 / ((nn + 1.3) * (nn + 1.3) + mu_eff);
        pol->c_mu = jl_min(1 - pol->c1, 2 * (mu_eff - 2 + 1 / mu_eff) / ((nn + 2) * (nn + 2) + mu_eff));
        double st = 0.0;
        for (int i = pol->m_elite; i < m; ++i) st += pol->ws[i];
        double f = -(1 + pol->c1 / pol->c_mu) / st;
        for (int i = pol->m_elite; i < m; ++i) pol->ws[i] *= f;
        pol->E_cma = sqrt(nn) * (1 - 1 / (4 * nn) + 1 / (21 * (nn * nn)));
    } else if (kind == ORC_POL_CEMPPI) {
        pol->m_elite = orc_m_elite(K, elite_threshold);
    }
    return 0;
}

void orc_policy_free(orc_policy *pol) { free(pol->U); free(pol->Sigma); free(pol->ws); memset(pol, 0, sizeof(*pol)); }

/* simulate_model(pol::AbstractGMPPI_Policy, env::AbstractEnv, E, Sigma_inv, U_orig) :261-278 */
void orc_simulate_model(const orc_policy *pol, const double *Ucur, const orc_env *env,
                        const double *E, const double *Sigma_inv, const double *U_orig,
                        double *cost, double *traj_log) {
    const int K = pol->K, T = pol->T, cs = pol->cs, as = pol->as;
    const double gamma = pol->lambda * (1 - pol->alpha);
    double *row = NULL;
    if (gamma != 0.0 && Sigma_inv) {           /* (gamma*U_orig')*Sigma_inv is k-independent */
        row = (double *)malloc(sizeof(double) * cs);
        for (int j = 0; j < cs; ++j) {
            double v = 0.0;
            for (int i = 0; i < cs; ++i) v += (gamma * U_orig[i]) * Sigma_inv[i + (size_t)j * cs];
            row[j] = v;
        }
    }
#ifdef _OPENMP
#pragma omp parallel for num_threads(pol->nthreads > 0 ? pol->nthreads : 1) schedule(static)
#endif
    for (int k = 0; k < K; ++k) {                                                  /* :269 */
        orc_env sim = *env;                                                        /* :270 copy(env) */
        double V[cs];
        for (int r = 0; r < cs; ++r) V[r] = Ucur[r] + E[r + (size_t)k * cs];       /* :271 */
        double control_cost = 0.0;
        if (row) for (int r = 0; r < cs; ++r) control_cost += row[r] * (V[r] - U_orig[r]); /* :272 (unclamped V) */
        orc_get_model_controls(pol->lo, pol->hi, as, V, T);                        /* :273 */
        double c = orc_rollout_model(&sim, T, V, traj_log ? traj_log + (size_t)k * T * env->ss : NULL); /* :274 */
        cost[k] = c + control_cost;                                                /* :275 */
    }
    free(row);
}

/* get_controls_roll_U! : src/utils.jl:88-101, with pol.U === pol.params.U0 */
void orc_roll_U(orc_policy *pol, const double *wc, double *control) {
    const int as = pol->as, cs = pol->cs;
    for (int i = 0; i < as; ++i) control[i] = jl_clamp(wc[i], pol->lo[i], pol->hi[i]);   /* :91 */
    if (pol->T > 1) {
        for (int i = 0; i < cs - as; ++i) pol->U[i] = wc[i + as];                  /* :95 */
        /* :96 pol.U[(end-as):end] = pol.params.U0[(end-as):end] -- same array: no-op */
    } else {
        memcpy(pol->U, wc, sizeof(double) * cs);                                   /* :98 */
    }
}

/* Threads for the dense helpers below.  The reference gets these from multi-threaded BLAS/LAPACK (OpenBLAS under
 * Julia); the naive loops here are at least spread over the same cores as the rollouts so the timed CPU baseline is
 * not dominated by serial linear algebra. */
static int g_dense_threads = 1;

/* E = L*Z, Z and E cs x K col-major.  [3P] Distributions 0.25 src/multivariate/mvnormal.jl: _rand!(rng, d::MvNormal, x) =
 * unwhiten!(d.Σ, randn!(rng, x)) (+ μ = 0); PDMats 0.11 src/pdmat.jl: unwhiten!(r, a::PDMat, x) = lmul!(chol_lower(a.chol), r) */
static void lmul_LZ(int cs, int K, const double *L, const double *Z, double *E) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_dense_threads) schedule(static)
#endif
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < cs; ++i) {
            double v = 0.0;
            for (int j = 0; j <= i; ++j) v += L[i + (size_t)j * cs] * Z[j + (size_t)k * cs];
            E[i + (size_t)k * cs] = v;
        }
}

typedef struct { double c; int i; } cost_idx;
static int cmp_cost_idx(const void *a, const void *b) {
    const cost_idx *x = (const cost_idx *)a, *y = (const cost_idx *)b;
    if (x->c < y->c) return -1;
    if (x->c > y->c) return 1;
    return (x->i > y->i) - (x->i < y->i);     /* stable: ties by index (Julia sortperm default) */
}
static void sortperm(const double *c, int K, int *order) {
    cost_idx *t = (cost_idx *)malloc(sizeof(cost_idx) * K);
    for (int k = 0; k < K; ++k) { t[k].c = c[k]; t[k].i = k; }
    qsort(t, K, sizeof(cost_idx), cmp_cost_idx);
    for (int k = 0; k < K; ++k) order[k] = t[k].i;
    free(t);
}

/* [3P] CovarianceEstimation 0.2 src/basicmethods.jl: cov(::SimpleCovariance, X; dims = 1) with X = elite' (m x cs):
 * corrected = false by default, i.e. divide by m */
static void cov_mle_cols(int cs, int m, const double *X /* cs x m col-major */, double *mean, double *S) {
    for (int r = 0; r < cs; ++r) {
        double s = 0.0;
        for (int j = 0; j < m; ++j) s += X[r + (size_t)j * cs];
        mean[r] = s / m;
    }
    for (int b = 0; b < cs; ++b)
        for (int a = 0; a <= b; ++a) {
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += (X[a + (size_t)j * cs] - mean[a]) * (X[b + (size_t)j * cs] - mean[b]);
            S[a + (size_t)b * cs] = S[b + (size_t)a * cs] = s / m;
        }
}

/* [3P] CovarianceEstimation 0.2 src/linearshrinkage.jl: cov(LinearShrinkage(DiagonalUnequalVariance(), :ss), X) --
 * linear_shrinkage(::DiagonalUnequalVariance, Xc, S, :ss, ...) [recalled from Schaefer & Strimmer 2005,
 * target D: shrink off-diagonals only; lambda* = sum_{i!=j} Var^(r_ij) / sum_{i!=j} r_ij^2 computed on
 * standardised data; S_shrunk = lambda*diag(S) + (1-lambda)*S].  UNPINNED. */
static void cov_ss_cols(int cs, int m, const double *X, double *mean, double *S) {
    cov_mle_cols(cs, m, X, mean, S);          /* uncorrected S (corrected=false default) */
    double *sd = (double *)malloc(sizeof(double) * cs);
    for (int a = 0; a < cs; ++a) sd[a] = sqrt(S[a + (size_t)a * cs]);
    double num = 0.0, den = 0.0;
    for (int b = 0; b < cs; ++b)
        for (int a = 0; a < cs; ++a) {
            if (a == b) continue;
            double rab = S[a + (size_t)b * cs] / (sd[a] * sd[b]);
            double v = 0.0;
            for (int j = 0; j < m; ++j) {
                double wj = ((X[a + (size_t)j * cs] - mean[a]) / sd[a]) * ((X[b + (size_t)j * cs] - mean[b]) / sd[b]);
                v += (wj - rab) * (wj - rab);
            }
            /* Var^(r_ab) = n/(n-1)^3 * sum_k (w_kab - wbar_ab)^2 */
            num += v * ((double)m / ((double)(m - 1) * (m - 1) * (m - 1)));
            den += rab * rab;
        }
    double lam = den > 0 ? num / den : 1.0;
    lam = jl_clamp(lam, 0.0, 1.0);
    for (int b = 0; b < cs; ++b)
        for (int a = 0; a < cs; ++a)
            if (a != b) S[a + (size_t)b * cs] *= (1 - lam);
    free(sd);
}

/* [3P] CovarianceEstimation 0.2 src/linearshrinkage.jl: LinearShrinkage(DiagonalUnequalVariance(), :lw) [recalled: Ledoit & Wolf intensity for the diagonal target on the
 * UNstandardised data: lambda* = sum_{i!=j} Var^(s_ij) / sum_{i!=j} s_ij^2, Var^(s_ij) = n/(n-1)^3 sum_k (w_kij - s_ij)^2,
 * w_kij = xc_ki xc_kj].  UNPINNED. */
static void cov_lw_cols(int cs, int m, const double *X, double *mean, double *S) {
    cov_mle_cols(cs, m, X, mean, S);
    double num = 0.0, den = 0.0;
    for (int b = 0; b < cs; ++b)
        for (int a = 0; a < cs; ++a) {
            if (a == b) continue;
            const double sab = S[a + (size_t)b * cs];
            double v = 0.0;
            for (int j = 0; j < m; ++j) {
                double wj = (X[a + (size_t)j * cs] - mean[a]) * (X[b + (size_t)j * cs] - mean[b]);
                v += (wj - sab) * (wj - sab);
            }
            num += v * ((double)m / ((double)(m - 1) * (m - 1) * (m - 1)));
            den += sab * sab;
        }
    double lam = den > 0 ? num / den : 1.0;
    lam = jl_clamp(lam, 0.0, 1.0);
    for (int b = 0; b < cs; ++b)
        for (int a = 0; a < cs; ++a)
            if (a != b) S[a + (size_t)b * cs] *= (1 - lam);
}

/* [3P] CovarianceEstimation 0.2 src/linearshrinkage.jl: LinearShrinkage(DiagonalCommonVariance(), :rblw / :oas) [Chen, Wiesel, Eldar & Hero 2010, eqs. (17) and (23)]:
 * F = tr(S)/p I;  rblw: lambda = ((n-2)/n tr(S^2) + tr(S)^2) / ((n+2)(tr(S^2) - tr(S)^2/p));
 * oas: lambda = ((1-2/p) tr(S^2) + tr(S)^2) / ((n+1-2/p)(tr(S^2) - tr(S)^2/p)); clamp to [0,1];  S <- (1-lambda) S + lambda F.
 * UNPINNED. */
static void cov_common_cols(int cs, int m, const double *X, double *mean, double *S, int oas) {
    cov_mle_cols(cs, m, X, mean, S);
    double tr = 0.0, tr2 = 0.0;
    for (int a = 0; a < cs; ++a) tr += S[a + (size_t)a * cs];
    for (int b = 0; b < cs; ++b) for (int a = 0; a < cs; ++a) tr2 += S[a + (size_t)b * cs] * S[a + (size_t)b * cs];
    const double p = cs, n = m;
    const double dd = tr2 - tr * tr / p;
    double lam = oas ? ((1 - 2 / p) * tr2 + tr * tr) / ((n + 1 - 2 / p) * dd) : ((n - 2) / n * tr2 + tr * tr) / ((n + 2) * dd);
    lam = (dd > 0) ? jl_clamp(lam, 0.0, 1.0) : 1.0;
    const double f = tr / p;
    for (int b = 0; b < cs; ++b)
        for (int a = 0; a < cs; ++a)
            S[a + (size_t)b * cs] = (1 - lam) * S[a + (size_t)b * cs] + ((a == b) ? lam * f : 0.0);
}

int orc_cov_estimate(int cs, int m, const double *X, int est, double *mean, double *S) {
    if (cs < 1 || m < 2 || !X || !mean || !S) return -1;
    switch (est) {
        case ORC_SIGMA_EST_MLE: cov_mle_cols(cs, m, X, mean, S); return 0;
        case ORC_SIGMA_EST_SS: cov_ss_cols(cs, m, X, mean, S); return 0;
        case ORC_SIGMA_EST_LW: cov_lw_cols(cs, m, X, mean, S); return 0;
        case ORC_SIGMA_EST_RBLW: cov_common_cols(cs, m, X, mean, S, 0); return 0;
        case ORC_SIGMA_EST_OAS: cov_common_cols(cs, m, X, mean, S, 1); return 0;
        default: return -1;
    }
}

/* [3P] StatsBase 0.34 src/cov.jl: mean_and_cov(x::DenseMatrix, w::AbstractWeights, dims = 2; corrected = false):
 * m = mean(x, w, dims = 2); scattermat(x, w, mean = m, dims = 2) / sum(w)   (policies.jl:730-733) */
static void wmean_wcov(int cs, int K, const double *E, const double *w, double *mu, double *S) {
    double wsum = 0.0;
    for (int k = 0; k < K; ++k) wsum += w[k];
    for (int r = 0; r < cs; ++r) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += E[r + (size_t)k * cs] * w[k];
        mu[r] = s / wsum;
    }
    if (!S) return;
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_dense_threads) schedule(dynamic, 1)
#endif
    for (int b = 0; b < cs; ++b)
        for (int a = 0; a <= b; ++a) {
            double s = 0.0;
            for (int k = 0; k < K; ++k)
                s += (E[a + (size_t)k * cs] - mu[a]) * w[k] * (E[b + (size_t)k * cs] - mu[b]);
            S[a + (size_t)b * cs] = S[b + (size_t)a * cs] = s * (1 / wsum);
        }
}

/* (pol::MPPI_Policy)(env) :121-146 with calculate_trajectory_costs :186-216 */
static int mppi_call(orc_policy *pol, const orc_env *env, const orc_noise *nz, orc_step_out *out) {
    const int K = pol->K, T = pol->T, as = pol->as, cs = pol->cs;
    const double gamma = pol->lambda * (1 - pol->alpha);
    double L[as * as], Sinv[as * as];
    if (orc_cholesky_lower(as, pol->Sigma, L)) return out->status = -2;             /* :192 */
    orc_inv_from_chol(as, L, Sinv);                                                /* :194 */
    double *E = out->E;
    for (int t = 0; t < T; ++t)                                                    /* :193 k fastest */
        for (int k = 0; k < K; ++k) {
            const double *z = nz->Z + ((size_t)t * K + k) * as;
            double *e = E + ((size_t)t * K + k) * as;
            for (int i = 0; i < as; ++i) { double v = 0; for (int j = 0; j <= i; ++j) v += L[i + j * as] * z[j]; e[i] = v; }
        }
    int status = 0;
    for (int k = 0; k < K; ++k) {                                                  /* :198 serial */
        orc_env sim = *env;
        double c = 0.0;
        for (int t = 0; t < T; ++t) {
            const double *Ei = E + ((size_t)t * K + k) * as;
            const double *ut = pol->U + t * as;
            double V[16];
            for (int i = 0; i < as; ++i) V[i] = ut[i] + Ei[i];                     /* :203 */
            double control_cost = 0.0;
            if (gamma != 0.0) {                                                    /* :204 */
                for (int j = 0; j < as; ++j) {
                    double rj = 0.0;
                    for (int i = 0; i < as; ++i) rj += (gamma * ut[i]) * Sinv[i + j * as];
                    control_cost += rj * Ei[j];
                }
            }
            orc_get_model_controls(pol->lo, pol->hi, as, V, 1);                    /* :205 */
            if (orc_env_step(&sim, V)) status = -3;                                /* :206 */
            c = c - orc_env_reward(&sim) + control_cost;                           /* :208 */
        }
        out->cost[k] = c;
    }
    orc_compute_weights(pol->lambda, out->cost, K, out->weights);                  /* :127 */
    double wn[cs];
    for (int i = 0; i < cs; ++i) wn[i] = 0.0;
    for (int t = 0; t < T; ++t)                                                    /* :131-136 */
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < as; ++i) wn[t * as + i] += out->weights[k] * E[((size_t)t * K + k) * as + i];
    double wc[cs];
    for (int i = 0; i < cs; ++i) wc[i] = pol->U[i] + wn[i];                        /* :137 */
    orc_roll_U(pol, wc, out->control);                                             /* :138 */
    out->iters_run = 1;
    return out->status = status;
}

/* (pol::AbstractGMPPI_Policy)(env) :221-238 + the per-variant calculate_trajectory_costs */
int orc_policy_call(orc_policy *pol, const orc_env *env, const orc_noise *nz, orc_step_out *out) {
    out->status = 0; out->iters_run = 0;
    if (pol->kind == ORC_POL_MPPI) return mppi_call(pol, env, nz, out);

    const int K = pol->K, cs = pol->cs, kind = pol->kind;
    const int N = (kind == ORC_POL_GMPPI) ? 1 : pol->N;
    g_dense_threads = pol->nthreads > 0 ? (pol->nthreads > 32 ? 32 : pol->nthreads) : 1;   /* small loops: more threads only add fork/join cost */
    const double gamma = pol->lambda * (1 - pol->alpha);
    const size_t nn = (size_t)cs * cs;
    double *U_orig = pol->U;                                  /* U_orig = pol.U (same array) */
    double *Ucur = (double *)malloc(sizeof(double) * cs);     /* pol.U rebinding inside the loop */
    memcpy(Ucur, U_orig, sizeof(double) * cs);
    double *Sig = (double *)malloc(sizeof(double) * nn);      /* Sigma' */
    memcpy(Sig, pol->Sigma, sizeof(double) * nn);
    double *L = (double *)malloc(sizeof(double) * nn);
    double *Sinv = gamma != 0.0 ? (double *)malloc(sizeof(double) * nn) : NULL;
    double *tmpS = (double *)malloc(sizeof(double) * nn);
    double *mu = (double *)malloc(sizeof(double) * cs);
    double *ws = (double *)malloc(sizeof(double) * K);
    int *order = (int *)malloc(sizeof(int) * K);
    double *E = out->E, *cost = out->cost;
    int status = 0, factored = 0;
    /* CMA state :536-545 */
    double sigma = pol->cma_sigma;
    double *p_sigma = (double *)calloc(cs, sizeof(double)), *p_Sigma = (double *)calloc(cs, sizeof(double));
    double *dw = (double *)calloc(cs, sizeof(double)), *C = NULL, *Cdw = NULL, *elite = NULL;
    if (kind == ORC_POL_CMAMPPI) { C = (double *)malloc(sizeof(double) * nn); Cdw = (double *)malloc(sizeof(double) * cs); }
    int m_elite = pol->m_elite;
    if (kind == ORC_POL_CEMPPI || kind == ORC_POL_CMAMPPI) elite = (double *)malloc(sizeof(double) * cs * (size_t)(m_elite > 0 ? m_elite : 1));
    const int sigma_fixed = (kind == ORC_POL_GMPPI || kind == ORC_POL_IMPPI || kind == ORC_POL_MUAISMPPI);

    int n;
    for (n = 1; n <= N; ++n) {
        /* P = MvNormal(Sigma') ; Sigma_inv = invcov(P)  (hoisted for fixed-Sigma variants :352-353,:650-651) */
        if (!(sigma_fixed && factored)) {
            const double *A = Sig;
            if (kind == ORC_POL_CMAMPPI && N > 1) {                                /* :550-554 */
                for (size_t i = 0; i < nn; ++i) tmpS[i] = sigma * sigma * Sig[i];
                A = tmpS;
            }
            if (orc_cholesky_lower(cs, A, L)) { status = -2; break; }              /* PosDefException */
            if (Sinv) orc_inv_from_chol(cs, L, Sinv);
            factored = 1;
        }
        if (out->Sigma_last) {
            if (kind == ORC_POL_CMAMPPI && N > 1) for (size_t i = 0; i < nn; ++i) out->Sigma_last[i] = sigma * sigma * Sig[i];
            else memcpy(out->Sigma_last, Sig, sizeof(double) * nn);
        }
        lmul_LZ(cs, K, L, nz->Z + (size_t)(n - 1) * cs * K, E);                    /* E = rand(rng,P,K) */
        orc_simulate_model(pol, Ucur, env, E, Sinv, U_orig, cost, NULL);
        out->iters_run = n;
        for (int k = 0; k < K; ++k) if (!(fabs(cost[k]) < INFINITY)) status = -3;    /* a rollout's env(a) threw (orc_rollout_model), or a NaN state poisoned the reward */
        if (status) break;
        if (n < N) {
            if (kind == ORC_POL_IMPPI || kind == ORC_POL_MUAISMPPI || kind == ORC_POL_MUSIGMAAISMPPI) {
                double lam = (kind == ORC_POL_IMPPI) ? pol->lambda : pol->lambda_ais;   /* :362 / :660 / :730 */
                orc_compute_weights(lam, cost, K, ws);
                if (kind == ORC_POL_MUSIGMAAISMPPI) {                              /* :731-734 */
                    wmean_wcov(cs, K, E, ws, mu, Sig);
                    for (int i = 0; i < cs; ++i) Sig[i + (size_t)i * cs] += 10e-9;
                } else {
                    wmean_wcov(cs, K, E, ws, mu, NULL);                            /* cov discarded :364,:662 */
                }
                for (int i = 0; i < cs; ++i) Ucur[i] = Ucur[i] + mu[i];
            } else if (kind == ORC_POL_PMCMPPI) {                                  /* :803-809 */
                orc_compute_weights(pol->lambda_ais, cost, K, ws);
                double *acc = (double *)malloc(sizeof(double) * K);
                int32_t *al = (int32_t *)malloc(sizeof(int32_t) * K);
                int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * K);
                orc_make_alias_table(ws, 1.0, K, acc, al);
                orc_alias_sample(acc, al, K, nz->res_i0 + (size_t)(n - 1) * K, nz->res_u + (size_t)(n - 1) * K, K, idx);
                if (out->res_idx0) memcpy(out->res_idx0 + (size_t)(n - 1) * K, idx, sizeof(int32_t) * K);
                /* [3P] StatsBase 0.34 src/cov.jl: mean_and_cov(x, dims = 2; corrected = true): unweighted mean, covariance / (K-1) */
                for (int r = 0; r < cs; ++r) {
                    double s = 0.0;
                    for (int k = 0; k < K; ++k) s += E[r + (size_t)idx[k] * cs];
                    mu[r] = s / K;
                }
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_dense_threads) schedule(dynamic, 1)
#endif
                for (int b = 0; b < cs; ++b)
                    for (int a = 0; a <= b; ++a) {
                        double s = 0.0;
                        for (int k = 0; k < K; ++k)
                            s += (E[a + (size_t)idx[k] * cs] - mu[a]) * (E[b + (size_t)idx[k] * cs] - mu[b]);
                        Sig[a + (size_t)b * cs] = Sig[b + (size_t)a * cs] = s / (K - 1);
                    }
                for (int i = 0; i < cs; ++i) Sig[i + (size_t)i * cs] += 10e-9;
                for (int i = 0; i < cs; ++i) Ucur[i] = Ucur[i] + mu[i];
                free(acc); free(al); free(idx);
            } else if (kind == ORC_POL_CEMPPI || kind == ORC_POL_CMAMPPI) {
                sortperm(cost, K, order);                                          /* :455 / :563 */
                for (int j = 0; j < m_elite; ++j) memcpy(elite + (size_t)j * cs, E + (size_t)order[j] * cs, sizeof(double) * cs);
                double maxdiff = -INFINITY;                                        /* :458-461 / :566-569 */
                for (int j = 0; j + 1 < m_elite; ++j) {
                    double d = fabs(cost[order[j + 1]] - cost[order[j]]);
                    if (d > maxdiff) maxdiff = d;
                }
                if (m_elite >= 2 && maxdiff < 10e-3) break;
                if (kind == ORC_POL_CEMPPI) {                                      /* :464-465 */
                    if (pol->sigma_est == ORC_SIGMA_EST_SS) cov_ss_cols(cs, m_elite, elite, mu, Sig);
                    else if (pol->sigma_est == ORC_SIGMA_EST_LW) cov_lw_cols(cs, m_elite, elite, mu, Sig);
                    else if (pol->sigma_est == ORC_SIGMA_EST_RBLW) cov_common_cols(cs, m_elite, elite, mu, Sig, 0);
                    else if (pol->sigma_est == ORC_SIGMA_EST_OAS) cov_common_cols(cs, m_elite, elite, mu, Sig, 1);
                    else cov_mle_cols(cs, m_elite, elite, mu, Sig);
                    for (int i = 0; i < cs; ++i) Sig[i + (size_t)i * cs] += 10e-9;
                    for (int i = 0; i < cs; ++i) Ucur[i] = Ucur[i] + mu[i];
                } else {
                    const double *cw = pol->ws;
                    const double c_s = pol->c_sigma, d_s = pol->d_sigma, c_S = pol->c_Sigma, c1 = pol->c1, c_mu = pol->c_mu;
                    const double mu_eff = pol->mu_eff, E_cma = pol->E_cma;
                    const double sigma_old = sigma;                                /* ds = elite_E/sigma :572 */
                    for (int r = 0; r < cs; ++r) {                                 /* :573-576 */
                        double s = 0.0;
                        for (int j = 0; j < m_elite; ++j) s += cw[j] * elite[r + (size_t)j * cs];
                        dw[r] = s;
                    }
                    for (int r = 0; r < cs; ++r) Ucur[r] += sigma * dw[r];         /* :577 */
                    if (orc_sym_pow(cs, Sig, -0.5, C)) { status = -2; break; }     /* :580 */
                    double sc = sqrt(c_s * (2 - c_s) * mu_eff);
                    for (int i = 0; i < cs; ++i) {                                 /* :581 (s*C)*dw */
                        double v = 0.0;
                        for (int j = 0; j < cs; ++j) v += (sc * C[i + (size_t)j * cs]) * dw[j];
                        Cdw[i] = v;
                    }
                    double nps = 0.0;
                    for (int i = 0; i < cs; ++i) { p_sigma[i] = (1 - c_s) * p_sigma[i] + Cdw[i]; nps += p_sigma[i] * p_sigma[i]; }
                    nps = sqrt(nps);
                    sigma *= exp(c_s / d_s * (nps / E_cma - 1));                   /* :582 */
                    int h_sigma = (nps / sqrt(1 - pow(1 - c_s, 2.0 * n)) < (1.4 + 2.0 / (cs + 1)) * E_cma) ? 1 : 0; /* :585 */
                    double sS = h_sigma * sqrt(c_S * (2 - c_S) * mu_eff);
                    for (int i = 0; i < cs; ++i) p_Sigma[i] = (1 - c_S) * p_Sigma[i] + sS * dw[i];   /* :586 */
                    /* δs[order[ii]] (:593) is a LINEAR index up to K into the cs x m_elite matrix: Julia throws BoundsError when cs*m_elite < K
                     * (the engine refuses such a handle at creation with MPOPIS_ERR_ARG and the same message).  Until round 6 this loop read past
                     * `elite` in that case. */
                    if ((long long)cs * m_elite < K) { status = -1; break; }
                    double normC = 0.0;                 /* Frobenius pieces of norm(C*scalar) */
                    double temp_sum = 0.0;                                         /* :588-596 (scalar! quirk) */
                    for (int ii = 0; ii < K; ++ii) {
                        int j = order[ii];              /* linear (0-based) index into ds = elite_E/sigma */
                        double d = elite[j] / sigma_old;                           /* requires j < cs*m_elite */
                        double w0;
                        if (cw[ii] >= 0) w0 = cw[ii];
                        else {
                            normC = 0.0;
                            for (size_t q = 0; q < nn; ++q) { double v = C[q] * d; normC += v * v; }
                            normC = sqrt(normC);
                            w0 = n * cw[ii] / (normC * normC);                     /* n = iteration index (quirk) */
                        }
                        temp_sum += w0 * d * d;
                    }
                    for (int b = 0; b < cs; ++b)                                   /* :598 */
                        for (int a = 0; a < cs; ++a) {
                            double S_ab = Sig[a + (size_t)b * cs];
                            tmpS[a + (size_t)b * cs] = (1 - c1 - c_mu) * S_ab
                                + c1 * (p_Sigma[a] * p_Sigma[b] + (1 - h_sigma) * c_S * (2 - c_S) * S_ab)
                                + c_mu * temp_sum;
                        }
                    for (int b = 0; b < cs; ++b)                                   /* :599 triu(S)+triu(S,1)' */
                        for (int a = 0; a < cs; ++a)
                            Sig[a + (size_t)b * cs] = (a <= b) ? tmpS[a + (size_t)b * cs] : tmpS[b + (size_t)a * cs];
                }
            }
        } else {
            /* n == N: weights from the policy's own lambda (:367,:665,:736,:811; CE/CMA/GMPPI after loop) */
        }
    }
    if (status == 0) {
        for (int k = 0; k < K; ++k)                                                /* E .+= (pol.U - U_orig) */
            for (int r = 0; r < cs; ++r) E[r + (size_t)k * cs] = E[r + (size_t)k * cs] + (Ucur[r] - U_orig[r]);
        if (out->U_last) memcpy(out->U_last, Ucur, sizeof(double) * cs);
        orc_compute_weights(pol->lambda, cost, K, out->weights);
        /* functor :226-231 */
        double *wc = (double *)malloc(sizeof(double) * cs);
        for (int r = 0; r < cs; ++r) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += out->weights[k] * E[r + (size_t)k * cs];
            wc[r] = U_orig[r] + s;
        }
        orc_roll_U(pol, wc, out->control);
        free(wc);
    }
    free(Ucur); free(Sig); free(L); free(Sinv); free(tmpS); free(mu); free(ws); free(order);
    free(p_sigma); free(p_Sigma); free(dw); free(C); free(Cdw); free(elite);
    return out->status = status;
}

/* ======================================================================================
 * closed loop harness: src/examples/car_example.jl:170-326; mountaincar_example.jl:125-180
 * noise streams: normals (step, iter); resampling draws (step, iter | 0x80000000)
 * ====================================================================================== */
int orc_run_trial(orc_policy *pol, orc_env *env, uint64_t seed, int num_steps, int laps,
                  orc_trial_record *rec, double *act_log) {
    return orc_run_trial_noise(pol, env, seed, num_steps, laps, 0.0, 0.0, 0.0, rec, act_log);
}

/* state_x_sigma / state_y_sigma / state_ψ_sigma: car_example.jl:38-40,224-236 (single car only); the three normals per
 * step come from the stream (step, 0x40000000) and are only drawn when a sigma is non-zero */
int orc_run_trial_noise(orc_policy *pol, orc_env *env, uint64_t seed, int num_steps, int laps,
                        double sx, double sy, double spsi, orc_trial_record *rec, double *act_log) {
    const int K = pol->K, cs = pol->cs, as = pol->as, T = pol->T;
    const int N = (pol->kind == ORC_POL_GMPPI || pol->kind == ORC_POL_MPPI) ? 1 : pol->N;
    size_t nZ = (pol->kind == ORC_POL_MPPI) ? (size_t)K * T * as : (size_t)cs * K;
    double *Z = (double *)malloc(sizeof(double) * nZ * N);
    int32_t *ri = (int32_t *)malloc(sizeof(int32_t) * (size_t)K * N);
    double *ru = (double *)malloc(sizeof(double) * (size_t)K * N);
    orc_step_out out; memset(&out, 0, sizeof(out));
    double control[16];
    out.control = control;
    out.cost = (double *)malloc(sizeof(double) * K);
    out.weights = (double *)malloc(sizeof(double) * K);
    out.E = (double *)malloc(sizeof(double) * nZ);
    memset(rec, 0, sizeof(*rec));
    double rew = 0.0; int cnt = 0, lap = 0; double prev_y = 0.0;
    int trk_viol = 0, beta_viol = 0, crash_viol = 0;
    double v_mean_sum = 0, v_max = -INFINITY, b_mean_sum = 0, b_max = -INFINITY;
    int lap_time[4] = {0, 0, 0, 0};
    int status = 0;
    const int is_car = env->kind == ORC_ENV_CAR;
    while (!env->done && cnt <= num_steps) {                                       /* :203 */
        for (int n = 0; n < N; ++n) {
            orc_philox_normals(seed, (uint32_t)cnt, (uint32_t)n, (int64_t)nZ, Z + nZ * n);
            if (pol->kind == ORC_POL_PMCMPPI)
                orc_philox_resample_draws(seed, (uint32_t)cnt, (uint32_t)n | 0x80000000u, K, ri + (size_t)K * n, ru + (size_t)K * n);
        }
        orc_noise nz = { Z, ri, ru };
        status = orc_policy_call(pol, env, &nz, &out);                             /* :205 */
        if (status) break;
        rec->rollouts += (double)out.iters_run * K;
        if (act_log) memcpy(act_log + (size_t)cnt * as, control, sizeof(double) * as);
        if (orc_env_step(env, control)) { status = -3; break; }                    /* :207 */
        cnt += 1;
        double step_rew = orc_env_reward(env);                                     /* :210 */
        rew += step_rew;
        if (is_car && env->ncars == 1 && (sx != 0.0 || sy != 0.0 || spsi != 0.0)) { /* :224-236 */
            double z[4];
            orc_philox_normals(seed, (uint32_t)(cnt - 1), 0x40000000u, 4, z);
            env->state[0] += sx * z[0];
            env->state[1] += sy * z[1];
            double dpsi = spsi * z[2];
            env->state[2] += dpsi;
            double c = cos(dpsi), sn = sin(dpsi), vx = env->state[3], vy = env->state[4];
            env->state[3] = c * vx + sn * vy;                                      /* passive rotation [c s; -s c] */
            env->state[4] = -sn * vx + c * vy;
        }
        if (!is_car) continue;
        double curr_y = env->state[1];                                             /* :241-253 */
        double vmean = 0, vmax = -INFINITY, bmean = 0, bmax = -INFINITY, d = INFINITY;
        for (int c = 0; c < env->ncars; ++c) {
            const double *s = env->state + 8 * c;
            if (c == 0 || s[1] < curr_y) curr_y = s[1];
            double v = sqrt(s[3] * s[3] + s[4] * s[4]);
            double b = fabs(orc_calculate_beta(s));
            vmean += v; bmean += b;
            if (v > vmax) vmax = v;
            if (b > bmax) bmax = b;
            double dd = sqrt(s[0] * s[0] + s[1] * s[1]);
            if (dd < d) d = dd;
        }
        vmean /= env->ncars; bmean /= env->ncars;
        v_mean_sum += vmean; b_mean_sum += bmean;
        if (vmax > v_max) v_max = vmax;
        if (bmax > b_max) b_max = bmax;
        if (step_rew < -4000) {                                                    /* :256-263 */
            int ex_b = 0, within_t = 1;
            for (int c = 0; c < env->ncars; ++c) {
                const double *s = env->state + 8 * c; double dist;
                if (fabs(orc_calculate_beta(s)) > env->params[ORC_CP_BETALIM]) ex_b = 1;
                if (!orc_within_track(env->P, env->tx, env->ty, env->tw, s, &dist)) within_t = 0;
            }
            if (ex_b) beta_viol += 1;
            if (!within_t) trk_viol += 1;
            double temp_rew = step_rew + ex_b * 5000 + (!within_t) * 1000000;
            if (temp_rew < -10500) crash_viol += 1;
        }
        if (prev_y < 0.0 && curr_y >= 0.0 && d <= 15.0) {                          /* :273-276 */
            lap += 1;
            if (lap <= 4) lap_time[lap - 1] = cnt;
        }
        if (lap >= laps || trk_viol > 10 || beta_viol > 50) env->done = 1;         /* :277-279 */
        prev_y = curr_y;
    }
    rec->rew = rew; rec->steps = cnt - 1; rec->rew_per_step = rew / (cnt - 1);     /* :287-289 */
    for (int i = 0; i < 4; ++i) rec->lap_t[i] = lap_time[i];
    if (is_car && cnt > 0) {
        rec->mean_v = v_mean_sum / cnt; rec->max_v = v_max;
        rec->mean_beta = b_mean_sum / cnt; rec->max_beta = b_max;
    }
    rec->beta_viol = beta_viol; rec->trk_viol = trk_viol; rec->crash_viol = crash_viol;
    free(Z); free(ri); free(ru); free(out.cost); free(out.weights); free(out.E);
    return status;
}

/* src/examples/example_utils.jl:2-10 (p=0.05, q=0.5); quantile(x,0.5) = Statistics stdlib quantile, type 7 (linear interpolation) [3P] */
static int cmp_d(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }
void orc_quantile_ci(const double *x, int n, double *lo, double *med, double *hi) {
    double *s = (double *)malloc(sizeof(double) * n);
    memcpy(s, x, sizeof(double) * n);
    qsort(s, n, sizeof(double), cmp_d);
    const double zm = -1.959963984540054, zp = 1.959963984540054, q = 0.5;
    int j = (int)ceil(n * q + zm * sqrt(n * q * (1 - q))); if (j < 1) j = 1;
    int k = (int)ceil(n * q + zp * sqrt(n * q * (1 - q))); if (k > n) k = n;
    *lo = s[j - 1]; *hi = s[k - 1];
    double h = (n - 1) * q; int fl = (int)floor(h);
    *med = (fl + 1 < n) ? s[fl] + (h - fl) * (s[fl + 1] - s[fl]) : s[fl];
    free(s);
}
