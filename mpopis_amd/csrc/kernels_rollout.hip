// kernels_rollout.hip -- the fused model-rollout kernel: replaces, per sample k,
//   simulate_model            src/mppi_mpopi_policies.jl:261-278  (V = pol.U + E[:,k], control cost, clamp)
//   calculate_trajectory_costs(::MPPI_Policy) inner loop  :198-214
//   get_model_controls        src/utils.jl:55-67
//   rollout_model             src/utils.jl:129-144
//   env(a) + reward(env)      car_racing.jl:238-344,201-213; multi-car_racing.jl:200-207,145-158;
//                             mountaincar_example.jl:4-22
// in ONE launch, state in registers, no per-sample env copies (the reference deep-copies the env
// per sample, :270).
//
// Mapping (CDNA4): lane = sample k, wave = car.  A workgroup is 64 samples x NC cars (NC waves); the
// cars of one sample exchange (x,y) through LDS once per model step for the pairwise terms of the
// multi-car reward.  E is [cs][K] (K fastest) so each per-step control load is one coalesced
// 512-B transaction per wave; the nominal control U, the env state and the 48-point track are
// wave-uniform and arrive through the scalar cache.  The kernel is FP64-VALU bound;
// HBM traffic is 8*cs bytes per sample.
#include "engine.h"

namespace mpopis {

// NC cars x SPB sample-waves per workgroup: the SPB*64 samples of a workgroup share one LDS copy of the track tables
// LOG: the trajectory logger is on (a.traj != nullptr) -- only then is the heading angle psi itself tracked
template <int NC, int SPB, bool LOG>
__global__ void __launch_bounds__(64 * NC * SPB) __attribute__((amdgpu_waves_per_eu(4, 4))) k_rollout_car(RolloutArgs a) {
    const int b = blockIdx.y;
    if (a.active && !a.active[b]) return;
    if (a.iters && blockIdx.x == 0 && threadIdx.x == 0) a.iters[b] = a.iter_n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = (NC > 1) ? wave % NC : 0;                   // car of this wave
    const int g = wave / NC;                                  // sample group of this wave
    const int k = (blockIdx.x * SPB + g) * 64 + lane;
    const int K = a.K, T = a.T;
    const bool valid = k < K;
    const int kk = valid ? k : K - 1;
    constexpr int as = 2 * NC, ss = 8 * NC;
    (void)ss;

    const CarParams& p = a.env.car;
    // stage the (wave-uniform, read-only) track in LDS: uniform-address ds_reads broadcast to all lanes
    extern __shared__ __attribute__((aligned(16))) double sh_trk[];
    const int P = a.env.track.P;
    const int W = a.env.track.nbrw, NS = P * (W + 1);
    double* sh_nd = sh_trk + 4 * P;                           // neighbour distances [P][W+1]
    int* sh_ni = reinterpret_cast<int*>(sh_nd + NS);          // neighbour indices   [P][W+1]
    for (int i = threadIdx.x; i < P; i += 64 * NC * SPB) {
        sh_trk[i] = a.env.track.x[i]; sh_trk[P + i] = a.env.track.y[i]; sh_trk[2 * P + i] = a.env.track.w[i];
        sh_trk[3 * P + i] = a.env.track.n2[i];
    }
    for (int i = threadIdx.x; i < NS; i += 64 * NC * SPB) { sh_nd[i] = a.env.track.nbr_dist[i]; sh_ni[i] = a.env.track.nbr_idx[i]; }
    __syncthreads();
    const Track tk{sh_trk, sh_trk + P, sh_trk + 2 * P, sh_trk + 3 * P, P, sh_ni, sh_nd, W};
    CarState s;                                               // wave-uniform start state (+ sin/cos), scalar loads
    {
        const double* xe = a.x0ext + ((size_t)b * NC + c) * kCarExt;
        s.x = xe[0]; s.y = xe[1]; s.psi = xe[2]; s.Vx = xe[3]; s.Vy = xe[4]; s.r = xe[5]; s.delta = xe[6]; s.pedal = xe[7];
        s.sp = xe[8]; s.cp = xe[9]; s.sd = xe[10]; s.cd = xe[11]; s.near = -1;
    }
    const double* Eb = a.E + (size_t)b * a.cs * K + (size_t)(2 * c) * K + kk;
    const double* Ub = a.Ucur + (size_t)b * a.cs + 2 * c;
    const double* Uo = a.Uorig + (size_t)b * a.cs + 2 * c;
    const double* gv = a.gvec ? a.gvec + (size_t)b * a.cs + 2 * c : nullptr;
    const double lo0 = a.env.lo[2 * c], hi0 = a.env.hi[2 * c], lo1 = a.env.lo[2 * c + 1], hi1 = a.env.hi[2 * c + 1];
    double* tr = LOG ? a.traj + ((size_t)b * K + kk) * (size_t)(ss * T) : nullptr;

    __shared__ double sh_xy[2][SPB][NC][2][64];
    __shared__ double sh_cost[SPB][NC][64];

    double cost = 0.0, cc = 0.0;
    double e0 = Eb[0], e1 = Eb[K], u0 = Ub[0], u1 = Ub[1];
    for (int t = 0; t < T; ++t) {
        const double v0 = u0 + e0, v1 = u1 + e1;                               // V = pol.U + E[:,k]  :271
        if (t + 1 < T) {                                                       // next step's noise (global) and nominal control (scalar) in flight during this step
            e0 = Eb[(size_t)(t + 1) * as * K]; e1 = Eb[(size_t)(t + 1) * as * K + K];
            u0 = Ub[(t + 1) * as]; u1 = Ub[(t + 1) * as + 1];
        }
        if (__builtin_expect(gv != nullptr, 0)) cc += gv[t * as] * (v0 - Uo[t * as]) + gv[t * as + 1] * (v1 - Uo[t * as + 1]);   // :272 (unclamped V; γ = 0 in every reference config)
        const double a0 = clampd_u(v0, lo0, hi0), a1 = clampd_u(v1, lo1, hi1); // get_model_controls
        car_action_step<LOG>(p, s, a0, a1, (t & 3) == 0);                      // unit-circle renormalisation every 4th step
        double rew = car_reward(p, tk, s.x, s.y, s.Vx, s.Vy, &s.near);
        if (NC > 1) {                                                          // multi-car_racing.jl:145-158
            const int buf = t & 1;
            sh_xy[buf][g][c][0][lane] = s.x;
            sh_xy[buf][g][c][1][lane] = s.y;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                if (j > c) {
                    const double dx = sh_xy[buf][g][j][0][lane] - s.x, dy = sh_xy[buf][g][j][1][lane] - s.y;
                    const double dd = sqrt(dx * dx + dy * dy);
                    rew += -dd;
                    if (dd <= 4.0) rew += -11000.0;
                }
            }
        }
        cost -= rew;                                                           // utils.jl:138
        if (LOG && valid) {                                                    // trajectories[k][t, :] utils.jl:140
            double s8[8];
            car_state_to8(s, s8);
#pragma unroll
            for (int i = 0; i < 8; ++i) tr[(size_t)(8 * c + i) * T + t] = s8[i];
        }
    }
    cost += cc;
    double total = cost;
    bool writer = true;
    if (NC > 1) {
        sh_cost[g][c][lane] = cost;
        __syncthreads();
        writer = (c == 0);
        if (c == 0) {
#pragma unroll
            for (int j = 1; j < NC; ++j) total += sh_cost[g][j][lane];
        }
    }
    if (writer && valid) a.cost[(size_t)b * K + k] = total;
    if (a.cmin && writer) {
        // ρ = minimum(costs) (utils.jl:81) accumulates here, one atomic per wave, so that the AIS reweighting can be folded into the moments
        // kernel (launch_wcov_mfma, weights from costs) instead of a launch of its own between the two
        unsigned long long key = valid ? cost_key(total) : ~0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(key, o, 64); key = (t < key) ? t : key; }
        if (lane == 0) atomicMin(&a.cmin[b], key);
        if (valid && !(fabs(total) < INFINITY) && a.status) atomicMin(&a.status[b], MPOPIS_ERR_ACTION);   // non-finite cost <=> NaN action (car_racing.jl:239)
    }
}

// MountainCar (ss = 2) and CartPole (ss = 4): scalar action, a handful of flops per step
template <int SS>
__global__ void __launch_bounds__(64) k_rollout_simple(RolloutArgs a) {
    const int b = blockIdx.y;
    if (a.active && !a.active[b]) return;
    if (a.iters && blockIdx.x == 0 && threadIdx.x == 0) a.iters[b] = a.iter_n;
    const int k = blockIdx.x * 64 + threadIdx.x;
    const int K = a.K, T = a.T;
    const bool valid = k < K;
    const int kk = valid ? k : K - 1;
    double s[SS];
#pragma unroll
    for (int i = 0; i < SS; ++i) s[i] = a.x0[b * SS + i];
    int t_env = a.t0 ? a.t0[b] : 0, done = a.done0 ? a.done0[b] : 0;
    const double* Eb = a.E + (size_t)b * a.cs * K + kk;
    const double* Ub = a.Ucur + (size_t)b * a.cs;
    const double* Uo = a.Uorig + (size_t)b * a.cs;
    const double* gv = a.gvec ? a.gvec + (size_t)b * a.cs : nullptr;
    double* tr = a.traj ? a.traj + ((size_t)b * K + kk) * (size_t)(SS * T) : nullptr;
    double cost = 0.0, cc = 0.0;
    for (int t = 0; t < T; ++t) {
        const double v = Ub[t] + Eb[(size_t)t * K];
        if (gv) cc += gv[t] * (v - Uo[t]);
        const double act = clampd(v, a.env.lo[0], a.env.hi[0]);
        simple_env_step(a.env, s, &t_env, &done, act);
        cost -= simple_env_reward(a.env, s, done);
        if (tr && valid) {
#pragma unroll
            for (int i = 0; i < SS; ++i) tr[i * T + t] = s[i];
        }
    }
    if (valid) a.cost[(size_t)b * K + k] = cost + cc;
}

// x0ext[b][car][12] = env state + sin/cos(psi), sin/cos(delta): evaluated once per trial and car
// instead of once per sample (the start state is shared by all K rollouts).
__global__ void k_extend_state(const double* x, double* xext, int n_cars_total) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_cars_total) return;
    CarState c;
    car_state_from8(c, x + (size_t)i * 8);
    double* o = xext + (size_t)i * kCarExt;
    o[0] = c.x; o[1] = c.y; o[2] = c.psi; o[3] = c.Vx; o[4] = c.Vy; o[5] = c.r; o[6] = c.delta; o[7] = c.pedal;
    o[8] = c.sp; o[9] = c.cp; o[10] = c.sd; o[11] = c.cd;
}
void launch_extend_state(const double* x, double* xext, int B, int ncars, hipStream_t st) {
    const int n = B * ncars;
    hipLaunchKernelGGL(k_extend_state, dim3((n + 63) / 64), dim3(64), 0, st, x, xext, n);
}

// Start of an MPC step, one launch instead of seven (at one trial the step is a chain of dependent launches, each boundary costs 2-4 us):
// status = 0 (unless sticky), active = alive gate (or 1), iters = 0, U_orig = the loop's pol.U = pol.U, and the car start states extended
// with sin/cos of psi / delta (as k_extend_state).
__global__ void __launch_bounds__(256) k_step_begin(int* status, int* active, const int* alive, int* iters, const double* U, double* Uin, double* Ucur,
                                                    int cs, const double* x, double* xext, int ncars, unsigned long long* cmin) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) { if (status) status[b] = 0; active[b] = alive ? alive[b] : 1; iters[b] = 0; if (cmin) cmin[b] = ~0ull; }
    for (int i = tid; i < cs; i += 256) { const double u = U[(size_t)b * cs + i]; Uin[(size_t)b * cs + i] = u; Ucur[(size_t)b * cs + i] = u; }
    if (x && tid < ncars) {
        const size_t i = (size_t)b * ncars + tid;
        CarState c;
        car_state_from8(c, x + i * 8);
        double* o = xext + i * kCarExt;
        o[0] = c.x; o[1] = c.y; o[2] = c.psi; o[3] = c.Vx; o[4] = c.Vy; o[5] = c.r; o[6] = c.delta; o[7] = c.pedal;
        o[8] = c.sp; o[9] = c.cp; o[10] = c.sd; o[11] = c.cd;
    }
}
void launch_step_begin(int* status, int* active, const int* alive, int* iters, const double* U, double* Uin, double* Ucur, int B, int cs,
                       const double* x, double* xext, int ncars, hipStream_t st, unsigned long long* cmin) {
    hipLaunchKernelGGL(k_step_begin, dim3(B), dim3(256), 0, st, status, active, alive, iters, U, Uin, Ucur, cs, x, xext, ncars, cmin);
}

void launch_rollout(const RolloutArgs& a, hipStream_t st) {
    if (a.env.kind == MPOPIS_ENV_MOUNTAINCAR) {
        hipLaunchKernelGGL(k_rollout_simple<2>, dim3((a.K + 63) / 64, a.B), dim3(64), 0, st, a);
        return;
    }
    if (a.env.kind == MPOPIS_ENV_CARTPOLE) {
        hipLaunchKernelGGL(k_rollout_simple<4>, dim3((a.K + 63) / 64, a.B), dim3(64), 0, st, a);
        return;
    }
    const int P = a.env.track.P, W = a.env.track.nbrw;
    const size_t lds = (size_t)4 * P * sizeof(double) + (size_t)P * (W + 1) * (sizeof(double) + sizeof(int));
    // small K: one sample-wave per workgroup keeps every wave on its own CU; large K: 4 sample-waves share the LDS tables
    const bool wide = a.K >= 1024;
    const dim3 g1((a.K + 63) / 64, a.B), g4((a.K + 255) / 256, a.B), g2((a.K + 127) / 128, a.B);
#define MPOPIS_LAUNCH_CAR(NC, SPB, GRID, BLOCK)                                                        \
    do {                                                                                               \
        if (a.traj) hipLaunchKernelGGL((k_rollout_car<NC, SPB, true>), GRID, dim3(BLOCK), lds, st, a);  \
        else        hipLaunchKernelGGL((k_rollout_car<NC, SPB, false>), GRID, dim3(BLOCK), lds, st, a); \
    } while (0)
    switch (a.env.ncars) {
        case 1: if (wide) MPOPIS_LAUNCH_CAR(1, 4, g4, 256); else MPOPIS_LAUNCH_CAR(1, 1, g1, 64); break;
        case 2: if (wide) MPOPIS_LAUNCH_CAR(2, 2, g2, 256); else MPOPIS_LAUNCH_CAR(2, 1, g1, 128); break;
        case 3: MPOPIS_LAUNCH_CAR(3, 1, g1, 192); break;
        case 4: MPOPIS_LAUNCH_CAR(4, 1, g1, 256); break;
        default: break;
    }
#undef MPOPIS_LAUNCH_CAR
}

}  // namespace mpopis
