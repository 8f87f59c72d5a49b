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
// Mapping (CDNA4): lane = one car of one sample.  One car (k_rollout_car): a wave = 64 samples; E is [cs][K] (K fastest) so
// each per-step control load is one coalesced 512-B transaction per wave; the nominal control U, the env state and
// the 48-point track are wave-uniform and arrive through the scalar cache / LDS broadcasts.  NC cars (k_rollout_cars):
// a wave = 64/NC samples x NC cars, lane = c * S + j, so the cars of one sample sit in ONE wave and exchange (x,y) for
// the pairwise terms of the multi-car reward by lane shuffles -- no LDS exchange, no barrier in the time loop (round 2
// ran wave = car with a workgroup barrier per model step: 5.9 cycles per VALU instruction at 64 trials against 4.8
// for the one-car kernel).  The kernels are FP64-VALU bound; HBM traffic is 8*cs bytes per sample.
#include "engine.h"

namespace mpopis {

// One car (NC = 1, CarRacingEnv): SPB sample-waves per workgroup share one LDS copy of the track tables
// LOG: the trajectory logger is on (a.traj != nullptr) -- only then is the heading angle psi itself tracked
// Track tables into LDS (dynamic LDS layout: ring table [(P+6)][4] + certification radii [2][P]; with ALL: + x, y, w, |q|^2 [4][P] + neighbour
// distances [P][W+1] + neighbour indices [P][W+1]).  Returns the Track the rollout uses; the caller synchronises.
template <bool ALL>
__device__ __forceinline__ Track stage_track(const Track& g, double* sh, int tid, int nthreads) {
    const int P = g.P, W = g.nbrw, NS = P * (W + 1);
    double* sh_ring = sh;
    double* sh_cert = sh + 4 * (P + 2 * kRingPad);
    for (int i = tid; i < 4 * (P + 2 * kRingPad); i += nthreads) sh_ring[i] = g.ring[i];
    for (int i = tid; i < 2 * P; i += nthreads) sh_cert[i] = g.ring_cert[i];      // three-point and five-point certificates
    if (!ALL) return Track{g.x, g.y, g.w, g.n2, P, g.nbr_idx, g.nbr_dist, W, sh_ring, sh_cert};
    double* sh_trk = sh_cert + 2 * P;
    double* sh_nd = sh_trk + 4 * P;
    int* sh_ni = reinterpret_cast<int*>(sh_nd + NS);
    for (int i = tid; i < P; i += nthreads) { sh_trk[i] = g.x[i]; sh_trk[P + i] = g.y[i]; sh_trk[2 * P + i] = g.w[i]; sh_trk[3 * P + i] = g.n2[i]; }
    for (int i = tid; i < NS; i += nthreads) { sh_nd[i] = g.nbr_dist[i]; sh_ni[i] = g.nbr_idx[i]; }
    return Track{sh_trk, sh_trk + P, sh_trk + 2 * P, sh_trk + 3 * P, P, sh_ni, sh_nd, W, sh_ring, sh_cert};
}
inline size_t track_lds_bytes(int P, int W, bool all) {
    return (size_t)(4 * (P + 2 * kRingPad) + 2 * P) * sizeof(double) + (all ? (size_t)4 * P * sizeof(double) + (size_t)P * (W + 1) * (sizeof(double) + sizeof(int)) : 0);
}

// TLDS: every track table fits the default 64 KB of dynamic LDS (P <= ~230 points: all bundled tracks).  Otherwise only the ring table of the
// straight-line nearest-point search is staged (40 B per point) and the general search -- first step of a rollout, lanes far off their
// anchor -- reads the coordinate / neighbour tables from global memory.
#ifdef MPOPIS_ROLL_PROF
// dev build (tools/roll_prof.sh): start / end time (s_memrealtime, 100 MHz) and hardware id of every wave of the last 1-car launch.
// What it showed (C5, 64 trials, 4 waves per SIMD): the waves of a SIMD do not progress together -- issue arbitration is oldest-first, the oldest
// wave runs at the lone-wave rate and ends at ~146 us, the youngest at ~376 us -- but evening them out with rotating s_setprio levels (per action,
// by wave slot and clock) narrowed the spread to 259..374 us WITHOUT shortening the launch: the SIMD's throughput is the same either way,
// ~5.0 cycles per FP64 instruction at 4 waves (tools/mfma_rate.hip measures 5.0 for bare v_fma_f64 streams at 4 waves per SIMD, 4.6 at 8), and
// 43.0 k instructions x 4 waves x 5.0 cycles = the measured launch.  The kernel is at the attainable issue rate; only fewer instructions help.
__device__ unsigned long long g_roll_prof[3 * 8192];
extern "C" int mpopis_debug_roll_prof(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_roll_prof), sizeof(g_roll_prof)); }
#endif
#ifdef MPOPIS_PATH_STATS
extern "C" int mpopis_debug_path_stats(unsigned long long* out, int reset) {      // dev build: read (and optionally clear) the path counters of car_dynamics.h
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_path_stats), sizeof(g_path_stats));
    if (reset) { unsigned long long z[8] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_path_stats), z, sizeof z); }
    return rc;
}
extern "C" int mpopis_debug_sick(unsigned char* out, int n, int reset) {       // per-thread flags of the launches since the last reset
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sick), (size_t)n);
    if (reset) { void* p = nullptr; rc |= (int)hipGetSymbolAddress(&p, HIP_SYMBOL(g_sick)); rc |= (int)hipMemset(p, 0, sizeof(g_sick)); }
    return rc;
}
#endif
template <int NC, int SPB, bool LOG, bool TLDS>
__global__ void __launch_bounds__(64 * NC * SPB) __attribute__((amdgpu_waves_per_eu(4, 4))) k_rollout_car(RolloutArgs a) {
    static_assert(NC == 1, "multi-car envs run k_rollout_cars");
#ifdef MPOPIS_ROLL_PROF
    const unsigned long long prof_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int b = blockIdx.y;
    if (a.active && !a.active[b]) return;
    if (a.iters && blockIdx.x == 0 && threadIdx.x == 0) a.iters[b] = a.iter_n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = 0;                                          // the car
    const int g = wave;                                       // sample group of this wave
    const int k = (blockIdx.x * SPB + g) * 64 + lane;
    const int K = a.K, T = a.T;
    const bool valid = k < K;
    const int kk = valid ? k : K - 1;
    constexpr int as = 2 * NC, ss = 8 * NC;
    (void)ss;

    const CarParams& p = a.env.car;
    // stage the (wave-uniform, read-only) track in LDS: uniform-address ds_reads broadcast to all lanes
    extern __shared__ __attribute__((aligned(16))) double sh_dyn[];
    const Track tk = stage_track<TLDS>(a.env.track, sh_dyn, threadIdx.x, 64 * NC * SPB);
    __syncthreads();
    CarState s;                                               // wave-uniform start state (+ sin/cos), scalar loads
    {
        const double* xe = a.x0ext + ((size_t)b * NC + c) * kCarExt;
        s.x = xe[0]; s.y = xe[1]; s.psi = xe[2]; s.Vx = xe[3]; s.Vy = xe[4]; s.r = xe[5]; s.delta = xe[6]; s.pedal = xe[7];
        s.sp = xe[8]; s.cp = xe[9]; s.sd = xe[10]; s.cd = xe[11]; s.near = (int)xe[12];
    }
    const double* Eb = a.E + (size_t)b * a.cs * K + (size_t)(2 * c) * K + kk;
    const double* Ub = a.Ucur + (size_t)b * a.cs + 2 * c;
    const double* Uo = a.Uorig + (size_t)b * a.cs + 2 * c;
    const double* gv = a.gvec ? a.gvec + (size_t)b * a.cs + 2 * c : nullptr;
    const double lo0 = a.env.lo[2 * c], hi0 = a.env.hi[2 * c], lo1 = a.env.lo[2 * c + 1], hi1 = a.env.hi[2 * c + 1];
    double* tr = LOG ? a.traj + ((size_t)b * K + kk) * (size_t)(ss * T) : nullptr;

    double cost = 0.0, cc = 0.0;
    double e0 = Eb[0], e1 = Eb[K], u0 = Ub[0], u1 = Ub[1];
    for (int t = 0; t < T; ++t) {
        const double v0 = u0 + e0, v1 = u1 + e1;                               // V = pol.U + E[:,k]  :271
        if (t + 1 < T) {                                                       // next step's noise (global) and nominal control (scalar) in flight during this step
            e0 = Eb[(size_t)(t + 1) * as * K]; e1 = Eb[(size_t)(t + 1) * as * K + K];
            u0 = Ub[(t + 1) * as]; u1 = Ub[(t + 1) * as + 1];
        }
        if (__builtin_expect(gv != nullptr, 0)) cc += control_cost_term(gv[t * as], v0 - Uo[t * as], gv[t * as + 1], v1 - Uo[t * as + 1]);   // :272 (unclamped V; γ = 0 in every reference config)
        const double a0 = clampd_u(v0, lo0, hi0), a1 = clampd_u(v1, lo1, hi1); // get_model_controls
        car_action_step<LOG>(p, s, a0, a1, (t & 3) == 0);                      // unit-circle renormalisation every 4th step
        const double rew = car_reward(p, tk, s.x, s.y, s.Vx, s.Vy, &s.near);
        cost -= rew;                                                           // utils.jl:138
        if (LOG && valid) {                                                    // trajectories[k][t, :] utils.jl:140
            double s8[8];
            car_state_to8(s, s8);
#pragma unroll
            for (int i = 0; i < 8; ++i) tr[(size_t)(8 * c + i) * T + t] = s8[i];
        }
    }
    cost += cc;
    const double total = cost;
#ifdef MPOPIS_PATH_STATS
    {   // dev build: rollouts of this launch that took the general sub-step at least once / that end stopped (per launch: flags cleared)
        const size_t t_ = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        const bool went = t_ < sizeof(g_sick) && (g_sick[t_] & 1);
        if (t_ < sizeof(g_sick)) g_sick[t_] = 0;
        const int nw = __popcll(__ballot(went)), ns = __popcll(__ballot(!(s.Vx > 0.5)));
        if (lane == 0) { atomicAdd(&g_path_stats[6], (unsigned long long)nw); atomicAdd(&g_path_stats[7], (unsigned long long)ns); }
    }
#endif
    if (valid) a.cost[(size_t)b * K + k] = total;
    if (a.cmin) {
        // ρ = minimum(costs) (utils.jl:81) accumulates here, one atomic per wave, so that the AIS reweighting can be folded into the moments
        // kernel (launch_wcov_mfma, weights from costs) instead of a launch of its own between the two
        unsigned long long key = valid ? cost_key(total) : ~0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(key, o, 64); key = (t < key) ? t : key; }
        if (lane == 0) atomicMin(&a.cmin[b], key);
        if (valid && !(fabs(total) < INFINITY) && a.status) status_raise(&a.status[b], MPOPIS_ERR_ACTION);   // non-finite cost <=> NaN action (car_racing.jl:239)
    }
#ifdef MPOPIS_ROLL_PROF
    if (lane == 0) {
        const int w = (blockIdx.y * gridDim.x + blockIdx.x) * SPB + wave;
        if (w < 8192) {
            unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_roll_prof[3 * w] = prof_t0; g_roll_prof[3 * w + 1] = __builtin_amdgcn_s_memrealtime(); g_roll_prof[3 * w + 2] = ((unsigned long long)xcc << 32) | hwid;
        }
    }
#endif
}

// One car, few rollouts (up to two rollout waves per SIMD -- one trial of K <= 4096, or the 8 .. 32-trials-per-GPU share of a strong-scaled
// run).  A wave alone on a SIMD issues one FP64 instruction per ~8.4 cycles whatever it does, so a rollout costs its instruction count and most
// of the chip idles.  Here a workgroup is TWO waves for 64 samples: wave 0 integrates the dynamics (V = U + E, clamp, car_action_step) and hands
// (x, y, Vx, Vy) after every model step to wave 1 through a two-slot LDS mailbox; wave 1 evaluates the reward (nearest-point search, lane test,
// drift penalty: ~110 of the ~885 instructions of a model step) and accumulates the cost.  Same arithmetic in the same order as k_rollout_car --
// bit-identical costs -- with the dynamics wave's chain 11 % shorter.  The mailbox is release / acquire at workgroup scope; the reward wave is
// ~8x faster than the dynamics wave, so the producer practically never waits for a free slot.
template <bool TLDS>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) k_rollout_car_duo(RolloutArgs a) {
    const int b = blockIdx.y;
    if (a.active && !a.active[b]) return;
    if (a.iters && blockIdx.x == 0 && threadIdx.x == 0) a.iters[b] = a.iter_n;
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // 0: dynamics, 1: reward
    const int k = blockIdx.x * 64 + lane;
    const int K = a.K, T = a.T;
    const bool valid = k < K;
    const int kk = valid ? k : K - 1;
    constexpr int as = 2;
    const CarParams& p = a.env.car;
    extern __shared__ __attribute__((aligned(16))) double sh_dyn[];
    __shared__ double sh_state[2][4][64];
    __shared__ double sh_cc[64];
    __shared__ int sh_ready, sh_done;                                          // model steps published by wave 0 / consumed by wave 1
    const Track tk = stage_track<TLDS>(a.env.track, sh_dyn, threadIdx.x, 128);
    if (threadIdx.x == 0) { sh_ready = 0; sh_done = 0; }
    __syncthreads();
    if (role == 0) {
        CarState s;
        {
            const double* xe = a.x0ext + (size_t)b * kCarExt;
            s.x = xe[0]; s.y = xe[1]; s.psi = xe[2]; s.Vx = xe[3]; s.Vy = xe[4]; s.r = xe[5]; s.delta = xe[6]; s.pedal = xe[7];
            s.sp = xe[8]; s.cp = xe[9]; s.sd = xe[10]; s.cd = xe[11]; s.near = (int)xe[12];
        }
        const double* Eb = a.E + (size_t)b * a.cs * K + kk;
        const double* Ub = a.Ucur + (size_t)b * a.cs;
        const double* Uo = a.Uorig + (size_t)b * a.cs;
        const double* gv = a.gvec ? a.gvec + (size_t)b * a.cs : nullptr;
        const double lo0 = a.env.lo[0], hi0 = a.env.hi[0], lo1 = a.env.lo[1], hi1 = a.env.hi[1];
        double cc = 0.0;
        double e0 = Eb[0], e1 = Eb[K], u0 = Ub[0], u1 = Ub[1];
        for (int t = 0; t < T; ++t) {
            const double v0 = u0 + e0, v1 = u1 + e1;                           // V = pol.U + E[:,k]  :271
            if (t + 1 < T) {
                e0 = Eb[(size_t)(t + 1) * as * K]; e1 = Eb[(size_t)(t + 1) * as * K + K];
                u0 = Ub[(t + 1) * as]; u1 = Ub[(t + 1) * as + 1];
            }
            if (__builtin_expect(gv != nullptr, 0)) cc += control_cost_term(gv[t * as], v0 - Uo[t * as], gv[t * as + 1], v1 - Uo[t * as + 1]);   // :272
            const double a0 = clampd_u(v0, lo0, hi0), a1 = clampd_u(v1, lo1, hi1);
            car_action_step<false>(p, s, a0, a1, (t & 3) == 0);
            const int slot = t & 1;
            if (t >= 2) while (__hip_atomic_load(&sh_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < t - 1) __builtin_amdgcn_s_sleep(1);   // slot free again?
            sh_state[slot][0][lane] = s.x; sh_state[slot][1][lane] = s.y; sh_state[slot][2][lane] = s.Vx; sh_state[slot][3][lane] = s.Vy;
            __hip_atomic_store(&sh_ready, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        sh_cc[lane] = cc;
        __hip_atomic_store(&sh_ready, T + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    double cost = 0.0;
    int near = (int)a.x0ext[(size_t)b * kCarExt + 12];                         // the anchor of the first search: the track point nearest to the start position
    for (int t = 0; t < T; ++t) {
        while (__hip_atomic_load(&sh_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < t + 1) __builtin_amdgcn_s_sleep(1);
        const int slot = t & 1;
        const double x = sh_state[slot][0][lane], y = sh_state[slot][1][lane], Vx = sh_state[slot][2][lane], Vy = sh_state[slot][3][lane];
        __hip_atomic_store(&sh_done, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);       // (release: the reads above are complete)
        const double rew = car_reward(p, tk, x, y, Vx, Vy, &near);
        cost -= rew;                                                           // utils.jl:138
    }
    while (__hip_atomic_load(&sh_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < T + 1) __builtin_amdgcn_s_sleep(1);
    cost += sh_cc[lane];
    const double total = cost;
    if (valid) a.cost[(size_t)b * K + k] = total;
    if (a.cmin) {
        unsigned long long key = valid ? cost_key(total) : ~0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(key, o, 64); key = (t < key) ? t : key; }
        if (lane == 0) atomicMin(&a.cmin[b], key);
        if (valid && !(fabs(total) < INFINITY) && a.status) status_raise(&a.status[b], MPOPIS_ERR_ACTION);
    }
}

// NC >= 2 cars (MultiCarRacingEnv): lane = c * S + j with S = 64 / NC samples per wave, so every car of a sample lives in the same wave.
// Per-lane (not wave-uniform) here: the start state, the nominal control U (vector loads, one step ahead like E) and the action bounds
// (LDS table).  The pairwise reward terms (multi-car_racing.jl:145-158) read the other cars' (x, y) by lane shuffles; the sample's cost is
// the sum over its cars in car order, gathered the same way.  SPB waves per workgroup share one LDS copy of the track tables.
template <int NC, int SPB, bool LOG, int WPE, bool TLDS>
__global__ void __launch_bounds__(64 * SPB) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_rollout_cars(RolloutArgs a) {
    static_assert(NC >= 2 && NC <= kMaxCars, "2..4 cars");
    constexpr int S = 64 / NC;                                // samples per wave
    const int b = blockIdx.y;
    if (a.active && !a.active[b]) return;
    if (a.iters && blockIdx.x == 0 && threadIdx.x == 0) a.iters[b] = a.iter_n;
    const int lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = min(lane / S, NC - 1), j = lane - c * S;    // car, sample within the wave (NC = 3: lane 63 idles as a duplicate)
    const int k = (blockIdx.x * SPB + g) * S + j;
    const int K = a.K, T = a.T;
    const bool valid = j < S && k < K;
    const int kk = min(k, K - 1);
    constexpr int as = 2 * NC, ss = 8 * NC;
    (void)ss;

    const CarParams& p = a.env.car;
    extern __shared__ __attribute__((aligned(16))) double sh_dyn[];
    __shared__ double sh_bnd[NC][4];                          // lo0, hi0, lo1, hi1 per car
    const Track tk = stage_track<TLDS>(a.env.track, sh_dyn, threadIdx.x, 64 * SPB);
    if (threadIdx.x < NC) {
        const int q = threadIdx.x;
        sh_bnd[q][0] = a.env.lo[2 * q]; sh_bnd[q][1] = a.env.hi[2 * q]; sh_bnd[q][2] = a.env.lo[2 * q + 1]; sh_bnd[q][3] = a.env.hi[2 * q + 1];
    }
    __syncthreads();
    CarState s;
    {
        const double* xe = a.x0ext + ((size_t)b * NC + c) * kCarExt;
        s.x = xe[0]; s.y = xe[1]; s.psi = xe[2]; s.Vx = xe[3]; s.Vy = xe[4]; s.r = xe[5]; s.delta = xe[6]; s.pedal = xe[7];
        s.sp = xe[8]; s.cp = xe[9]; s.sd = xe[10]; s.cd = xe[11]; s.near = (int)xe[12];
    }
    const double* Eb = a.E + (size_t)b * a.cs * K + (size_t)(2 * c) * K + kk;
    const double* Ub = a.Ucur + (size_t)b * a.cs + 2 * c;
    const double* Uo = a.Uorig + (size_t)b * a.cs + 2 * c;
    const double* gv = a.gvec ? a.gvec + (size_t)b * a.cs + 2 * c : nullptr;
    double* tr = LOG ? a.traj + ((size_t)b * K + kk) * (size_t)(ss * T) : nullptr;

    double cost = 0.0, cc = 0.0;
    double e0 = Eb[0], e1 = Eb[K], u0 = Ub[0], u1 = Ub[1];
    for (int t = 0; t < T; ++t) {
        const double v0 = u0 + e0, v1 = u1 + e1;                               // V = pol.U + E[:,k]  :271
        if (t + 1 < T) {                                                       // next step's noise and nominal control in flight during this step
            e0 = Eb[(size_t)(t + 1) * as * K]; e1 = Eb[(size_t)(t + 1) * as * K + K];
            u0 = Ub[(t + 1) * as]; u1 = Ub[(t + 1) * as + 1];
        }
        if (__builtin_expect(gv != nullptr, 0)) cc += control_cost_term(gv[t * as], v0 - Uo[t * as], gv[t * as + 1], v1 - Uo[t * as + 1]);   // :272
        const double a0 = clampd_v(v0, sh_bnd[c][0], sh_bnd[c][1]), a1 = clampd_v(v1, sh_bnd[c][2], sh_bnd[c][3]);   // get_model_controls (NaN passes through)
        car_action_step<LOG>(p, s, a0, a1, (t & 3) == 0);                      // unit-circle renormalisation every 4th step
        double rew = car_reward(p, tk, s.x, s.y, s.Vx, s.Vy, &s.near);
#pragma unroll
        for (int q = 1; q < NC; ++q) {                                         // multi-car_racing.jl:145-158: the cars behind this one in the env's order
            const double xq = __shfl(s.x, q * S + j, 64), yq = __shfl(s.y, q * S + j, 64);
            if (q > c) {
                const double dx = xq - s.x, dy = yq - s.y;
                const double dd = fast_sqrt(fma(dx, dx, dy * dy));             // 1 ulp (car_dynamics.h); coincident cars give 1e-150, not 0
                rew += -dd;
                if (dd <= 4.0) rew += -11000.0;
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
#pragma unroll
    for (int q = 1; q < NC; ++q) total += __shfl(cost, q * S + j, 64);         // meaningful on the car-0 lanes: cost_0 + cost_1 + ...
    const bool writer = valid && c == 0;
    if (writer) a.cost[(size_t)b * K + k] = total;
    if (a.cmin) {
        unsigned long long key = writer ? cost_key(total) : ~0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(key, o, 64); key = (t < key) ? t : key; }
        if (lane == 0) atomicMin(&a.cmin[b], key);
        if (writer && !(fabs(total) < INFINITY) && a.status) status_raise(&a.status[b], MPOPIS_ERR_ACTION);
    }
}

// The two-wave form of k_rollout_cars for few rollout waves (k_rollout_car_duo explains why): wave 0 integrates all cars of its S samples, wave 1 takes
// (x, y, Vx, Vy) of every lane per model step from the LDS mailbox and evaluates the per-car reward AND the pair terms (lane shuffles among
// its own lanes, as in k_rollout_cars), accumulates the cost and gathers the sample's total over its cars.  Same arithmetic, same order.
template <int NC, bool TLDS>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3, 3))) k_rollout_cars_duo(RolloutArgs a) {
    static_assert(NC >= 2 && NC <= kMaxCars, "2..4 cars");
    constexpr int S = 64 / NC;
    const int b = blockIdx.y;
    if (a.active && !a.active[b]) return;
    if (a.iters && blockIdx.x == 0 && threadIdx.x == 0) a.iters[b] = a.iter_n;
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // 0: dynamics, 1: reward
    const int c = min(lane / S, NC - 1), j = lane - c * S;
    const int k = blockIdx.x * S + j;
    const int K = a.K, T = a.T;
    const bool valid = j < S && k < K;
    const int kk = min(k, K - 1);
    constexpr int as = 2 * NC;
    const CarParams& p = a.env.car;
    extern __shared__ __attribute__((aligned(16))) double sh_dyn[];
    __shared__ double sh_bnd[NC][4];
    __shared__ double sh_state[2][4][64];
    __shared__ double sh_cc[64];
    __shared__ int sh_ready, sh_done;
    const Track tk = stage_track<TLDS>(a.env.track, sh_dyn, threadIdx.x, 128);
    if (threadIdx.x < NC) {
        const int q = threadIdx.x;
        sh_bnd[q][0] = a.env.lo[2 * q]; sh_bnd[q][1] = a.env.hi[2 * q]; sh_bnd[q][2] = a.env.lo[2 * q + 1]; sh_bnd[q][3] = a.env.hi[2 * q + 1];
    }
    if (threadIdx.x == 0) { sh_ready = 0; sh_done = 0; }
    __syncthreads();
    if (role == 0) {
        CarState s;
        {
            const double* xe = a.x0ext + ((size_t)b * NC + c) * kCarExt;
            s.x = xe[0]; s.y = xe[1]; s.psi = xe[2]; s.Vx = xe[3]; s.Vy = xe[4]; s.r = xe[5]; s.delta = xe[6]; s.pedal = xe[7];
            s.sp = xe[8]; s.cp = xe[9]; s.sd = xe[10]; s.cd = xe[11]; s.near = (int)xe[12];
        }
        const double* Eb = a.E + (size_t)b * a.cs * K + (size_t)(2 * c) * K + kk;
        const double* Ub = a.Ucur + (size_t)b * a.cs + 2 * c;
        const double* Uo = a.Uorig + (size_t)b * a.cs + 2 * c;
        const double* gv = a.gvec ? a.gvec + (size_t)b * a.cs + 2 * c : nullptr;
        double cc = 0.0;
        double e0 = Eb[0], e1 = Eb[K], u0 = Ub[0], u1 = Ub[1];
        for (int t = 0; t < T; ++t) {
            const double v0 = u0 + e0, v1 = u1 + e1;                           // V = pol.U + E[:,k]  :271
            if (t + 1 < T) {
                e0 = Eb[(size_t)(t + 1) * as * K]; e1 = Eb[(size_t)(t + 1) * as * K + K];
                u0 = Ub[(t + 1) * as]; u1 = Ub[(t + 1) * as + 1];
            }
            if (__builtin_expect(gv != nullptr, 0)) cc += control_cost_term(gv[t * as], v0 - Uo[t * as], gv[t * as + 1], v1 - Uo[t * as + 1]);   // :272
            const double a0 = clampd_v(v0, sh_bnd[c][0], sh_bnd[c][1]), a1 = clampd_v(v1, sh_bnd[c][2], sh_bnd[c][3]);
            car_action_step<false>(p, s, a0, a1, (t & 3) == 0);
            const int slot = t & 1;
            if (t >= 2) while (__hip_atomic_load(&sh_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < t - 1) __builtin_amdgcn_s_sleep(1);
            sh_state[slot][0][lane] = s.x; sh_state[slot][1][lane] = s.y; sh_state[slot][2][lane] = s.Vx; sh_state[slot][3][lane] = s.Vy;
            __hip_atomic_store(&sh_ready, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        sh_cc[lane] = cc;
        __hip_atomic_store(&sh_ready, T + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    double cost = 0.0;
    int near = (int)a.x0ext[((size_t)b * NC + c) * kCarExt + 12];
    for (int t = 0; t < T; ++t) {
        while (__hip_atomic_load(&sh_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < t + 1) __builtin_amdgcn_s_sleep(1);
        const int slot = t & 1;
        const double x = sh_state[slot][0][lane], y = sh_state[slot][1][lane], Vx = sh_state[slot][2][lane], Vy = sh_state[slot][3][lane];
        __hip_atomic_store(&sh_done, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        double rew = car_reward(p, tk, x, y, Vx, Vy, &near);
#pragma unroll
        for (int q = 1; q < NC; ++q) {                                         // multi-car_racing.jl:145-158
            const double xq = __shfl(x, q * S + j, 64), yq = __shfl(y, q * S + j, 64);
            if (q > c) {
                const double dx = xq - x, dy = yq - y;
                const double dd = fast_sqrt(fma(dx, dx, dy * dy));
                rew += -dd;
                if (dd <= 4.0) rew += -11000.0;
            }
        }
        cost -= rew;                                                           // utils.jl:138
    }
    while (__hip_atomic_load(&sh_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < T + 1) __builtin_amdgcn_s_sleep(1);
    cost += sh_cc[lane];
    double total = cost;
#pragma unroll
    for (int q = 1; q < NC; ++q) total += __shfl(cost, q * S + j, 64);
    const bool writer = valid && c == 0;
    if (writer) a.cost[(size_t)b * K + k] = total;
    if (a.cmin) {
        unsigned long long key = writer ? cost_key(total) : ~0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(key, o, 64); key = (t < key) ? t : key; }
        if (lane == 0) atomicMin(&a.cmin[b], key);
        if (writer && !(fabs(total) < INFINITY) && a.status) status_raise(&a.status[b], MPOPIS_ERR_ACTION);
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
        // a NaN action: RL.jl's act! asserts `a in action_space(env)` and the rollout dies there.  MountainCar's reward carries the NaN state into the cost by
        // itself; CartPole's reward (1 until done) does not look at the state's values, so the cost is poisoned explicitly -- a non-finite cost is what raises
        // MPOPIS_ERR_ACTION (round 6, tests/test_gpu_edge_cases.py)
        if (act != act) cost = act;
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
// ... and the track point nearest to the start position (full scan, once per trial and car): the anchor of every rollout's first nearest-point
// search, which then takes the ring paths like every later step instead of sending the whole wave through the general search
__device__ __forceinline__ void write_xext(const double* x8, double* o, const Track& tk) {
    CarState c;
    car_state_from8(c, x8);
    o[0] = c.x; o[1] = c.y; o[2] = c.psi; o[3] = c.Vx; o[4] = c.Vy; o[5] = c.r; o[6] = c.delta; o[7] = c.pedal;
    o[8] = c.sp; o[9] = c.cp; o[10] = c.sd; o[11] = c.cd;
    int near = -1;
    if (tk.P > 0) { double dist; (void)within_track(tk, c.x, c.y, &dist, &near); }
    o[12] = (double)near;
}
__global__ void k_extend_state(const double* x, double* xext, int n_cars_total, Track tk) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_cars_total) return;
    write_xext(x + (size_t)i * 8, xext + (size_t)i * kCarExt, tk);
}
void launch_extend_state(const double* x, double* xext, int B, int ncars, hipStream_t st, const Track& tk) {
    const int n = B * ncars;
    hipLaunchKernelGGL(k_extend_state, dim3((n + 63) / 64), dim3(64), 0, st, x, xext, n, tk);
}

// Start of an MPC step, one launch instead of seven (at one trial the step is a chain of dependent launches, each boundary costs 2-4 us):
// status = 0 (unless sticky), active = alive gate (or 1), iters = 0, U_orig = the loop's pol.U = pol.U, and the car start states extended
// with sin/cos of psi / delta (as k_extend_state).
__global__ void __launch_bounds__(256) k_step_begin(int* status, int* active, const int* alive, int* iters, const double* U, double* Uin, double* Ucur,
                                                    int cs, const double* x, double* xext, int ncars, unsigned long long* cmin, Track tk, unsigned long long* iters_acc) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        if (iters_acc) iters_acc[b] += (unsigned long long)iters[b];          // the previous step's executed iterations (CE / CMA early breaks counted as run)
        if (status) status[b] = 0; active[b] = alive ? alive[b] : 1; iters[b] = 0; if (cmin) cmin[b] = ~0ull;
    }
    for (int i = tid; i < cs; i += 256) { const double u = U[(size_t)b * cs + i]; Uin[(size_t)b * cs + i] = u; Ucur[(size_t)b * cs + i] = u; }
    if (!x) return;
    if (tid < ncars) {
        const size_t i = (size_t)b * ncars + tid;
        write_xext(x + i * 8, xext + i * kCarExt, Track{nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr});      // (nearest point: below)
    }
    // the track point nearest to each car's start position -- the first minimum of |q_i|^2 - 2 q_i.p, exactly what within_track's full scan
    // returns -- found by the whole workgroup (one thread scanning the track serially put 3-4 us on the critical path of EVERY MPC step: C2 0.130 -> 0.134 ms)
    __shared__ double sh_v[4]; __shared__ int sh_i[4];
    for (int c = 0; c < ncars; ++c) {
        const double* s8 = x + ((size_t)b * ncars + c) * 8;
        const double m2x = -2.0 * s8[0], m2y = -2.0 * s8[1];
        double bv = 0.0; int bi = -1;
        for (int i = tid; i < tk.P; i += 256) {
            const double d = track_key(tk.x, tk.y, tk.n2, i, m2x, m2y);        // the same key and order as within_track (car_dynamics.h)
            if (bi < 0 || track_key_before(d, i, bv, bi)) { bv = d; bi = i; }  // ascending i within a thread: its first minimum
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (oi >= 0 && (bi < 0 || track_key_before(ov, oi, bv, bi))) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { sh_v[tid >> 6] = bv; sh_i[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) if (sh_i[w] >= 0 && (bi < 0 || track_key_before(sh_v[w], sh_i[w], bv, bi))) { bv = sh_v[w]; bi = sh_i[w]; }
            xext[((size_t)b * ncars + c) * kCarExt + 12] = (double)bi;
        }
    }
}
void launch_step_begin(int* status, int* active, const int* alive, int* iters, const double* U, double* Uin, double* Ucur, int B, int cs,
                       const double* x, double* xext, int ncars, hipStream_t st, unsigned long long* cmin, const Track& tk, unsigned long long* iters_acc) {
    hipLaunchKernelGGL(k_step_begin, dim3(B), dim3(256), 0, st, status, active, alive, iters, U, Uin, Ucur, cs, x, xext, ncars, cmin, tk, iters_acc);
}

// Dynamic LDS a rollout kernel may request without raising its limit: the default 64 KB minus the kernels' STATIC LDS (two-wave kernels:
// mailbox 4 KB + control-cost column 0.5 KB + flags + per-car bounds, ~4.7 KB).  Beyond it the limit is raised before the launch, once per
// kernel and device, to what the largest supported track needs: the ring-only layout at kMaxTrackPoints (48 P + 192 B = 98 496 B at P = 2048)
// plus the static part, rounded up -- 112 KB of the CU's 160 KB.
constexpr size_t kRolloutDynLdsDefault = 56 * 1024;
constexpr int kRolloutDynLdsRaised = 112 * 1024;
static_assert((size_t)(4 * (kMaxTrackPoints + 2 * kRingPad) + 2 * kMaxTrackPoints) * sizeof(double) + 8 * 1024 <= (size_t)kRolloutDynLdsRaised,
              "ring table of the largest track + static LDS must fit the raised limit");
// one `seen` mask per kernel (non-type template parameter): large tracks need the dynamic-LDS limit raised, per device
template <void (*KERNEL)(RolloutArgs)>
static void launch_rollout_kernel(dim3 grid, int block, size_t lds, hipStream_t st, const RolloutArgs& a) {
    static std::atomic<unsigned long long> seen{0};
    if (lds > kRolloutDynLdsDefault) ensure_dyn_lds((const void*)KERNEL, kRolloutDynLdsRaised, seen);
    hipLaunchKernelGGL(KERNEL, grid, dim3(block), lds, st, a);
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
    // every table in LDS when that fits the default 64 KB (all bundled tracks: 48-60 points); larger tracks (Track(infile; sample_factor = 1):
    // ~1000 points) stage the ring table only (48 P + 192 B: 96.2 KB at the 2048-point limit; the kernels' dynamic-LDS limit is raised to 112 KB once)
    const bool tl = track_lds_bytes(P, W, true) <= kRolloutDynLdsDefault;      // (static LDS of the two-wave kernels counted: P = 221, 222 used to total 65.7-66 KB without the limit being raised)
    const size_t lds = track_lds_bytes(P, W, tl);
    // small K: one sample-wave per workgroup keeps every wave on its own CU; large K: 4 waves share the LDS tables
    const bool wide = a.K >= 1024;
    const dim3 g1((a.K + 63) / 64, a.B), g4((a.K + 255) / 256, a.B);
#define MPOPIS_LAUNCH_K(KERNEL, GRID, BLOCK) launch_rollout_kernel<KERNEL>(GRID, BLOCK, lds, st, a)
#define MPOPIS_LAUNCH_CAR(NC, SPB, GRID, BLOCK)                                                        \
    do {                                                                                               \
        if (a.traj) { if (tl) MPOPIS_LAUNCH_K((k_rollout_car<NC, SPB, true, true>), GRID, BLOCK); else MPOPIS_LAUNCH_K((k_rollout_car<NC, SPB, true, false>), GRID, BLOCK); }   \
        else        { if (tl) MPOPIS_LAUNCH_K((k_rollout_car<NC, SPB, false, true>), GRID, BLOCK); else MPOPIS_LAUNCH_K((k_rollout_car<NC, SPB, false, false>), GRID, BLOCK); } \
    } while (0)
#define MPOPIS_LAUNCH_CARS_W(NC, WPE, SPB, GRID, BLOCK)                                                                                                      \
    do {                                                                                                                                                    \
        if (a.traj) { if (tl) MPOPIS_LAUNCH_K((k_rollout_cars<NC, SPB, true, WPE, true>), GRID, BLOCK); else MPOPIS_LAUNCH_K((k_rollout_cars<NC, SPB, true, WPE, false>), GRID, BLOCK); }   \
        else        { if (tl) MPOPIS_LAUNCH_K((k_rollout_cars<NC, SPB, false, WPE, true>), GRID, BLOCK); else MPOPIS_LAUNCH_K((k_rollout_cars<NC, SPB, false, WPE, false>), GRID, BLOCK); } \
    } while (0)
    // waves per SIMD of the multi-car kernel: 3 (168 VGPRs, no spills; 1411 us at 64 three-car trials) beats 4 (128 VGPRs, 33 spills: 1509 us)
#define MPOPIS_LAUNCH_CARS(NC)                                                                                   \
    do {                                                                                                         \
        const int S_ = 64 / NC;                                                                                  \
        const dim3 gw((a.K + 4 * S_ - 1) / (4 * S_), a.B), gn((a.K + S_ - 1) / S_, a.B);                          \
        const long long waves_ = (long long)a.B * ((a.K + S_ - 1) / S_) * std::max(1, a.share);                  \
        if (!a.traj && waves_ <= (env_duo >= 0 ? env_duo : kDuoCarsWaves * coop_max_workgroups())) {             \
            if (tl) MPOPIS_LAUNCH_K((k_rollout_cars_duo<NC, true>), gn, 128); else MPOPIS_LAUNCH_K((k_rollout_cars_duo<NC, false>), gn, 128); \
        } else if (wide) MPOPIS_LAUNCH_CARS_W(NC, 3, 4, gw, 256); else MPOPIS_LAUNCH_CARS_W(NC, 3, 1, gn, 64);    \
    } while (0)
    // two-wave kernels: MPOPIS_ROLLOUT_DUO = largest rollout-wave count (all parts of a multi-stream schedule together) that still takes them; 0: never
    static const int env_duo = [] { const char* e = getenv("MPOPIS_ROLLOUT_DUO"); return e ? atoi(e) : -1; }();
    constexpr int kDuoCarsWaves = 5;                           // multi-car (3 waves per SIMD): measured, C4 shapes -- 5.53 -> 4.86 ms at one trial, 7.18 -> 6.66 at 6 (1176 waves), 7.49 -> 7.68 at 8
    switch (a.env.ncars) {
        case 1: {
            // up to 2 rollout waves per SIMD (measured crossover, C5 shapes: 3.49 vs 3.95 ms at 32 trials, 5.36 vs 4.71 at 48; the duo kernel is held to
            // 128 VGPRs = 4 waves per SIMD, so two dynamics and two reward waves share a SIMD there):
            // dynamics and reward in two waves per 64 samples (k_rollout_car_duo); MPOPIS_ROLLOUT_DUO=0: never
            const long long waves = (long long)a.B * ((a.K + 63) / 64) * std::max(1, a.share);
            if (!a.traj && waves <= (env_duo >= 0 ? env_duo : 8 * coop_max_workgroups())) {
                if (tl) MPOPIS_LAUNCH_K((k_rollout_car_duo<true>), g1, 128); else MPOPIS_LAUNCH_K((k_rollout_car_duo<false>), g1, 128);
            } else if (wide) MPOPIS_LAUNCH_CAR(1, 4, g4, 256); else MPOPIS_LAUNCH_CAR(1, 1, g1, 64);
            break;
        }
        case 2: MPOPIS_LAUNCH_CARS(2); break;
        case 3: MPOPIS_LAUNCH_CARS(3); break;
        case 4: MPOPIS_LAUNCH_CARS(4); break;
        default: break;
    }
#undef MPOPIS_LAUNCH_CARS
#undef MPOPIS_LAUNCH_CARS_W
#undef MPOPIS_LAUNCH_K
#undef MPOPIS_LAUNCH_CAR
}

}  // namespace mpopis
