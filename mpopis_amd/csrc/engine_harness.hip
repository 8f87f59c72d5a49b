// engine_harness.hip -- Level 3: the closed-loop trial harness, resident on the device.
//   simulate_car_racing trial loop   src/examples/car_example.jl:170-326
//   simulate_mountaincar trial loop  src/examples/mountaincar_example.jl:125-180
//   simulate_cartpole trial loop     src/examples/cartpole_example.jl:110-160
// All B trial slots advance together: policy step -> real env step -> bookkeeping, with no host
// round trip per MPC step (the host only polls the alive flags every few steps to stop early).
// A slot that terminates (laps done, > 10 track violations, > 50 β violations, MountainCar done, or
// cnt > num_steps) is frozen: its `alive` flag gates every kernel of later steps.
#include "engine.h"
#include "engine_handle.h"
#include "philox.h"
#include <math.h>
#include <algorithm>

using namespace mpopis;

namespace mpopis {

// per-slot accumulator (doubles): see kH_* indices
enum { kH_rew = 0, kH_cnt, kH_lap, kH_prev_y, kH_trk, kH_beta, kH_crash, kH_vmean, kH_vmax, kH_bmean, kH_bmax,
       kH_lap0, kH_lap1, kH_lap2, kH_lap3, kH_rollouts, kH_N = 16 };

__global__ void k_harness_init(double* hs, int* alive, int B) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    for (int i = 0; i < kH_N; ++i) hs[(size_t)b * kH_N + i] = 0.0;
    hs[(size_t)b * kH_N + kH_vmax] = -INFINITY; hs[(size_t)b * kH_N + kH_bmax] = -INFINITY;
    alive[b] = 1;
}

// after env(act): cnt += 1; rew += reward(env); logging; violations; lap counter; termination
struct StateNoise { double sx, sy, spsi; const uint64_t* seeds; const double* rng_tab; };
__global__ void k_harness_update(EnvDesc env, double* x, const int* done_env, const double* reward, const int* iters,
                                 double* hs, int* alive, const double* control, double* actlog, int step, int num_steps, int laps, int K, int B,
                                 StateNoise nz) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B || !alive[b]) return;
    double* h = hs + (size_t)b * kH_N;
    if (nz.seeds && env.kind == MPOPIS_ENV_CAR && env.ncars == 1) {            // :224-236 (sim_type == :cr only), after reward(env)
        double z0, z1, z2, z3;
        philox_normal_pair(nz.seeds[b], (uint32_t)step, 0x40000000u, 0, nz.rng_tab, &z0, &z1);      // (tables read from global memory: a few lanes per step)
        philox_normal_pair(nz.seeds[b], (uint32_t)step, 0x40000000u, 1, nz.rng_tab, &z2, &z3);
        double* s = x + (size_t)b * 8;
        s[0] += nz.sx * z0; s[1] += nz.sy * z1;
        const double dpsi = nz.spsi * z2;
        s[2] += dpsi;
        const double c = cos(dpsi), sn = sin(dpsi), vx = s[3], vy = s[4];      // passive rotation matrix [c s; -s c]
        s[3] = c * vx + sn * vy; s[4] = -sn * vx + c * vy;
    }
    if (actlog) for (int i = 0; i < env.as; ++i) actlog[((size_t)b * (num_steps + 1) + step) * env.as + i] = control[(size_t)b * env.as + i];
    h[kH_rollouts] += (double)iters[b] * K;
    const double step_rew = reward[b];
    h[kH_rew] += step_rew;                                                     // :210-211
    const double cnt = h[kH_cnt] + 1.0;                                        // :208
    h[kH_cnt] = cnt;
    bool done = false;
    if (env.kind != MPOPIS_ENV_CAR) {
        done = done_env[b] != 0;                                               // env.done from RL.jl _step!
    } else {
        const int NC = env.ncars;
        const double* xb = x + (size_t)b * 8 * NC;
        double curr_y = xb[1], vmean = 0, vmax = -INFINITY, bmean = 0, bmax = -INFINITY, d = INFINITY;   // :241-253,:265-270
        bool ex_b = false, within_t = true;
        for (int c = 0; c < NC; ++c) {
            const double* s = xb + 8 * c;
            if (s[1] < curr_y) curr_y = s[1];
            const double v = sqrt(s[3] * s[3] + s[4] * s[4]);
            const double be = fabs(atan2(s[4], s[3]));
            vmean += v; bmean += be; vmax = fmax(vmax, v); bmax = fmax(bmax, be);
            d = fmin(d, sqrt(s[0] * s[0] + s[1] * s[1]));
            if (be > env.car.blim) ex_b = true;
            double dist;
            if (!within_track(env.track, s[0], s[1], &dist, nullptr)) within_t = false;
        }
        h[kH_vmean] += vmean / NC; h[kH_bmean] += bmean / NC;
        h[kH_vmax] = fmax(h[kH_vmax], vmax); h[kH_bmax] = fmax(h[kH_bmax], bmax);
        if (step_rew < -4000) {                                                // :256-263
            if (ex_b) h[kH_beta] += 1;
            if (!within_t) h[kH_trk] += 1;
            const double temp_rew = step_rew + (ex_b ? 5000.0 : 0.0) + (!within_t ? 1000000.0 : 0.0);
            if (temp_rew < -10500) h[kH_crash] += 1;
        }
        if (h[kH_prev_y] < 0.0 && curr_y >= 0.0 && d <= 15.0) {                // :273-276
            h[kH_lap] += 1;
            const int lap = (int)h[kH_lap];
            if (lap >= 1 && lap <= 4) h[kH_lap0 + lap - 1] = cnt;
        }
        if (h[kH_lap] >= laps || h[kH_trk] > 10 || h[kH_beta] > 50) done = true;   // :277-279
        h[kH_prev_y] = curr_y;
    }
    if (done || !(cnt <= num_steps)) alive[b] = 0;                              // while !env.done && cnt <= num_steps
}

__global__ void k_zero_status(int* st, int B) { const int b = blockIdx.x * 64 + threadIdx.x; if (b < B) st[b] = 0; }

}  // namespace mpopis

int mpopis_handle::run_trials(int num_steps, int laps, double* records, double* actions) {
    if (env.kind == MPOPIS_ENV_CAR && env.track.P == 0) { err = "track not set"; return MPOPIS_ERR_ARG; }
    if (num_steps < 0 || laps < 0 || laps > 4) { err = "need num_steps >= 0 and 0 <= laps <= 4"; return MPOPIS_ERR_ARG; }
    if (hipSetDevice(cfg.device) != hipSuccess) { err = "hipSetDevice failed"; return MPOPIS_ERR_HIP; }
    if (!d_hs) {
        if (hipMalloc((void**)&d_hs, sizeof(double) * kH_N * B) != hipSuccess || hipMalloc((void**)&d_alive, sizeof(int) * B) != hipSuccess) { err = "hipMalloc failed"; return MPOPIS_ERR_HIP; }
        allocs.push_back(d_hs); allocs.push_back(d_alive);
    }
    double* d_actlog = nullptr;
    if (actions) {
        if (hipMalloc((void**)&d_actlog, sizeof(double) * (size_t)B * (num_steps + 1) * as) != hipSuccess) { err = "hipMalloc failed"; return MPOPIS_ERR_HIP; }
        (void)hipMemsetAsync(d_actlog, 0, sizeof(double) * (size_t)B * (num_steps + 1) * as, stream);
    }
    hipLaunchKernelGGL(k_harness_init, dim3((B + 63) / 64), dim3(64), 0, stream, d_hs, d_alive, B);
    hipLaunchKernelGGL(k_zero_status, dim3((B + 63) / 64), dim3(64), 0, stream, d_status, B);
    mpc_step = 0;
    status_sticky = true;                              // errors of any MPC step survive to the end of the call
    std::vector<int> h_alive(B, 1);
    int worst = 0;
    for (int s = 0; s <= num_steps; ++s) {
        alive_gate = d_alive;
        int rc = policy_step_enqueue(false);
        alive_gate = nullptr;
        if (rc) { status_sticky = false; if (d_actlog) (void)hipFree(d_actlog); return rc; }
        launch_env_step(env, d_x, d_t, d_done, d_control, d_reward, d_status, d_alive, B, stream);
        const bool noisy = noise_sx != 0.0 || noise_sy != 0.0 || noise_spsi != 0.0;
        hipLaunchKernelGGL(k_harness_update, dim3((B + 63) / 64), dim3(64), 0, stream, env, d_x, d_done, d_reward, d_iters, d_hs, d_alive,
                           d_control, d_actlog, s, num_steps, laps, K, B,
                           StateNoise{noise_sx, noise_sy, noise_spsi, noisy ? d_seeds : (const uint64_t*)nullptr, d_rng_tab});
        // error status is sticky per call of policy_step_enqueue (it clears d_status): fold it into the host view now and then
        if ((s & 7) == 7 || s == num_steps) {
            (void)hipMemcpyAsync(h_alive.data(), d_alive, sizeof(int) * B, hipMemcpyDeviceToHost, stream);
            (void)hipMemcpyAsync(h_status.data(), d_status, sizeof(int) * B, hipMemcpyDeviceToHost, stream);
            if (h_coop_timeouts && !coop_disabled) (void)hipMemcpyAsync(h_coop_timeouts, d_coop_timeouts, sizeof(int), hipMemcpyDeviceToHost, stream);
            if (hipStreamSynchronize(stream) != hipSuccess) { err = "stream sync failed"; if (d_actlog) (void)hipFree(d_actlog); return MPOPIS_ERR_HIP; }
            if (h_coop_timeouts && *h_coop_timeouts > 0) coop_disabled = true;   // a cluster gave up (and was redone): stop using clusters, also within this call
            for (int b = 0; b < B; ++b) worst = mpopis::worse_status(worst, h_status[b]);
            bool any = false;
            for (int b = 0; b < B; ++b) any |= h_alive[b] != 0;
            if (!any || worst) break;
        }
    }
    status_sticky = false;
    std::vector<double> hs((size_t)B * kH_N);
    if (hipMemcpyAsync(hs.data(), d_hs, sizeof(double) * hs.size(), hipMemcpyDeviceToHost, stream) != hipSuccess) { err = "copy failed"; return MPOPIS_ERR_HIP; }
    if (actions) (void)hipMemcpyAsync(actions, d_actlog, sizeof(double) * (size_t)B * (num_steps + 1) * as, hipMemcpyDeviceToHost, stream);
    if (hipStreamSynchronize(stream) != hipSuccess) { err = "stream sync failed"; return MPOPIS_ERR_HIP; }
    if (d_actlog) (void)hipFree(d_actlog);
    for (int b = 0; b < B; ++b) {                      // car_example.jl:287-302
        const double* h = hs.data() + (size_t)b * kH_N;
        double* r = records + (size_t)b * MPOPIS_RECORD_LEN;
        const double cnt = h[kH_cnt];
        r[0] = h[kH_rew]; r[1] = cnt - 1; r[2] = h[kH_rew] / (cnt - 1);
        r[3] = h[kH_lap0]; r[4] = h[kH_lap1]; r[5] = h[kH_lap2]; r[6] = h[kH_lap3];
        const bool car = env.kind == MPOPIS_ENV_CAR;
        r[7] = car && cnt > 0 ? h[kH_vmean] / cnt : 0.0; r[8] = car ? h[kH_vmax] : 0.0;
        r[9] = car && cnt > 0 ? h[kH_bmean] / cnt : 0.0; r[10] = car ? h[kH_bmax] : 0.0;
        r[11] = h[kH_beta]; r[12] = h[kH_trk]; r[13] = h[kH_crash]; r[14] = h[kH_rollouts]; r[15] = (double)worst;
    }
    if (worst == MPOPIS_ERR_NOT_PD) err = "PosDefException: proposal covariance is not positive definite";
    else if (worst == MPOPIS_ERR_ACTION) err = "Action is not in action space (non-finite control/cost)";
    else if (worst == MPOPIS_ERR_NUMERIC) err = "cmamppi: Σ^-0.5 δw could not be formed (non-finite covariance, trace or δw)";
    return worst;
}
