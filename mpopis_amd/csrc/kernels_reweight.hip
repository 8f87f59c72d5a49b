// kernels_reweight.hip -- softmax re-weighting and the weighted control update:
//   compute_weights(::Information_Theoretic, costs)       src/utils.jl:79-86
//   weighted_noise[r] = weights' * E[r, :]                src/mppi_mpopi_policies.jl:226-229 (:131-136 for :mppi)
//   StatsBase mean(E, pw, dims=2) of the AIS mean update  :364,:662,:732
//   get_controls_roll_U!                                  src/utils.jl:88-101
//   E .+= (pol.U - U_orig)                                :370,:468,:602,:668,:739,:814 (folded into consumers)
// plus the layout converters of the C ABI and the real-env step.
// All of these are HBM/L2-bound reductions over k: wave64 shuffles + one LDS hop per workgroup.
#include "engine.h"

namespace mpopis {

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// all threads get the result; sh must hold >= blockDim/64 doubles; safe to call repeatedly
template <bool IS_MIN>
__device__ __forceinline__ double block_reduce(double v, double* sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = IS_MIN ? wave_min(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = (lane < nw) ? sh[lane] : (IS_MIN ? INFINITY : 0.0);
    r = IS_MIN ? wave_min(r) : wave_sum(r);
    return r;
}

__global__ void __launch_bounds__(1024) k_weights(const double* __restrict__ cost, double* __restrict__ w, int K,
                                                  double neg_inv_lambda, const int* active, int* status, double* __restrict__ wsum) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    __shared__ double sh[16];
    const double* c = cost + (size_t)b * K;
    double* wo = w + (size_t)b * K;
    double m = INFINITY;
    bool bad = false;
    if (K <= 8 * (int)blockDim.x) {
        // the slot's costs fit the workgroup's registers (<= 8 per thread): one read of the costs, one write of the weights, instead of three
        // passes through memory with a dependent round trip each; same per-thread summation order as the loops below (bit-identical)
        double cv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = threadIdx.x + u * blockDim.x; cv[u] = (k < K) ? c[k] : INFINITY; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = threadIdx.x + u * blockDim.x; if (k < K) { m = fmin(m, cv[u]); bad |= !(fabs(cv[u]) < INFINITY); } }
        m = block_reduce<true>(m, sh);                                         // ρ = minimum(costs)
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = threadIdx.x + u * blockDim.x; if (k < K) { cv[u] = exp(neg_inv_lambda * (cv[u] - m)); s += cv[u]; } }
        s = block_reduce<false>(s, sh);                                        // η
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = threadIdx.x + u * blockDim.x; if (k < K) { const double v = cv[u] / s; wo[k] = v; t += v; } }
        if (wsum) {
            t = block_reduce<false>(t, sh);
            if (threadIdx.x == 0) wsum[b] = t;
        }
        if (bad && status) status_raise(&status[b], MPOPIS_ERR_ACTION);
        return;
    }
    for (int k = threadIdx.x; k < K; k += blockDim.x) { const double v = c[k]; m = fmin(m, v); bad |= !(fabs(v) < INFINITY); }
    m = block_reduce<true>(m, sh);                                             // ρ = minimum(costs)
    double s = 0.0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const double e = exp(neg_inv_lambda * (c[k] - m));                     // exp(-1/λ * (c - ρ))
        wo[k] = e; s += e;
    }
    s = block_reduce<false>(s, sh);                                            // η
    double t = 0.0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) { const double v = wo[k] / s; wo[k] = v; t += v; }
    if (wsum) {                                                                // Σ_k w_k of the normalised weights (≈ 1), for the scatter finish
        t = block_reduce<false>(t, sh);
        if (threadIdx.x == 0) wsum[b] = t;
    }
    if (bad && status) status_raise(&status[b], MPOPIS_ERR_ACTION);               // non-finite cost <=> NaN action (car_racing.jl:239)
}

void launch_weights(const double* cost, double* w, int B, int K, double lambda, const int* active, int* status, hipStream_t s, double* wsum) {
    hipLaunchKernelGGL(k_weights, dim3(B), dim3(K >= 1024 ? 1024 : 256), 0, s, cost, w, K, -1 / lambda, active, status, wsum);
}

__global__ void __launch_bounds__(256) k_wmean(const double* __restrict__ E, const double* __restrict__ w,
                                               const double* shiftA, const double* shiftB, double* __restrict__ out,
                                               int cs, int K, int normalize, const int* active) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.y, r = blockIdx.x;
    if (active && !active[b]) return;
    __shared__ double sh[4];
    const double* e = E + ((size_t)b * cs + r) * K;
    const double* wb = w + (size_t)b * K;
    const double sft = shiftA ? (shiftA[(size_t)b * cs + r] - shiftB[(size_t)b * cs + r]) : 0.0;
    double acc = 0.0, ws = 0.0;
    if ((K & 1) == 0) {                                                        // 16 B per lane
        const double2* e2 = reinterpret_cast<const double2*>(e);
        const double2* w2 = reinterpret_cast<const double2*>(wb);
        for (int k = threadIdx.x; k < K / 2; k += 256) {
            const double2 ev = e2[k], wv = w2[k];
            acc = fma(wv.x, ev.x + sft, acc); acc = fma(wv.y, ev.y + sft, acc);
            ws += wv.x + wv.y;
        }
    } else {
        for (int k = threadIdx.x; k < K; k += 256) { acc = fma(wb[k], e[k] + sft, acc); ws += wb[k]; }
    }
    acc = block_reduce<false>(acc, sh);
    if (normalize) { ws = block_reduce<false>(ws, sh); acc = acc / ws; }
    if (threadIdx.x == 0) out[(size_t)b * cs + r] = acc;
}

void launch_wmean(const double* E, const double* w, const double* shiftA, const double* shiftB, double* out,
                  int B, int cs, int K, int normalize, const int* active, hipStream_t s) {
    hipLaunchKernelGGL(k_wmean, dim3(cs, B), dim3(256), 0, s, E, w, shiftA, shiftB, out, cs, K, normalize, active);
}

// weighted_controls = pol.U + weighted_noise; control = clamp(wc[1:as]); roll (alias quirk: the last
// `as` entries of U never change)
__global__ void __launch_bounds__(256) k_finalize(const double* __restrict__ wn, double* U, double* control,
                                                  int cs, int as, int T, EnvDesc env) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) double wc[];
    const int b = blockIdx.x;
    double* Ub = U + (size_t)b * cs;
    for (int r = threadIdx.x; r < cs; r += blockDim.x) wc[r] = Ub[r] + wn[(size_t)b * cs + r];
    __syncthreads();
    for (int i = threadIdx.x; i < as; i += blockDim.x) control[(size_t)b * as + i] = clampd(wc[i], env.lo[i], env.hi[i]);
    if (T > 1) { for (int r = threadIdx.x; r < cs - as; r += blockDim.x) Ub[r] = wc[r + as]; }
    else       { for (int r = threadIdx.x; r < cs; r += blockDim.x) Ub[r] = wc[r]; }
}

void launch_finalize_env(const double* wn, double* U, double* control, int B, int cs, int as, int T,
                         const EnvDesc& env, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize, dim3(B), dim3(256), cs * sizeof(double), s, wn, U, control, cs, as, T, env);
}

// cs x K column-major (ABI / Julia)  <->  [cs][K] rows (engine)
__global__ void __launch_bounds__(256) k_transpose_in(const double* __restrict__ src, size_t src_stride, double* __restrict__ dst, int cs, int K) {
    __shared__ double tile[32][33];
    const int b = blockIdx.z;
    const double* s = src + (size_t)b * src_stride;
    double* d = dst + (size_t)b * cs * K;
    const int r0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                    // 32 x 8
    for (int j = ty; j < 32; j += 8) { const int k = k0 + j, r = r0 + tx; if (k < K && r < cs) tile[j][tx] = s[(size_t)k * cs + r]; }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) { const int r = r0 + j, k = k0 + tx; if (k < K && r < cs) d[(size_t)r * K + k] = tile[tx][j]; }
}
__global__ void __launch_bounds__(256) k_transpose_out(const double* __restrict__ src, const double* shiftA, const double* shiftB,
                                                       double* __restrict__ dst, int cs, int K) {
    __shared__ double tile[32][33];
    const int b = blockIdx.z;
    const double* s = src + (size_t)b * cs * K;
    double* d = dst + (size_t)b * cs * K;
    const int r0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, k = k0 + tx;
        if (k < K && r < cs) tile[j][tx] = s[(size_t)r * K + k] + (shiftA ? (shiftA[(size_t)b * cs + r] - shiftB[(size_t)b * cs + r]) : 0.0);
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) { const int k = k0 + j, r = r0 + tx; if (k < K && r < cs) d[(size_t)k * cs + r] = tile[tx][j]; }
}
void launch_transpose_in(const double* src, double* dst, int B, int cs, int K, hipStream_t s, size_t src_stride) {
    hipLaunchKernelGGL(k_transpose_in, dim3((cs + 31) / 32, (K + 31) / 32, B), dim3(256), 0, s, src, src_stride ? src_stride : (size_t)cs * K, dst, cs, K);
}
void launch_transpose_out(const double* src, const double* shiftA, const double* shiftB, double* dst, int B, int cs, int K, hipStream_t s) {
    hipLaunchKernelGGL(k_transpose_out, dim3((cs + 31) / 32, (K + 31) / 32, B), dim3(256), 0, s, src, shiftA, shiftB, dst, cs, K);
}

// env(action); reward(env) for the resident real envs (one workgroup per slot, lane c = car c)
__global__ void __launch_bounds__(64) k_env_step(EnvDesc env, double* x, int* t, int* done, const double* action,
                                                 double* reward, int* status, const int* alive) {
    const int b = blockIdx.x, c = threadIdx.x;
    if (alive && !alive[b]) return;
    __shared__ double srew[kMaxCars];
    if (env.kind != MPOPIS_ENV_CAR) {
        if (c == 0) {
            const double a = action[b];
            if (!(a >= env.lo[0] && a <= env.hi[0])) { if (status) status_raise(&status[b], MPOPIS_ERR_ACTION); }
            int tt = t[b], dd = done[b];
            simple_env_step(env, x + (size_t)b * env.ss, &tt, &dd, a);
            t[b] = tt; done[b] = dd;
            if (reward) reward[b] = simple_env_reward(env, x + (size_t)b * env.ss, dd);
        }
        return;
    }
    const int NC = env.ncars;
    double* xb = x + (size_t)b * 8 * NC;
    if (c < NC) {
        const double a0 = action[(size_t)b * 2 * NC + 2 * c], a1 = action[(size_t)b * 2 * NC + 2 * c + 1];
        if (NC == 1 && !(a0 >= env.lo[0] && a0 <= env.hi[0] && a1 >= env.lo[1] && a1 <= env.hi[1])) {
            if (status) status_raise(&status[b], MPOPIS_ERR_ACTION);                // car_racing.jl:239
        }
        CarState s;
        car_state_from8(s, xb + 8 * c);
        car_action_step(env.car, s, a0, a1);
        car_state_to8(s, xb + 8 * c);
        srew[c] = car_reward(env.car, env.track, s.x, s.y, s.Vx, s.Vy);
    }
    __syncthreads();
    if (c == 0) {
        t[b] += 1;
        if (reward) {
            double rew = 0.0;
            for (int i = 0; i < NC; ++i) {
                rew += srew[i];
                for (int j = i + 1; j < NC; ++j) {
                    const double dx = xb[8 * j] - xb[8 * i], dy = xb[8 * j + 1] - xb[8 * i + 1];
                    const double dd = sqrt(dx * dx + dy * dy);
                    rew += -dd;
                    if (dd <= 4.0) rew += -11000.0;
                }
            }
            reward[b] = rew;
        }
    }
}
// reward(env) / within_track / β of the resident state, no step
__global__ void __launch_bounds__(64) k_env_query(EnvDesc env, const double* x, const int* done, double* reward, int* within, double* dist, double* beta) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (env.kind != MPOPIS_ENV_CAR) {
        if (reward) reward[b] = simple_env_reward(env, x + (size_t)b * env.ss, done[b]);
        if (within) within[b] = 1;
        return;
    }
    const int NC = env.ncars;
    const double* xb = x + (size_t)b * 8 * NC;
    double rew = 0.0; int win = 1;
    for (int i = 0; i < NC; ++i) {
        const double* s = xb + 8 * i;
        double d;
        const bool w = within_track(env.track, s[0], s[1], &d, nullptr);
        if (!w) win = 0;
        if (dist) dist[(size_t)b * NC + i] = d;
        if (beta) beta[(size_t)b * NC + i] = atan2(s[4], s[3]);
        rew += car_reward(env.car, env.track, s[0], s[1], s[3], s[4]);
        for (int j = i + 1; j < NC; ++j) {
            const double dx = xb[8 * j] - s[0], dy = xb[8 * j + 1] - s[1];
            const double dd = sqrt(dx * dx + dy * dy);
            rew += -dd;
            if (dd <= 4.0) rew += -11000.0;
        }
    }
    if (reward) reward[b] = rew;
    if (within) within[b] = win;
}
void launch_env_query(const EnvDesc& env, const double* x, const int* done, double* reward, int* within, double* dist, double* beta, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_env_query, dim3(B), dim3(64), 0, s, env, x, done, reward, within, dist, beta);
}

void launch_env_step(const EnvDesc& env, double* x, int* t, int* done, const double* action, double* reward, int* status, const int* alive, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_env_step, dim3(B), dim3(64), 0, s, env, x, t, done, action, reward, status, alive);
}

}  // namespace mpopis
