// car_dynamics.h -- model dynamics + reward of the MPOPIS envs as inlined FP64 device math.
//
// What it computes (reference, paths relative to the MPOPIS repo root):
//   car_action_step  == CarRacingEnv functor + _step!      src/envs/car_racing.jl:238-250,282-344
//   car_reward       == reward(::CarRacingEnv)             src/envs/car_racing.jl:201-213
//   within_track     == within_track(::Track, pos)         src/envs/car_racing_tracks/car_racing_tracks.jl:68-92
//   mc_step/mc_reward== MountainCarEnv act!/_step! (RL.jl) + reward override
//                                                          src/examples/mountaincar_example.jl:4-22
//   cp_step/cp_reward== CartPoleEnv act!/_step!/reward (RL.jl), src/examples/cartpole_example.jl:3-6
//
// How (MI355X-first, not a transcription): one lane integrates one car.  The reference evaluates
// per Euler sub-step 3 atan2, 2 atan, 2 tan, 3 sincos pairs and 2 sqrt; in FP64 those are ~1000
// VALU instructions of OCML code per sub-step and dominate the whole MPC step.  The fast path here
// removes every transcendental from the sub-step with exact identities:
//   * tan(atan2(y,x) - d) = (y cos d - x sin d)/(x cos d + y sin d); |a| < atan(T) <=> in-half-plane
//     and |tan a| < T; so slip angles are never materialised (all signs of Vx handled, see
//     car_action_step).
//   * the pedal, hence fx, fz and the brush-model constants (fy_max, 3fy_max/C, C^2/3fy_max,
//     C^3/27fy_max^2) are constant over the 10 sub-steps of one action -> hoisted (2 sqrt per
//     action instead of per sub-step).
//   * sin/cos of delta and psi advance by angle-addition with the per-sub-step increments
//     (|d_delta| <= 0.0157, |d_psi| = |psi_dot|*0.01), carried in the state and renormalised once per
//     action; psi's atan(sin,cos) wrap is a conditional +-2pi; |β| > β_limit is a ratio test.
// These agree with the literal formulas to rounding (~1e-15 relative per step); the parity
// tests bound the end-to-end deviation against the CPU oracle at 1e-9 relative, far inside the
// 1e-5 contract.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MP_HD __host__ __device__ __forceinline__
#else
#define MP_HD inline
#endif

// dev builds (tools/path_stats.sh, -DMPOPIS_PATH_STATS): wave-level counts of which path a rollout kernel took -- [0] sub-steps, [1] sub-steps in which
// some lane took the general sub-step, [2] lanes in it, [3] reward evaluations, [4] evaluations through the general nearest-point search
#if defined(MPOPIS_PATH_STATS) && defined(__HIPCC__)
static __device__ unsigned long long g_path_stats[8];
static __device__ unsigned char g_sick[1 << 22];      // per thread of the launch: bit 0 took the general sub-step, bit 1 failed the 3-point ring certificate (own lane), bit 2 would fail a 5-point one
#endif
#if defined(MPOPIS_PATH_STATS) && defined(__HIP_DEVICE_COMPILE__)
#define MPOPIS_STAT(i, n) do { const unsigned long long ex_ = __builtin_amdgcn_read_exec(); \
        if ((int)(threadIdx.x & 63) == __ffsll((long long)ex_) - 1) atomicAdd(&g_path_stats[i], (unsigned long long)(n)); } while (0)
#define MPOPIS_STAT_LANES() __popcll(__builtin_amdgcn_read_exec())
#define MPOPIS_SICK(bit) do { const size_t t_ = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; if (t_ < sizeof(g_sick)) g_sick[t_] |= (bit); } while (0)
#endif
#ifndef MPOPIS_STAT
#define MPOPIS_STAT(i, n) do { } while (0)
#define MPOPIS_STAT_LANES() 0
#define MPOPIS_SICK(bit) do { } while (0)
#endif

namespace mpopis {

constexpr int kCarNParams = 20;
constexpr int kMcNParams = 8;
constexpr double kTwoPi = 6.283185307179586476925286766559;
constexpr double kPi = 3.141592653589793238462643383279;

struct CarParams {
    // CarRacingEnvParams, src/envs/car_racing.jl:2-21 (+dt, δt :33-34)
    double m, Izz, h, lf, lr, CD0, CD1, Caf, Car, muf, mur, dmax, ddotmax, Fxmax, Fxmin, lbrake, ldrive, blim, dt, ddt;
    // derived on the host once
    int nsub;            // round(Int, dt/δt) :299
    int blim_acute;      // β_limit < pi/2
    double inv_m, inv_Izz, L, tan_blim, inv_dt, inv_L, fz0f, fz0r;
    double k_v, k_rf, k_rr;   // δt/m, δt l_f/Izz, δt l_r/Izz (Euler updates of :326-328 with the constants folded)
    double mfz_f0, mfz_f1, mfz_r0, mfz_r1;   // μ f_z(fx) = mfz_0 -/+ mfz_1 fx for the front / rear axle (:262-272 with μ folded in)
};

MP_HD CarParams make_car_params(const double* p) {
    CarParams c;
    c.m = p[0]; c.Izz = p[1]; c.h = p[2]; c.lf = p[3]; c.lr = p[4]; c.CD0 = p[5]; c.CD1 = p[6];
    c.Caf = p[7]; c.Car = p[8]; c.muf = p[9]; c.mur = p[10]; c.dmax = p[11]; c.ddotmax = p[12];
    c.Fxmax = p[13]; c.Fxmin = p[14]; c.lbrake = p[15]; c.ldrive = p[16]; c.blim = p[17];
    c.dt = p[18]; c.ddt = p[19];
    c.nsub = (int)nearbyint(c.dt / c.ddt);
    c.inv_m = 1 / c.m; c.inv_Izz = 1 / c.Izz; c.L = c.lr + c.lf;
    c.inv_dt = 1 / c.dt; c.inv_L = 1 / c.L;
    c.fz0f = c.m * c.lr * 9.81; c.fz0r = c.m * c.lf * 9.81;                    // :262-272
    c.k_v = c.ddt * c.inv_m; c.k_rf = c.ddt * c.inv_Izz * c.lf; c.k_rr = c.ddt * c.inv_Izz * c.lr;
    c.mfz_f0 = c.muf * c.fz0f * c.inv_L; c.mfz_f1 = c.muf * c.h * c.inv_L; c.mfz_r0 = c.mur * c.fz0r * c.inv_L; c.mfz_r1 = c.mur * c.h * c.inv_L;
    c.blim_acute = c.blim < 0.5 * kPi;
    c.tan_blim = c.blim_acute ? tan(c.blim) : tan(kPi - c.blim);
    return c;
}

struct Track {           // env.track.{x′,y′,lane_width′} (+ n2[i] = x′[i]^2 + y′[i]^2, derived)
    const double* x; const double* y; const double* w; const double* n2; int P;
    // optional neighbour tables for the anchored nearest-point search (nullptr => always full scan):
    // row i = the nbrw points closest to q_i in ascending distance (rank 0 = i itself), row stride nbrw + 1
    const int* nbr_idx; const double* nbr_dist; int nbrw;
    // optional ring table for the straight-line fast paths of the rollout kernels (nullptr => within_track only): entry e, -kRingPad <= e <= P-1+kRingPad,
    // at ring[4 (e + kRingPad) .. +3] = {x, y, |q|^2, lane_width} of track point e mod P -- the ring neighbours of any point are then at
    // fixed offsets (no wrap-around index arithmetic); ring_cert[i] = ring_r2[i], ring_cert[P + i] = ring5_r2[i] (see below)
    const double* ring = nullptr; const double* ring_cert = nullptr;
};
constexpr int kRingPad = 3;          // a nearest point up to two ring steps from the anchor, plus its own ring neighbours
constexpr int kTrackNbrW = 16;
constexpr int kMaxTrackPoints = 2048;   // mpopis_set_track's limit (the rollout kernels stage 48 B per point in LDS)
// Row i of nbr_dist has one spare slot (index nbrw): it holds ring_r2[i] = (1 - 1e-9) x the squared distance from q_i to the nearest
// track point that is NOT one of its ring neighbours {i-1, i, i+1} (+inf when there is none).  If a position p satisfies
// 4 |p - q_i|^2 < ring_r2[i], every non-ring point j is farther from p than q_i is (|p - q_j| >= |q_j - q_i| - |p - q_i| > |p - q_i|),
// so the nearest point is one of the three ring candidates -- the common case for a car that moved <= 3 m since the last step.
// ring5_r2[i] is the same bound over the points outside {i-2 .. i+2}: on the bundled tracks (25 m between points, lane half-width 15 m) the
// three-point certificate gives out ~12 m from the anchor -- cars use the width of the road, so mid-lap 10-35 % of the rollouts lose it at
// some step and take their whole wave through the general search -- while the five-point one holds to ~18 m, i.e. everywhere inside the lane.

}  // namespace mpopis
#include <algorithm>
#include <utility>
#include <vector>
namespace mpopis {
// host: ring table (see Track::ring) and certification radii from the neighbour table
inline void build_track_ring(int P, const double* x, const double* y, const double* w, const double* n2, const std::vector<double>& nd, std::vector<double>& ring, std::vector<double>& cert);
// host: neighbour tables of the anchored nearest-point search (row stride W + 1, see Track) for P points; W = min(kTrackNbrW, P)
inline void build_track_tables(int P, const double* x, const double* y, std::vector<double>& nd, std::vector<int>& ni) {
    const int W = std::min<int>(kTrackNbrW, P), S = W + 1;
    nd.assign((size_t)P * S, 0.0); ni.assign((size_t)P * S, 0);
    for (int i = 0; i < P; ++i) {
        std::vector<std::pair<double, int>> v(P);
        for (int j = 0; j < P; ++j) v[j] = {sqrt((x[j] - x[i]) * (x[j] - x[i]) + (y[j] - y[i]) * (y[j] - y[i])), j};
        v[i].first = -1.0;                                       // rank 0 = the point itself
        std::sort(v.begin(), v.end());
        for (int c = 0; c < W; ++c) { nd[(size_t)i * S + c] = std::max(v[c].first, 0.0); ni[(size_t)i * S + c] = v[c].second; }
        const int im = (i == 0) ? P - 1 : i - 1, ip = (i == P - 1) ? 0 : i + 1;
        double r2 = INFINITY;
        for (int j = 0; j < P; ++j) if (j != i && j != im && j != ip) r2 = fmin(r2, (x[j] - x[i]) * (x[j] - x[i]) + (y[j] - y[i]) * (y[j] - y[i]));
        nd[(size_t)i * S + W] = r2 * (1.0 - 1e-9);
    }
}
// host: (1 - 1e-9) x the squared distance from q_i to the nearest track point more than `width` ring steps away (+inf when there is none)
inline double ring_cert_radius2(int P, const double* x, const double* y, int i, int width) {
    double r2 = INFINITY;
    for (int j = 0; j < P; ++j) {
        int d = j > i ? j - i : i - j;
        if (d > P - d) d = P - d;
        if (d > width) r2 = fmin(r2, (x[j] - x[i]) * (x[j] - x[i]) + (y[j] - y[i]) * (y[j] - y[i]));
    }
    return r2 * (1.0 - 1e-9);
}
// n2[i] = |q_i|^2 exactly as the caller uploads it for the general search (one computation, copied: ring_candidates and within_track must
// see bit-identical distances, or a near-tie could resolve differently depending on the path a wave took)
inline void build_track_ring(int P, const double* x, const double* y, const double* w, const double* n2, const std::vector<double>& nd, std::vector<double>& ring, std::vector<double>& cert) {
    const int W = std::min<int>(kTrackNbrW, P), S = W + 1;
    ring.assign((size_t)(P + 2 * kRingPad) * 4, 0.0); cert.assign((size_t)2 * P, 0.0);
    for (int e = -kRingPad; e < P + kRingPad; ++e) {
        const int i = ((e % P) + P) % P;
        double* o = ring.data() + (size_t)(e + kRingPad) * 4;
        o[0] = x[i]; o[1] = y[i]; o[2] = n2[i]; o[3] = w[i];
    }
    for (int i = 0; i < P; ++i) { cert[i] = nd[(size_t)i * S + W]; cert[P + i] = ring_cert_radius2(P, x, y, i, 2); }
}

// Rounding discipline of everything below: NO implicit a*b + c.  hipcc's default (-ffp-contract=fast, which this library is built with) lets the
// BACKEND fuse a multiply into a dependent add wherever it finds one, per inlining context and whatever pragma is in force -- the same source
// would then round differently in two kernels, and the wave-specialised rollout kernels, which evaluate parts of a model step in different
// waves, would stop agreeing bit for bit with the one-wave kernel.  So the model code leaves the compiler nothing to fuse: every fused
// multiply-add is written as fma(), every product that feeds an add or subtract unfused in the reference is kept away from it by being an
// fma operand already.  tools/check_contract.sh verifies it: the rollout kernels compile to the same instruction stream under
// -ffp-contract=fast and =off outside inlined libm code (sin / cos / fmod of the cold paths).  The pragma below makes the same statement
// for front ends that honour it (=on / =off builds, the host shim).
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
// all lanes / some lane of the wave (device); the single caller (host)
MP_HD bool wave_all(bool v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __all(v);
#else
    return v;
#endif
}
MP_HD unsigned long long wave_ballot(bool v) {     // the lanes for which v holds, as a scalar mask (host: bit 0)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(v);
#else
    return v ? 1ull : 0ull;
#endif
}
MP_HD bool wave_any(bool v) { return wave_ballot(v) != 0; }
MP_HD bool wave_mask_full(unsigned long long m) {   // m covers every active lane (host: the single caller)
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long ex = __builtin_amdgcn_read_exec();
    return (m & ex) == ex;
#else
    return (m & 1ull) != 0;
#endif
}
MP_HD int wave_lane() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)(threadIdx.x & 63);
#else
    return 0;
#endif
}

MP_HD double jl_sign(double v) { return (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : v); }
MP_HD double clampd(double v, double lo, double hi) { return v > hi ? hi : (v < lo ? lo : v); }

// clampd for wave-uniform bounds (scalar registers): v_min + v_max + NaN pass-through (clampd(NaN) = NaN, which is what
// makes a NaN action poison the rollout cost like the reference's "not in action space" error), 5 VALU ops instead of 12
MP_HD double clampd_u(double v, double lo, double hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    double t, r;
    asm("v_min_f64 %0, %1, %2" : "=v"(t) : "v"(v), "s"(hi));
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(t), "s"(lo));
    return (v != v) ? v : r;
#else
    return clampd(v, lo, hi);
#endif
}

// clampd for per-lane bounds in vector registers (multi-car kernel: the bounds differ between the cars of a wave): as clampd_u
MP_HD double clampd_v(double v, double lo, double hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    double t, r;
    asm("v_min_f64 %0, %1, %2" : "=v"(t) : "v"(v), "v"(hi));
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(t), "v"(lo));
    return (v != v) ? v : r;
#else
    return clampd(v, lo, hi);
#endif
}

// one Newton step on the v_rcp_f64 seed: <= 19 ulp (2.1e-15, measured in tools/rcp_acc.hip) -- used only for the slip tangents
// of the hot sub-step, where the result feeds a cubic whose own evaluation carries a comparable rounding error
MP_HD double fast_rcp1(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_amdgcn_rcp(v);
    return fma(fma(-v, r, 1.0), r, r);
#else
    return 1.0 / v;
#endif
}

MP_HD double fast_rcp(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    // v_rcp_f64 seed + 2 Newton steps: <= 1 ulp for normal-range inputs (Vx, rotated Vx here)
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.0), r, r);
    r = fma(fma(-v, r, 1.0), r, r);
    return r;
#else
    return 1.0 / v;
#endif
}

// sqrt for arguments that are >= 0 and not huge: v_rsq_f64 seed (2^-24) + one coupled Newton step + one residual
// correction = 1 ulp max error (measured, tools/rcp_acc.hip), without the denormal-range rescaling of the library
// sqrt (8 VALU ops instead of 18).  The 1e-300 bias keeps an exact 0 finite (result 1e-150) and is absorbed by any
// normal-range argument; NaN propagates.
MP_HD double fast_sqrt(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    v += 1e-300;
    const double y = __builtin_amdgcn_rsq(v);
    double g = v * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    return fma(fma(-g, g, v), h, g);
#else
    return sqrt(v);
#endif
}

// max(min(v, thr), -thr) (v_min_f64 + v_max_f64 with a negated source, no input canonicalisation; callers
// guarantee non-NaN arguments)
MP_HD double clamp_sym(double v, double thr) {
#if defined(__HIP_DEVICE_COMPILE__)
    double t, r;
    asm("v_min_f64 %0, %1, %2" : "=v"(t) : "v"(v), "v"(thr));
    asm("v_max_f64 %0, %1, -%2" : "=v"(r) : "v"(t), "v"(thr));
    return r;
#else
    return fmax(fmin(v, thr), -thr);
#endif
}

// fma with all three operands in VGPRs.  hipcc turns `p = fma(p, x, C)` with a loop-invariant constant C into
// `v_mov_b64 tmp, C; v_fmac_f64 tmp, p, x` (one extra VALU op per Horner step); the explicit VOP3 form needs no copy.
MP_HD double fma_v(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return fma(a, b, c);
#endif
}

struct TireK { double fymax, thr, k2, k3; };

// fy_max of an axle (:253) and, from it, the rest of the brush-model constants.  Two functions because the general sub-step only needs fy_max
// while the tyre saturates (the usual state of a car that is not moving forward) -- the other three are functions of fy_max alone, so deriving them
// on demand gives the same bits as deriving them up front.
MP_HD double tire_fymax(double mufz, double fxt) { return fast_sqrt(fmax(fma(mufz, mufz, -(fxt * fxt)), 1e-8)); }   // mufz = μ f_z
MP_HD TireK tire_from_fymax(double fymax, double Ca) {
    TireK k;
    k.fymax = fymax;
    const double rf = fast_rcp(k.fymax), rc = 1.0 / Ca;        // rc: Ca is wave-uniform
    k.thr = 3 * k.fymax * rc;                                  // tan of the switch angle :255
    k.k2 = ((Ca * Ca) * (1.0 / 3.0)) * rf;                     // C^2/(3 fy_max)
    k.k3 = ((Ca * Ca * Ca) * (1.0 / 27.0)) * (rf * rf);        // C^3/(27 fy_max^2)
    return k;
}
MP_HD TireK tire_consts(double mufz, double Ca, double fxt) { return tire_from_fymax(tire_fymax(mufz, fxt), Ca); }

// Car state as the kernels carry it: the reference's 8 doubles plus sin/cos of psi and delta, which
// are advanced by angle addition and never re-evaluated inside a rollout.
struct CarState { double x, y, psi, Vx, Vy, r, delta, pedal, sp, cp, sd, cd; int near; };   // near: nearest track point of the last reward (-1: unknown)
constexpr int kCarExt = 13;      // 8 state doubles, sin / cos of psi and delta, and the track point nearest to the start position (as a double)

MP_HD void car_state_from8(CarState& c, const double* s) {     // the only place sin/cos are evaluated
    c.x = s[0]; c.y = s[1]; c.psi = s[2]; c.Vx = s[3]; c.Vy = s[4]; c.r = s[5]; c.delta = s[6]; c.pedal = s[7];
    c.sp = sin(c.psi); c.cp = cos(c.psi); c.sd = sin(c.delta); c.cd = cos(c.delta); c.near = -1;
}
MP_HD void car_state_to8(const CarState& c, double* s) {
    s[0] = c.x; s[1] = c.y; s[2] = c.psi; s[3] = c.Vx; s[4] = c.Vy; s[5] = c.r; s[6] = c.delta; s[7] = c.pedal;
}

// sin/cos for |v| <= kTinyAngle = 1/32: the per-sub-step yaw / steering increments.  Truncation: sin v^9/9! (2.5e-18
// relative), cos v^8/8! (2.3e-17 absolute) -- below half an ulp, with 4 + 3 VALU ops after v^2.
constexpr double kTinyAngle = 0.03125;
MP_HD void sincos_tiny(double v, double* s, double* c) {
    const double v2 = v * v;
    double ps = -1.0 / 5040.0;
    ps = fma_v(ps, v2, 1.0 / 120.0);
    ps = fma_v(ps, v2, -1.0 / 6.0);
    *s = fma(ps * v2, v, v);
    // 1/24 is written one ulp high: its exact low dword coincides with that of -1/6, and hipcc then rebuilds the shared
    // half with a v_mov per use inside the sub-step loop (the ulp is invisible: the coefficient multiplies v^4)
    double pc = -1.0 / 720.0;
    pc = fma_v(pc, v2, 0x1.5555555555556p-5);
    pc = fma_v(pc, v2, -0.5);
    *c = fma(pc, v2, 1.0);
}

// brush tyre, linear branch, Horner form of  -C ta + k2 |ta| ta - k3 ta^3  (:256)
MP_HD double tire_poly(double ta, double Ca, const TireK& k) {
    const double at = fabs(ta);
    return ta * fma(at, fma(-k.k3, at, k.k2), -Ca);
}

// What the general sub-step needs of an action's forces for ONE value of sign(Vx) (:310-318): the axle drive / brake forces and the two fy_max.
// They depend on (pedal, sign Vx) only, so a rollout kernel derives them once per action and sign instead of once per sub-step (2 square roots).
struct AxleForces { double fxf, fxr, fymf, fymr; };
MP_HD AxleForces car_axle_forces(const CarParams& p, double pedal, double sg) {
    AxleForces a;
    const double fx = fma(p.Fxmax, fmax(pedal, 0.0), fma(p.Fxmin * fmin(pedal, 0.0), sg, pedal * 0.0));     // :310-312 (pedal * 0.0: a NaN pedal propagates, see car_action_consts)
    const double lam = (pedal <= 0) ? p.lbrake : p.ldrive;
    a.fxf = lam * fx; a.fxr = (1 - lam) * fx;
    a.fymf = tire_fymax(fma(-p.mfz_f1, fx, p.mfz_f0), a.fxf);                                // :262-272 (same derived constants as the hot path)
    a.fymr = tire_fymax(fma(p.mfz_r1, fx, p.mfz_r0), a.fxr);
    return a;
}

// The action-only half of env(a): everything a model step needs that depends on the ACTION and the steering angle but not on the vehicle state
// (position, velocities, yaw) -- the brush-model constants for sign(Vx) = +1 (:310-318) and the steering increment of a sub-step.  The steering
// angle itself evolves from the actions alone (:295-301), so a rollout kernel may compute all of this in another wave than the integration
// (k_rollout_car_trio); car_action_step below is the two halves back to back.
struct ActionConsts { double pedal, fxf, fxr0; TireK kf, kr; };
MP_HD ActionConsts car_action_consts(const CarParams& p, double a1) {
    ActionConsts k;
    const double pedal = a1;                                                   // :297
    // forces and brush-model constants for sign(Vx) = +1, constant over the sub-steps (:310-318)
    // fmax / fmin (v_max_f64 / v_min_f64) return the OTHER operand for a NaN: a NaN pedal would give fx = 0 and a finite rollout, where the reference's
    // max(pedal, 0) propagates it and env(a) throws "Action is not in action space" (car_racing.jl:239, :310).  pedal * 0.0 is 0 for every finite pedal and
    // NaN for a NaN one: as the addend of the second product it poisons fx -- and through it the state and the cost, which is what raises
    // MPOPIS_ERR_ACTION -- for one multiply per action (found by tests/test_dynamics_shim.py, round 6).
#ifndef MPOPIS_PEDAL_NAN
#define MPOPIS_PEDAL_NAN 1
#endif
#if MPOPIS_PEDAL_NAN
    const double fx = fma(p.Fxmax, fmax(pedal, 0.0), fma(p.Fxmin, fmin(pedal, 0.0), pedal * 0.0));
#else
    const double fx = fma(p.Fxmax, fmax(pedal, 0.0), p.Fxmin * fmin(pedal, 0.0));
#endif
    const double lam = (pedal <= 0) ? p.lbrake : p.ldrive;
    const double fxf = lam * fx, fxr = (1 - lam) * fx;
    k.pedal = pedal; k.fxf = fxf;
    k.fxr0 = fma(1 - lam, fx, -p.CD0);                         // rear drive force minus the constant part of the drag (:308); explicit: (1-λ) fx - CD0 is a*b + c
    k.kf = tire_consts(fma(-p.mfz_f1, fx, p.mfz_f0), p.Caf, fxf);
    k.kr = tire_consts(fma(p.mfz_r1, fx, p.mfz_r0), p.Car, fxr);
    return k;
}
// steering increment of one sub-step for the steering command a0 at steering angle delta (:295-296, :301), and its sin / cos
MP_HD void car_steer_step(const CarParams& p, double a0, double delta, double* dd_out, double* sdd_out, double* cdd_out) {
    const double tgt = fma(a0, p.dmax, -delta);
    // :295-296  min(|tgt|/dt, δ_dot_max)·sign(tgt): the magnitude is 0 when tgt is ±0, so copysign is the same function;
    // a NaN target (NaN action) must stay NaN (fmin would drop it)
    const double rmag = fmin(fabs(tgt) * p.inv_dt, p.ddotmax);
    const double rate = (tgt != tgt) ? tgt : copysign(rmag, tgt);
    const double dd = rate * p.ddt;
    double sdd, cdd;
    sincos_tiny(dd, &sdd, &cdd);                               // |dd| <= ddotmax*δt = 0.0157 with the reference's parameters
    if (__builtin_expect(fabs(dd) > kTinyAngle, 0)) { sdd = sin(dd); cdd = cos(dd); }   // (user-set δ_dot_max > 179 deg/s: library path)
    *dd_out = dd; *sdd_out = sdd; *cdd_out = cdd;
}
// first-order renormalisation of a (sin, cos) pair (the norm drifts by ~1e-16 per rotation, so callers inside long rollouts do this every few steps only)
MP_HD void renorm_pair(double& sn, double& cs) {
    const double f = fma(-0.5, fma(sn, sn, cs * cs), 1.5);
    sn *= f; cs *= f;
}
// delta += dd (:301) as a rotation of (sin delta, cos delta)
MP_HD void steer_rotate(double& sd, double& cd, double sdd, double cdd) {
    const double s2 = fma(sd, cdd, cd * sdd), c2 = fma(cd, cdd, -(sd * sdd));
    sd = s2; cd = c2;
}

// One Euler sub-step (:304-333; delta, sd, cd already advanced) for a lane in ANY state, as two halves: the tyre / axle forces by the lane's own
// rules, and ONE integration tail for all lanes.
//   HOT rules (Vx > 0 and the front slip in the forward half plane, xq > 0): branch-free, one shared reciprocal; the brush model is C1 at the
//     switch angle and saturates at -fy_max sign(alpha) beyond it (:255-259), so the cubic is evaluated at the clamped slip tangent.
//   GENERAL rules (stopped, sliding backwards, NaN), forces for the lane's sign(Vx) derived on the spot:
//     rear : alpha_r = atan2(yr,Vx).  |alpha_r| < atan(T)  <=>  Vx > 0 and |yr/Vx| < T; tan(alpha_r) = yr/Vx;
//            otherwise saturated with sign(alpha_r) = sign(yr) (atan2(0,0) = 0 gives fy = 0).
//     front: alpha_f = atan2(yf,Vx) - delta is the angle of q = R(-delta)(Vx,yf) up to a 2pi wrap that can only occur when |alpha_f| > pi
//            (saturated anyway).  Linear branch <=> q.x > 0 and |q.y/q.x| < T; saturated sign = sign(q.y) if q.x > 0 else sign(yf).
//     The switch tangent and the cubic's coefficients are derived from fy_max inside the linear branches only.
// Why two halves: a rollout that brakes to a standstill flips between the two rule sets every sub-step for the rest of its horizon (the
// reference's brake force flips with sign(Vx), :311) and takes its whole wave with it -- in a closed loop that is the usual state of a wave
// (2-11 % of the rollouts of a call stop inside the horizon, i.e. nearly every wave of 64 holds one).  Such a wave executes hot forces + general
// forces + one tail (~120 VALU) instead of two complete sub-steps (~160).  Round 5 measured the alternatives on the 64-trial headline workload
// (one stream; reset state / closed loop / frozen at step 100, ms per step): two complete sub-steps 5.68 / 6.36 / 7.72; this form with the cold
// lanes' position updated directly (a second divergent region) 5.76 / 6.30 / 7.26; a separate loop nest for waves with a slow lane, the
// sign(Vx) = -1 forces derived once per action -- inlined: 6.16 / 6.81 / 7.94 (the allocator spills the next step's prefetched noise around the
// whole action and copies five registers per trip), out of line (noinline, state through the stack): 6.47 / 8.51 / 14.3 (call ABI: 56 VGPR
// spills in the caller).  rdt = r δt is carried across sub-steps (this sub-step's "old yaw rate x δt" is the previous one's dψ).
template <bool PSI>
MP_HD void car_substep(const CarParams& p, const ActionConsts& k, double sd, double cd,
                              double& x, double& y, double& psi, double& Vx, double& Vy, double& r, double& sp, double& cp, double& rdt, double& sx, double& sy) {
    const double yf = fma(p.lf, r, Vy), yr = fma(-p.lr, r, Vy);                // :304-305 numerators
    double xq = fma(Vx, cd, yf * sd), yq = fma(yf, cd, -(Vx * sd));            // (Vx, yf) rotated by -delta
    const bool cold = !(Vx > 0.0 && xq > 0.0);
    double fyr, flat, flon, frear;
    if (__builtin_expect(cold, 0)) {                                           // stopped / sliding backwards / NaN
        MPOPIS_STAT(1, 1); MPOPIS_STAT(2, MPOPIS_STAT_LANES()); MPOPIS_SICK(1);
        const double sg = jl_sign(Vx);
        const AxleForces a = car_axle_forces(p, k.pedal, sg);
        const double f_drag = fma(p.CD1, fabs(Vx), p.CD0);
        const double fx_aero = (Vx > 0.0) ? f_drag : ((Vx < 0.0) ? -f_drag : f_drag * sg);
        fyr = (Vx == 0.0 && yr == 0.0) ? 0.0 : ((yr >= 0.0) ? -a.fymr : a.fymr);
        if (Vx > 0.0) {
            const TireK kr = tire_from_fymax(a.fymr, p.Car);
            const double ta = yr / Vx;
            if (fabs(ta) < kr.thr) fyr = tire_poly(ta, p.Car, kr);
        }
        if (Vx == 0.0 && yf == 0.0) { xq = cd; yq = -sd; }     // atan2(0,0) = 0 -> alpha_f = -delta
        double fyf = ((xq > 0.0 ? yq : yf) >= 0.0) ? -a.fymf : a.fymf;
        if (xq > 0.0) {
            const TireK kf = tire_from_fymax(a.fymf, p.Caf);
            const double ta = yq / xq;
            if (fabs(ta) < kf.thr) fyf = tire_poly(ta, p.Caf, kf);
        }
        flat = fma(fyf, cd, a.fxf * sd); flon = fma(a.fxf, cd, -(fyf * sd));
        frear = a.fxr - fx_aero;
    } else {
        const double rinv = fast_rcp1(Vx * xq);
        const double tar = yr * (rinv * xq), taf = yq * (rinv * Vx);           // tan(alpha_r), tan(alpha_f)
        fyr = tire_poly(clamp_sym(tar, k.kr.thr), p.Car, k.kr);
        const double fyf = tire_poly(clamp_sym(taf, k.kf.thr), p.Caf, k.kf);
        flat = fma(fyf, cd, k.fxf * sd); flon = fma(k.fxf, cd, -(fyf * sd));
        frear = fma(-p.CD1, Vx, k.fxr0);
    }
    const double rd = rdt;                                                     // old yaw rate x δt
    const double Vy1 = fma(p.k_v, flat + fyr, fma(-rd, Vx, Vy));               // :323,:328
    const double Vx1 = fma(p.k_v, flon + frear, fma(rd, Vy, Vx));              // :324,:327 (+ drag :308)
    r = fma(p.k_rf, flat, fma(-p.k_rr, fyr, r));                               // :322,:326
    Vx = Vx1; Vy = Vy1;
    rdt = r * p.ddt;
    const double dpsi = rdt;
    if (PSI) psi += dpsi;                                                      // :329
    double sq, cq;
    sincos_tiny(dpsi, &sq, &cq);
    const double sp0 = sp, cp0 = cp;
    { const double s2 = fma(sp, cq, cp * sq), c2 = fma(cp, cq, -(sp * sq)); sp = s2; cp = c2; }
    if (__builtin_expect(fabs(dpsi) > kTinyAngle, 0)) {
        const int nrot = (int)fmin(ceil(fabs(dpsi) * (1.0 / kTinyAngle)), 8192.0);
        sincos_tiny(dpsi / nrot, &sq, &cq);
        sp = sp0; cp = cp0;
        for (int q = 0; q < nrot; ++q) { const double s2 = fma(sp, cq, cp * sq), c2 = fma(cp, cq, -(sp * sq)); sp = s2; cp = c2; }
        if (PSI) psi = fmod(psi, kTwoPi);
    }
    if (PSI) psi -= (psi > kPi) ? kTwoPi : ((psi < -kPi) ? -kTwoPi : 0.0);     // :330 atan(sin,cos)
    if (PSI) {
        x = fma(fma(Vx, cp, -(Vy * sp)), p.ddt, x);                            // :331
        y = fma(fma(Vx, sp, Vy * cp), p.ddt, y);                               // :332
    } else {
        sx = fma(Vx, cp, fma(-Vy, sp, sx));                                    // Σ (Vx cos ψ - Vy sin ψ); x += δt Σ after the action (every lane,
        sy = fma(Vx, sp, fma(Vy, cp, sy));                                     // whichever force rules it went by: one more divergent region costs the hot path 2.5 %)
    }
}

// The state half of env(a): nsub Euler sub-steps (:299-333) of (x, y, psi, Vx, Vy, r) and (sin psi, cos psi) under the action constants k;
// (sdd, cdd) = sin / cos of the steering increment of a sub-step (delta += dd as a rotation of (sin delta, cos delta), :301).
// Rollouts (PSI = false) read the position only after the action: the position increments are summed in (sx, sy) and the common factor δt is
// applied once.
template <bool PSI>
MP_HD void car_integrate(const CarParams& p, const ActionConsts& k, double sdd, double cdd,
                         double& x, double& y, double& psi, double& Vx, double& Vy, double& r, double& sp, double& cp, double& sd, double& cd) {
    double rdt = r * p.ddt, sx = 0.0, sy = 0.0;
    for (int it = 0; it < p.nsub; it += 2) {                                   // two per trip: no loop-carried register copies
        steer_rotate(sd, cd, sdd, cdd);                                        // delta += dd :301
        MPOPIS_STAT(0, 1);
        car_substep<PSI>(p, k, sd, cd, x, y, psi, Vx, Vy, r, sp, cp, rdt, sx, sy);
        if (it + 1 < p.nsub) {                                                 // (odd sub-step counts: wave-uniform branch)
            steer_rotate(sd, cd, sdd, cdd);
            MPOPIS_STAT(0, 1);
            car_substep<PSI>(p, k, sd, cd, x, y, psi, Vx, Vy, r, sp, cp, rdt, sx, sy);
        }
    }
    if (!PSI) { x = fma(sx, p.ddt, x); y = fma(sy, p.ddt, y); }
}

// env(a) for one car (a0 steering, a1 pedal, already clamped): src/envs/car_racing.jl:282-344.
// Transcendental-free (requires |delta| < pi/2, guaranteed by delta_max and actions in [-1,1]).
// PSI = false drops the bookkeeping of the heading ANGLE (accumulate + wrap, :329-330): the dynamics and the reward
// only consume sin/cos(psi), which are carried by rotation, so rollouts that do not log trajectories never need it
// (c.psi is then left untouched = stale).
template <bool PSI = true>
MP_HD void car_action_step(const CarParams& p, CarState& c, double a0, double a1, bool renorm = true) {
    double x = c.x, y = c.y, psi = c.psi, Vx = c.Vx, Vy = c.Vy, r = c.r;
    double sp = c.sp, cp = c.cp, sd = c.sd, cd = c.cd;
    if (__builtin_expect(renorm, 0)) { renorm_pair(sp, cp); renorm_pair(sd, cd); }     // keep (sin,cos) pairs on the unit circle
    double dd, sdd, cdd;
    car_steer_step(p, a0, c.delta, &dd, &sdd, &cdd);
    const ActionConsts k = car_action_consts(p, a1);
    car_integrate<PSI>(p, k, sdd, cdd, x, y, psi, Vx, Vy, r, sp, cp, sd, cd);
    // delta advanced nsub times by dd (:301); the loop above only consumes sin/cos(delta)
    double delta = c.delta;
    if (PSI) { for (int it = 0; it < p.nsub; ++it) delta += dd; }              // real env / logged states: literal summation
    else delta = fma((double)p.nsub, dd, delta);
    c.x = x; c.y = y; if (PSI) c.psi = psi; c.Vx = Vx; c.Vy = Vy; c.r = r; c.delta = delta; c.pedal = k.pedal;
    c.sp = sp; c.cp = cp; c.sd = sd; c.cd = cd;
}

// Tail of within_track (car_racing_tracks.jl:75-90): nearest point p1 with its ring predecessor pm / successor pp -> distance of p from the
// line through p1 and the nearer of the two, and the lane test.  Shared by the general search and the ring fast path (identical arithmetic,
// so a rollout's cost does not depend on which of the two found the nearest point).
MP_HD bool track_project(double px, double py, double p1x, double p1y, double pmx, double pmy, double ppx, double ppy, double lane_w, double* dist_out) {
    const double ax = pmx - px, ay = pmy - py, bx = ppx - px, by = ppy - py;
    // :77-79 `dist(prev) <= dist(next)` decided on the squared distances (sqrt is monotone).  Only when the two squares
    // differ by an ulp or two could the rounding of the reference's square roots turn `>` into its tie (-> prev); at that
    // point the position is equidistant from both neighbours to 1e-16 and the reference's own choice is rounding noise.
    const double dm2 = fma(ax, ax, ay * ay), dp2 = fma(bx, bx, by * by);
    const bool prev = dm2 <= dp2;
    const double p2x = prev ? pmx : ppx, p2y = prev ? pmy : ppy;
    const double ux = px - p1x, uy = py - p1y, vx = p2x - p1x, vy = p2y - p1y;
    const double t = fma(ux, vx, uy * vy) * fast_rcp(fma(vx, vx, vy * vy));    // :87
    const double ex = fma(t, vx, p1x) - px, ey = fma(t, vy, p1y) - py;         // :88-89
    const double dist = fast_sqrt(fma(ex, ex, ey * ey));
    *dist_out = dist;
    return dist < lane_w;                                                      // :90
}

// within_track(track, pos): car_racing_tracks.jl:68-92.
// findmin over |q_i - p|^2 (:71-73) is evaluated as v_i = |q_i|^2 - 2 q_i.p (|p|^2 is common to all i): 2 FMAs per
// point; near-ties (< 1e-10 m^2 apart) may resolve differently from the literal form, which is harmless (the two
// candidates then share the projected segment).  `anchor` (in/out, -1 = none) is the nearest point found by the
// previous call of the same rollout: with D = |p - q_anchor|, every point farther than 2D from q_anchor is farther
// than D from p (triangle inequality), so only the first few entries of the anchor's neighbour list need scanning --
// the result is the exact argmin (ties -> lowest index, like findmin), typically after 3-5 instead of P evaluations.
// The search key of track point i for the position p (m2 = -2 p): v_i = |q_i|^2 - 2 q_i.p, and the order findmin imposes on (key, index) pairs --
// smaller key first, ties to the lower index.  ONE definition for every place that looks for a nearest point (within_track's three scans, the
// ring tiers' fall-back, k_step_begin's workgroup-wide scan for the start anchor): they must agree bit for bit.
MP_HD double track_key(const double* x, const double* y, const double* n2, int i, double m2x, double m2y) { return fma(y[i], m2y, fma(x[i], m2x, n2[i])); }
MP_HD bool track_key_before(double d, int j, double best, int mi) { return d < best || (d == best && j < mi); }

MP_HD bool within_track_m2(const Track& tk, double px, double py, double m2x, double m2y, double* dist_out, int* anchor) {    // m2 = -2 p, from the caller
    int mi = -1;
    double best = 0.0;
    const int a0 = anchor ? *anchor : -1;
    double p1x = 0.0, p1y = 0.0, pmx = 0.0, pmy = 0.0, ppx = 0.0, ppy = 0.0;      // nearest point, its ring predecessor / successor
    bool have_pts = false;
    if (__builtin_expect(a0 >= 0 && tk.nbr_idx, 1)) {
        const int S = tk.nbrw + 1;
        // ring candidates {a0-1, a0, a0+1}: all addresses known up front (one LDS round trip), no list indirection
        const int am = (a0 == 0) ? tk.P - 1 : a0 - 1, ap = (a0 == tk.P - 1) ? 0 : a0 + 1;
        const double x0 = tk.x[a0], y0 = tk.y[a0], xm = tk.x[am], ym = tk.y[am], xp = tk.x[ap], yp = tk.y[ap];
        const double d0 = track_key(tk.x, tk.y, tk.n2, a0, m2x, m2y);
        const double D02 = d0 + fma(px, px, py * py);                        // |p - q_a0|^2
        if (__builtin_expect(4.0 * D02 < tk.nbr_dist[a0 * S + tk.nbrw], 1)) {  // certified: no non-ring point can be nearer (see Track)
            const double dm = track_key(tk.x, tk.y, tk.n2, am, m2x, m2y), dp = track_key(tk.x, tk.y, tk.n2, ap, m2x, m2y);
            mi = a0; best = d0;
            if (track_key_before(dm, am, best, mi)) { best = dm; mi = am; }      // first minimum: ties -> lowest index, like findmin
            if (track_key_before(dp, ap, best, mi)) { best = dp; mi = ap; }
            // the nearest point's own ring neighbours: two of the three are already here, the third is one more point
            const int e = (mi == am) ? ((am == 0) ? tk.P - 1 : am - 1) : ((ap == tk.P - 1) ? 0 : ap + 1);
            const double xe = tk.x[e], ye = tk.y[e];
            if (mi == a0) { p1x = x0; p1y = y0; pmx = xm; pmy = ym; ppx = xp; ppy = yp; }
            else if (mi == am) { p1x = xm; p1y = ym; pmx = xe; pmy = ye; ppx = x0; ppy = y0; }
            else { p1x = xp; p1y = yp; pmx = x0; pmy = y0; ppx = xe; ppy = ye; }
            have_pts = true;
        } else {
            // far from the anchor (or a track that folds back on itself): scan the anchor's neighbour list up to the triangle bound
            mi = a0; best = d0;
            const double bound = fma(2.0, fast_sqrt(fmax(D02, 0.0)), 1e-6);
            bool closed = false;
            for (int c = 1; c < tk.nbrw; ++c) {
                if (tk.nbr_dist[a0 * S + c] >= bound) { closed = true; break; }
                const int j = tk.nbr_idx[a0 * S + c];
                const double d = track_key(tk.x, tk.y, tk.n2, j, m2x, m2y);
                if (track_key_before(d, j, best, mi)) { best = d; mi = j; }
            }
            if (__builtin_expect(!closed && tk.nbrw < tk.P, 0)) mi = -1;   // list exhausted before the bound: full scan
        }
    }
    if (__builtin_expect(mi < 0, 0)) {
        mi = 0;
        best = track_key(tk.x, tk.y, tk.n2, 0, m2x, m2y);
#pragma unroll 8
        for (int i = 1; i < tk.P; ++i) {                                       // first minimum (ascending i: track_key_before without its tie clause)
            const double d = track_key(tk.x, tk.y, tk.n2, i, m2x, m2y);
            mi = (d < best) ? i : mi;
            best = fmin(best, d);
        }
    }
    if (anchor) *anchor = mi;
    if (!have_pts) {
        const int im = (mi == 0) ? tk.P - 1 : mi - 1;                          // mod1 :75-76
        const int ip = (mi == tk.P - 1) ? 0 : mi + 1;
        p1x = tk.x[mi]; p1y = tk.y[mi]; pmx = tk.x[im]; pmy = tk.y[im]; ppx = tk.x[ip]; ppy = tk.y[ip];
    }
    return track_project(px, py, p1x, p1y, pmx, pmy, ppx, ppy, tk.w[mi], dist_out);
}

MP_HD bool within_track(const Track& tk, double px, double py, double* dist_out, int* anchor) {
    return within_track_m2(tk, px, py, -2.0 * px, -2.0 * py, dist_out, anchor);
}

// Straight-line fast path of the nearest-point search for the rollout kernels (anchor = the previous step's nearest point, ring table in LDS).
// Applies when the anchor's certification holds (the nearest point is one of {a0-1, a0, a0+1}, see Track) and the three candidate distances
// are pairwise different (no tie to break by index): then the argmin is two comparisons, its ring neighbours sit at fixed offsets of the padded
// table, and no index arithmetic, list scan or point permutation is needed.  Returns false when it does not apply (first step, NaN position,
// far from the anchor, exact ties): the caller then runs within_track, which finds the same point by the general rules -- and both end in
// track_project, so the result never depends on the path taken.  rel (out) = nearest point - anchor in ring steps (-1, 0, +1).
// (device: returns the wave mask of the lanes it applies to -- compares straight into scalar registers, ANDed there -- and the caller tests
// "all lanes"; going through a per-lane bool and __all costs two more vector instructions per evaluation)
MP_HD unsigned long long ring_candidates(const double* ring, const double* cert, int a0, double px, double py, double m2x, double m2y, int* rel) {
    const int a = a0 < 0 ? 0 : a0;                                             // branch-free: a missing anchor reads entry 0 and reports "not applicable"
    const double* e0 = ring + (size_t)4 * (a + kRingPad);
    const double d0 = fma(e0[1], m2y, fma(e0[0], m2x, e0[2]));                 // |q|^2 - 2 q.p, as in within_track
    const double dm = fma(e0[-3], m2y, fma(e0[-4], m2x, e0[-2]));
    const double dp = fma(e0[5], m2y, fma(e0[4], m2x, e0[6]));
    const double D02 = d0 + fma(px, px, py * py);
    *rel = (dm < d0 && dm < dp) ? -1 : ((dp < d0 && dp < dm) ? 1 : 0);
#if defined(__HIP_DEVICE_COMPILE__)
    // __builtin_amdgcn_[fs]cmp: LLVM predicate codes -- 38 = signed >, 4 = ordered <, 14 = unordered or != (C's != : true for NaN)
    return __builtin_amdgcn_sicmp(a0, -1, 38) & __builtin_amdgcn_fcmp(4.0 * D02, cert[a], 4) & __builtin_amdgcn_fcmp(d0, dm, 14) &
           __builtin_amdgcn_fcmp(d0, dp, 14) & __builtin_amdgcn_fcmp(dm, dp, 14);
#else
    return ((a0 >= 0) & (4.0 * D02 < cert[a]) & (d0 != dm) & (d0 != dp) & (dm != dp)) ? 1ull : 0ull;
#endif
}
// Second tier, tried by a wave in which some lane failed the three-point test: five candidates {a0-2 .. a0+2} under the wider certificate
// cert5 = ring_cert + P (see Track).  Applies when that certificate holds and the smallest of the five distances is attained ONCE (a tie would
// have to be broken by track index, which the general search does).  rel (out) in -2 .. 2.
MP_HD unsigned long long ring5_candidates(const double* ring, const double* cert5, int a0, double px, double py, double m2x, double m2y, int* rel) {   // (wave mask, like ring_candidates)
    const int a = a0 < 0 ? 0 : a0;
    const double* e0 = ring + (size_t)4 * (a + kRingPad);
    const double d0 = fma(e0[1], m2y, fma(e0[0], m2x, e0[2]));
    const double dm = fma(e0[-3], m2y, fma(e0[-4], m2x, e0[-2]));
    const double dp = fma(e0[5], m2y, fma(e0[4], m2x, e0[6]));
    const double dmm = fma(e0[-7], m2y, fma(e0[-8], m2x, e0[-6]));
    const double dpp = fma(e0[9], m2y, fma(e0[8], m2x, e0[10]));
    const double D02 = d0 + fma(px, px, py * py);
    double best = d0; int r = 0;
    if (dm < best) { best = dm; r = -1; }
    if (dp < best) { best = dp; r = 1; }
    if (dmm < best) { best = dmm; r = -2; }
    if (dpp < best) { best = dpp; r = 2; }
    const int hits = (d0 == best) + (dm == best) + (dp == best) + (dmm == best) + (dpp == best);
    *rel = r;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sicmp(a0, -1, 38) & __builtin_amdgcn_fcmp(4.0 * D02, cert5[a], 4) & __builtin_amdgcn_sicmp(hits, 1, 32);   // (a NaN position fails the certificate)
#else
    return ((a0 >= 0) & (4.0 * D02 < cert5[a]) & (hits == 1)) ? 1ull : 0ull;
#endif
}
MP_HD bool ring_project(const double* ring, int a0, int rel, double px, double py, double* dist_out) {
    const double* e = ring + (size_t)4 * (a0 + rel + kRingPad);
    return track_project(px, py, e[0], e[1], e[-4], e[-3], e[4], e[5], e[3], dist_out);
}
MP_HD int ring_wrap(int mi, int P) { return (mi < 0) ? mi + P : ((mi >= P) ? mi - P : mi); }

// reward(env::CarRacingEnv): src/envs/car_racing.jl:201-213
// |atan2(Vy,Vx)| > β_limit without the atan2 (tan_blim = tan(β_limit) or tan(pi-β_limit), host-side)
MP_HD bool exceed_beta(const CarParams& p, double Vx, double Vy) {
    if (p.blim_acute) return (Vx > 0.0) ? (fabs(Vy) > p.tan_blim * Vx) : !(Vx == 0.0 && Vy == 0.0);
    return (Vx < 0.0) && (fabs(Vy) < p.tan_blim * (-Vx));
}

MP_HD double car_reward(const CarParams& p, const Track& tk, double x, double y, double Vx, double Vy, int* anchor = nullptr) {
    double dist;
    bool within;
    int rel = 0;
    // rollout kernels (anchor carried, ring table staged): one wave-uniform branch -- every lane on the straight-line path, or the whole
    // wave through the general search (first step of a rollout, a lane far off its anchor, exact ties)
    const double m2x = -2.0 * x, m2y = -2.0 * y;               // -2 p of the search key |q|^2 - 2 q.p: once, for whichever search runs
    const unsigned long long fast_lanes = (tk.ring && anchor) ? ring_candidates(tk.ring, tk.ring_cert, *anchor, x, y, m2x, m2y, &rel) : 0ull;
    const bool fast = (fast_lanes >> wave_lane()) & 1ull;      // (dev builds with MPOPIS_PATH_STATS only; dead otherwise)
    (void)fast;
    MPOPIS_STAT(3, 1);
#if defined(MPOPIS_PATH_STATS) && defined(__HIP_DEVICE_COMPILE__)
    if (tk.ring && anchor && *anchor >= 0 && !fast) {
        MPOPIS_SICK(2);
        // would a 5-point ring certificate hold?  radius = half the distance from the anchor to the nearest point outside {a-2..a+2} (brute force here)
        const int a_ = *anchor; double r2_ = INFINITY;
        for (int j_ = 0; j_ < tk.P; ++j_) { int d_ = j_ - a_; if (d_ < 0) d_ = -d_; if (d_ > tk.P - d_) d_ = tk.P - d_; if (d_ > 2) { const double ex_ = tk.ring[4 * (j_ + kRingPad)] - tk.ring[4 * (a_ + kRingPad)], ey_ = tk.ring[4 * (j_ + kRingPad) + 1] - tk.ring[4 * (a_ + kRingPad) + 1]; r2_ = fmin(r2_, ex_ * ex_ + ey_ * ey_); } }
        const double dx_ = x - tk.ring[4 * (a_ + kRingPad)], dy_ = y - tk.ring[4 * (a_ + kRingPad) + 1];
        if (!(4.0 * (dx_ * dx_ + dy_ * dy_) < r2_)) MPOPIS_SICK(4);
    }
#endif
    if (__builtin_expect(tk.ring && anchor && wave_mask_full(fast_lanes), 1)) {
        within = ring_project(tk.ring, *anchor, rel, x, y, &dist);
        *anchor = ring_wrap(*anchor + rel, tk.P);
    } else if (tk.ring && anchor && tk.P >= 5 && wave_mask_full(ring5_candidates(tk.ring, tk.ring_cert + tk.P, *anchor, x, y, m2x, m2y, &rel))) {
        // some lane is farther from its anchor than the three-point certificate reaches (a car using the width of the road): five candidates, same tail
        MPOPIS_STAT(5, 1);
        within = ring_project(tk.ring, *anchor, rel, x, y, &dist);
        *anchor = ring_wrap(*anchor + rel, tk.P);
    } else {
        MPOPIS_STAT(4, 1);
        within = within_track_m2(tk, x, y, m2x, m2y, &dist, anchor);
    }
    double rew = 0.0;
    if (!within) rew += -1000000.0;
    if (exceed_beta(p, Vx, Vy)) rew += -5000.0;                                // exceed_β :184-189
    rew += -dist;
    rew = fma(2.0, fast_sqrt(fma(Vx, Vx, Vy * Vy)), rew);
    return rew;
}

// one step's term of the control cost γ U_orig' Σ^-1 (V - U_orig) for a two-dimensional action (mppi_mpopi_policies.jl:272)
MP_HD double control_cost_term(double g0, double d0, double g1, double d1) { return fma(g0, d0, g1 * d1); }

// ---- MountainCar (continuous) -------------------------------------------------------------------
struct McParams { double min_pos, max_pos, max_speed, goal_pos, goal_vel, power, gravity; int max_steps; };

MP_HD McParams make_mc_params(const double* p) {
    McParams m;
    m.min_pos = p[0]; m.max_pos = p[1]; m.max_speed = p[2]; m.goal_pos = p[3]; m.goal_vel = p[4];
    m.power = p[5]; m.gravity = p[6]; m.max_steps = (int)p[7];
    return m;
}

MP_HD void mc_step(const McParams& p, double* s, int* t, int* done, double force) {
    *t += 1;
    double x = s[0], v = s[1];
    v += force * p.power + cos(3 * x) * (-p.gravity);
    v = clampd(v, -p.max_speed, p.max_speed);
    x += v;
    x = clampd(x, p.min_pos, p.max_pos);
    if (x == p.min_pos && v < 0) v = 0;
    *done = ((x >= p.goal_pos && v >= p.goal_vel) || (*t >= p.max_steps)) ? 1 : 0;
    s[0] = x; s[1] = v;
}

// src/examples/mountaincar_example.jl:10-22
MP_HD double mc_reward(const McParams& p, const double* s, int done) {
    double rew = 0.0;
    if (s[0] >= p.goal_pos && s[1] >= p.goal_vel) rew += 100000;
    rew += fabs(s[1]);
    rew += done ? 0.0 : -1.0;
    return rew;
}

// ---- CartPole (continuous) ----------------------------------------------------------------------
// RL.jl CartPoleEnv(continuous=true) [third-party, recalled: unpinned] driven by the functor of
// src/examples/cartpole_example.jl:3-6; reward is RL.jl's own (done ? 0 : 1).
struct CpParams { double gravity, masscart, masspole, totalmass, halflength, polemasslength, forcemag, dt, theta_thr, x_thr; int max_steps; };

MP_HD CpParams make_cp_params(const double* p) {
    CpParams c;
    c.gravity = p[0]; c.masscart = p[1]; c.masspole = p[2]; c.totalmass = p[3]; c.halflength = p[4]; c.polemasslength = p[5];
    c.forcemag = p[6]; c.dt = p[7]; c.theta_thr = p[8]; c.x_thr = p[9]; c.max_steps = (int)p[10];
    return c;
}

MP_HD void cp_step(const CpParams& p, double* s, int* t, int* done, double a) {
    *t += 1;
    const double force = a * p.forcemag;
    const double xdot = s[1], theta = s[2], thetadot = s[3];
    const double costheta = cos(theta), sintheta = sin(theta);
    const double tmp = (force + p.polemasslength * (thetadot * thetadot) * sintheta) / p.totalmass;
    const double thetaacc = (p.gravity * sintheta - costheta * tmp) /
                            (p.halflength * (4.0 / 3.0 - p.masspole * (costheta * costheta) / p.totalmass));
    const double xacc = tmp - p.polemasslength * thetaacc * costheta / p.totalmass;
    s[0] += p.dt * xdot;
    s[1] += p.dt * xacc;
    s[2] += p.dt * thetadot;
    s[3] += p.dt * thetaacc;
    *done = (fabs(s[0]) > p.x_thr || fabs(s[2]) > p.theta_thr || *t > p.max_steps) ? 1 : 0;
}

MP_HD double cp_reward(int done) { return done ? 0.0 : 1.0; }
#if defined(__clang__)
#pragma STDC FP_CONTRACT DEFAULT      // back to the translation unit's own setting (whatever -ffp-contract says), not a hard-coded mode
#endif

}  // namespace mpopis
