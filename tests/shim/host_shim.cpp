// Test-only host build of the engine's shared dynamics header (mpopis_amd/csrc/car_dynamics.h), so
// the fast-path algebra used by the HIP rollout kernel can be checked against the CPU oracle on a
// machine without a GPU.  Not part of the product; never loaded by mpopis_amd.
#include "../../mpopis_amd/csrc/car_dynamics.h"
#include <vector>
using namespace mpopis;
#include <algorithm>
static std::vector<double> g_nd, g_ring, g_cert; static std::vector<int> g_ni;
static Track mk(int P, const double* tx, const double* ty, const double* tw, std::vector<double>& n2) {
    n2.resize(P); for (int i = 0; i < P; ++i) n2[i] = tx[i] * tx[i] + ty[i] * ty[i];
    const int W = std::min<int>(kTrackNbrW, P);
    build_track_tables(P, tx, ty, g_nd, g_ni);
    build_track_ring(P, tx, ty, tw, n2.data(), g_nd, g_ring, g_cert);
    return Track{tx, ty, tw, n2.data(), P, g_ni.data(), g_nd.data(), W, g_ring.data(), g_cert.data()};
}
extern "C" {
void shim_car_action_step(const double* p20, double* s8, double a0, double a1) {
    CarParams p = make_car_params(p20);
    CarState c; car_state_from8(c, s8);
    car_action_step(p, c, a0, a1);
    car_state_to8(c, s8);
}
double shim_car_reward(const double* p20, int P, const double* tx, const double* ty, const double* tw, const double* s8) {
    CarParams p = make_car_params(p20);
    std::vector<double> n2; Track tk = mk(P, tx, ty, tw, n2);
    return car_reward(p, tk, s8[0], s8[1], s8[3], s8[4]);
}
// full single-car rollout: controls as x T (already clamped); returns -sum(reward)
double shim_car_rollout(const double* p20, int P, const double* tx, const double* ty, const double* tw,
                        double* s8, const double* ctrl, int T) {
    CarParams p = make_car_params(p20);
    std::vector<double> n2; Track tk = mk(P, tx, ty, tw, n2);
    double c = 0.0;
    CarState st; car_state_from8(st, s8);          // sin/cos evaluated once, then carried (as in the kernel)
    for (int t = 0; t < T; ++t) { car_action_step(p, st, ctrl[2 * t], ctrl[2 * t + 1]); c -= car_reward(p, tk, st.x, st.y, st.Vx, st.Vy, &st.near); }
    car_state_to8(st, s8);
    return c;
}
void shim_mc_step(const double* p8, double* s2, int* t, int* done, double f) { McParams p = make_mc_params(p8); mc_step(p, s2, t, done, f); }
double shim_mc_reward(const double* p8, const double* s2, int done) { McParams p = make_mc_params(p8); return mc_reward(p, s2, done); }
}
extern "C" int shim_within_anchor(int P, const double* tx, const double* ty, const double* tw, double px, double py, int* anchor, double* dist) {
    static std::vector<double> n2; static Track tk; static const double* last = nullptr;
    if (last != tx) { tk = mk(P, tx, ty, tw, n2); last = tx; }
    return within_track(tk, px, py, dist, anchor) ? 1 : 0;
}
// the rollout kernels' path: ring fast path when it applies, general search otherwise (car_dynamics.h: car_reward); *fast = which one ran
extern "C" int shim_within_ring(int P, const double* tx, const double* ty, const double* tw, double px, double py, int* anchor, double* dist, int* fast) {
    static std::vector<double> n2; static Track tk; static const double* last = nullptr;
    if (last != tx) { tk = mk(P, tx, ty, tw, n2); last = tx; }
    int rel = 0;
    bool ok = ring_candidates(tk.ring, tk.ring_cert, *anchor, px, py, -2.0 * px, -2.0 * py, &rel);
    *fast = ok ? 1 : 0;
    if (!ok && tk.P >= 5) {                                    // second tier of car_reward: five candidates under the wider certificate
        ok = ring5_candidates(tk.ring, tk.ring_cert + tk.P, *anchor, px, py, -2.0 * px, -2.0 * py, &rel);
        if (ok) *fast = 2;
    }
    if (ok) {
        const bool w = ring_project(tk.ring, *anchor, rel, px, py, dist);
        *anchor = ring_wrap(*anchor + rel, tk.P);
        return w ? 1 : 0;
    }
    return within_track(tk, px, py, dist, anchor) ? 1 : 0;
}
