"""Thin NumPy-facing wrapper over the C ABI: one `Engine` = one mpopis_handle (B resident trials).

Array conventions follow the reference (Julia, column-major): E is handed over as an array of
shape (B, K, cs) in C order, i.e. for every trial the memory of Julia's cs x K matrix.
"""
import ctypes as C
import os
import numpy as np

from . import _lib
from ._lib import Config, Noise, MPOPISError, check

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def default_track(width=15.0):
    """The reference's default track: curve.csv sub-sampled with sample_factor=20 (48 points),
    lane half-width 15 (car_racing_tracks.jl:21-23,30; car_racing.jl:91)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "curve_sf20.csv")
    d = np.loadtxt(path, delimiter=",")
    return _f64(d[:, 0]), _f64(d[:, 1]), _f64(np.full(d.shape[0], width))


class Engine:
    def __init__(self, env_kind="car", num_cars=1, policy="gmppi", num_samples=50, horizon=50, batch=1,
                 lam=1.0, alpha=1.0, ais_its=10, lam_ais=20.0, elite_threshold=0.8, sigma_est="mle",
                 cma_sigma=1.0, seed=0, device=0, log_trajectories=False, track="default",
                 env_params=None, cov=None, U0=None):
        L = _lib.lib()
        cfg = Config()
        cfg.device = device
        cfg.env_kind = _lib.ENV_CAR if env_kind == "car" else _lib.ENV_MOUNTAINCAR
        cfg.num_cars = num_cars if env_kind == "car" else 0
        cfg.policy = _lib.POLICY_IDS[policy.lstrip(":")]
        cfg.num_samples, cfg.horizon, cfg.batch, cfg.ais_its = num_samples, horizon, batch, ais_its
        cfg.sigma_est = _lib.SIGMA_EST_IDS[sigma_est]
        cfg.log_trajectories = int(log_trajectories)
        cfg.lambda_, cfg.alpha, cfg.lambda_ais = lam, alpha, lam_ais
        cfg.elite_threshold, cfg.cma_sigma, cfg.seed = elite_threshold, cma_sigma, seed
        self._h = C.c_void_p()
        rc = L.mpopis_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = L.mpopis_last_error(None)
            raise MPOPISError(rc, msg.decode() if msg else "")
        self.L = L
        self.policy = policy.lstrip(":")
        self.B, self.K, self.T = batch, num_samples, horizon
        self.as_ = 2 * num_cars if env_kind == "car" else 1
        self.ss = 8 * num_cars if env_kind == "car" else 2
        self.cs = self.as_ * horizon
        self.N = 1 if self.policy in ("mppi", "gmppi") else max(1, ais_its)
        self.env_kind = env_kind
        if env_params is not None:
            p = _f64(env_params)
            check(self._h, L.mpopis_set_env_params(self._h, _d(p), len(p)))
        if env_kind == "car" and track is not None:
            tx, ty, tw = default_track() if isinstance(track, str) else track
            self.set_track(tx, ty, tw)
        if cov is not None:
            self.set_Sigma(cov)
        if U0 is not None:
            U0 = _f64(U0)
            if U0.size == self.as_:
                U0 = np.tile(U0, horizon)                       # :61-63
            if U0.size != self.cs:
                raise MPOPISError(_lib.ERR_ARG, "U₀ must be length of action space or control space")   # :64
            self.set_U(np.tile(U0, (batch, 1)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.L.mpopis_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- env / policy state -------------------------------------------------------------------
    def set_track(self, tx, ty, tw):
        tx, ty, tw = _f64(tx), _f64(ty), _f64(tw)
        check(self._h, self.L.mpopis_set_track(self._h, _d(tx), _d(ty), _d(tw), len(tx)))

    def set_action_bounds(self, lo, hi):
        lo, hi = _f64(lo), _f64(hi)
        check(self._h, self.L.mpopis_set_action_bounds(self._h, _d(lo), _d(hi)))

    def reset(self):
        check(self._h, self.L.mpopis_reset(self._h))

    def set_state(self, x, t=None, done=None):
        x = _f64(x).reshape(self.B, self.ss)
        t = None if t is None else np.ascontiguousarray(t, dtype=np.int32)
        done = None if done is None else np.ascontiguousarray(done, dtype=np.int32)
        check(self._h, self.L.mpopis_set_state(self._h, _d(x), _i(t), _i(done)))

    def get_state(self):
        x = np.zeros((self.B, self.ss))
        t = np.zeros(self.B, dtype=np.int32)
        done = np.zeros(self.B, dtype=np.int32)
        check(self._h, self.L.mpopis_get_state(self._h, _d(x), _i(t), _i(done)))
        return x, t, done

    def set_U(self, U):
        U = _f64(U).reshape(self.B, self.cs)
        check(self._h, self.L.mpopis_set_U(self._h, _d(U)))

    def get_U(self):
        U = np.zeros((self.B, self.cs))
        check(self._h, self.L.mpopis_get_U(self._h, _d(U)))
        return U

    def set_Sigma(self, cov):
        cov = np.asarray(cov, dtype=np.float64)
        if cov.ndim == 1:
            cov = np.diag(cov)                                  # block_diagm(::Vector) utils.jl:9-11
        if cov.ndim != 2 or cov.shape[0] != cov.shape[1]:
            raise MPOPISError(_lib.ERR_ARG, "Covriance must be square")      # :80
        cm = np.ascontiguousarray(cov.T)                        # column-major
        check(self._h, self.L.mpopis_set_Sigma(self._h, _d(cm), cov.shape[0]))

    def seed(self, seed):
        check(self._h, self.L.mpopis_seed(self._h, seed))

    # ---- Level 1 ----------------------------------------------------------------------------------
    def rollout_costs(self, U, E, x0=None, U_orig=None, Sigma_inv=None):
        """simulate_model: U (B,cs); E (B,K,cs) [= Julia cs x K per trial]; returns cost (B,K)."""
        U = _f64(U).reshape(self.B, self.cs)
        E = _f64(E).reshape(self.B, self.K, self.cs)
        x0 = None if x0 is None else _f64(x0).reshape(self.B, self.ss)
        Uo = None if U_orig is None else _f64(U_orig).reshape(self.B, self.cs)
        Si = None if Sigma_inv is None else np.ascontiguousarray(np.asarray(Sigma_inv, dtype=np.float64).T)
        cost = np.zeros((self.B, self.K))
        check(self._h, self.L.mpopis_rollout_costs(self._h, _d(x0), _d(U), _d(Uo), _d(E), _d(Si), _d(cost)))
        return cost

    # ---- Level 2 ----------------------------------------------------------------------------------
    def policy_step(self, Z=None, res_i0=None, res_u=None, want_E=False, minimal=False):
        """pol(env) for all slots.  Z: injected normals, (B,N,K,cs) [G-variants] or (B,T,K,as) [:mppi];
        None => device RNG.  Returns dict(control, cost, weights, iters_run, [E], [res_idx0]);
        minimal=True copies back only control and iters_run (what `act = pol(env)` needs)."""
        B, K, cs, N = self.B, self.K, self.cs, self.N
        nz = None
        keep = []
        if Z is not None:
            Z = _f64(Z)
            assert Z.size == B * N * K * cs, "noise size"
            nz = Noise()
            nz.Z = _d(Z)
            keep.append(Z)
            if res_i0 is not None:
                ri = np.ascontiguousarray(res_i0, dtype=np.int32)
                ru = _f64(res_u)
                nz.res_i0, nz.res_u = _i(ri), _d(ru)
                keep += [ri, ru]
        control = np.zeros((B, self.as_))
        cost = None if minimal else np.zeros((B, K))
        w = None if minimal else np.zeros((B, K))
        E = np.zeros(B * K * cs) if want_E else None
        ridx = None if minimal else np.zeros((B, max(N - 1, 1), K), dtype=np.int32)
        iters = np.zeros(B, dtype=np.int32)
        rc = self.L.mpopis_policy_step(self._h, C.byref(nz) if nz is not None else None, _d(control), _d(cost), _d(w),
                                       _d(E), _i(ridx), _i(iters))
        check(self._h, rc)
        out = dict(control=control, cost=cost, weights=w, iters_run=iters, res_idx0=ridx)
        if want_E:
            out["E"] = E.reshape(B, self.T, K, self.as_) if self.policy == "mppi" else E.reshape(B, K, cs)
        return out

    def env_step(self, action):
        a = _f64(action).reshape(self.B, self.as_)
        rew = np.zeros(self.B)
        check(self._h, self.L.mpopis_env_step(self._h, _d(a), _d(rew)))
        return rew

    def env_query(self):
        """(reward, within, dist, beta) of the resident envs without stepping."""
        nc = max(1, self.as_ // 2) if self.env_kind == "car" else 1
        rew = np.zeros(self.B)
        within = np.zeros(self.B, dtype=np.int32)
        dist = np.zeros((self.B, nc))
        beta = np.zeros((self.B, nc))
        check(self._h, self.L.mpopis_env_query(self._h, _d(rew), _i(within), _d(dist), _d(beta)))
        return rew, within.astype(bool), dist, beta

    def get_trajectories(self):
        out = np.zeros((self.B, self.K, self.ss, self.T))       # per sample: Julia (T x ss) column-major
        check(self._h, self.L.mpopis_get_trajectories(self._h, _d(out)))
        return out.transpose(0, 1, 3, 2)                        # -> (B, K, T, ss)

    # ---- Level 3 ----------------------------------------------------------------------------------
    def run_trials(self, num_steps=200, laps=2, log_actions=False):
        rec = np.zeros((self.B, _lib.RECORD_LEN))
        acts = np.zeros((self.B, num_steps + 1, self.as_)) if log_actions else None
        check(self._h, self.L.mpopis_run_trials(self._h, num_steps, laps, _d(rec), _d(acts)))
        return (rec, acts) if log_actions else rec

    # ---- measurement --------------------------------------------------------------------------------
    def timing_enable(self, on=True):
        check(self._h, self.L.mpopis_timing_enable(self._h, int(on)))

    def timing_reset(self):
        check(self._h, self.L.mpopis_timing_reset(self._h))

    def timing_read(self):
        names = C.create_string_buffer(256)
        ms = np.zeros(16)
        cnt = np.zeros(16, dtype=np.int64)
        n = C.c_int32(16)
        check(self._h, self.L.mpopis_timing_read(self._h, names, 256, _d(ms), cnt.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(n)))
        nm = names.value.decode().split(";")
        return {nm[i]: (float(ms[i]), int(cnt[i])) for i in range(n.value)}

    def bench_policy_steps(self, steps):
        ms = C.c_double()
        rl = C.c_double()
        check(self._h, self.L.mpopis_bench_policy_steps(self._h, steps, C.byref(ms), C.byref(rl)))
        return ms.value, rl.value
