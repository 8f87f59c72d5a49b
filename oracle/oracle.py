"""ctypes binding of the CPU oracle (oracle/mpopis_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package `mpopis_amd`.  PARITY UNPINNED (see the C header).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("MPOPIS_ORACLE_LIB") or os.path.join(_HERE, "libmpopis_oracle.so")      # (override: the sanitizer build of tests/dev/oracle_sanitizers.sh)

ENV_MOUNTAINCAR, ENV_CAR, ENV_CARTPOLE = 0, 1, 2
POL = dict(mppi=0, gmppi=1, imppi=2, cemppi=3, cmamppi=4, muaismppi=5, musigmaaismppi=6, pmcmppi=7)
POL.update({"μaismppi": 5, "μΣaismppi": 6})
SIGMA_EST = dict(mle=0, ss=1, lw=2, rblw=3, oas=4)
CP_N, MP_N = 20, 8


def build(force=False):
    if os.environ.get("MPOPIS_ORACLE_LIB"):
        return _LIB_PATH
    src = os.path.join(_HERE, "mpopis_oracle.c")
    hdr = os.path.join(_HERE, "mpopis_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libmpopis_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Env(C.Structure):
    _fields_ = [("kind", C.c_int), ("ncars", C.c_int), ("ss", C.c_int), ("as_", C.c_int),
                ("params", C.c_double * CP_N), ("P", C.c_int),
                ("tx", C.POINTER(C.c_double)), ("ty", C.POINTER(C.c_double)), ("tw", C.POINTER(C.c_double)),
                ("state", C.c_double * 64), ("t", C.c_int), ("done", C.c_int)]


class Policy(C.Structure):
    _fields_ = [("kind", C.c_int), ("K", C.c_int), ("T", C.c_int), ("as_", C.c_int), ("cs", C.c_int),
                ("ss", C.c_int), ("N", C.c_int),
                ("lambda_", C.c_double), ("alpha", C.c_double), ("lambda_ais", C.c_double),
                ("elite_threshold", C.c_double), ("cma_sigma", C.c_double),
                ("sigma_est", C.c_int), ("nthreads", C.c_int),
                ("U", C.POINTER(C.c_double)), ("Sigma", C.POINTER(C.c_double)),
                ("lo", C.c_double * 16), ("hi", C.c_double * 16),
                ("m_elite", C.c_int), ("ws", C.POINTER(C.c_double)),
                ("mu_eff", C.c_double), ("c_sigma", C.c_double), ("d_sigma", C.c_double),
                ("c_Sigma", C.c_double), ("c1", C.c_double), ("c_mu", C.c_double), ("E_cma", C.c_double)]


class Noise(C.Structure):
    _fields_ = [("Z", C.POINTER(C.c_double)), ("res_i0", C.POINTER(C.c_int32)), ("res_u", C.POINTER(C.c_double))]


class StepOut(C.Structure):
    _fields_ = [("control", C.POINTER(C.c_double)), ("cost", C.POINTER(C.c_double)),
                ("weights", C.POINTER(C.c_double)), ("E", C.POINTER(C.c_double)),
                ("res_idx0", C.POINTER(C.c_int32)), ("Sigma_last", C.POINTER(C.c_double)),
                ("U_last", C.POINTER(C.c_double)),
                ("iters_run", C.c_int), ("status", C.c_int)]


class TrialRecord(C.Structure):
    _fields_ = [("rew", C.c_double), ("steps", C.c_double), ("rew_per_step", C.c_double),
                ("lap_t", C.c_double * 4), ("mean_v", C.c_double), ("max_v", C.c_double),
                ("mean_beta", C.c_double), ("max_beta", C.c_double),
                ("beta_viol", C.c_double), ("trk_viol", C.c_double), ("crash_viol", C.c_double),
                ("rollouts", C.c_double)]


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_calc_tire_fy.restype = C.c_double
        L.orc_calc_tire_fy.argtypes = [C.c_double] * 5
        L.orc_calc_tire_fz.restype = C.c_double
        L.orc_calc_tire_fz.argtypes = [_dp, C.c_double, C.c_char]
        L.orc_car_reward.restype = C.c_double
        L.orc_calculate_beta.restype = C.c_double
        L.orc_env_reward.restype = C.c_double
        L.orc_rollout_model.restype = C.c_double
        L.orc_compute_weights.argtypes = [C.c_double, _dp, C.c_int, _dp]
        L.orc_m_elite.argtypes = [C.c_int, C.c_double]
        L.orc_sym_pow.argtypes = [C.c_int, _dp, C.c_double, _dp]
        L.orc_philox_normals.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, _dp]
        L.orc_philox_resample_draws.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _ip, _dp]
        L.orc_make_alias_table.argtypes = [_dp, C.c_double, C.c_int, _dp, _ip]
        L.orc_policy_create.argtypes = [C.POINTER(Policy), C.c_int, C.POINTER(Env), C.c_int, C.c_int,
                                        C.c_double, C.c_double, _dp, C.c_int, _dp, C.c_int, C.c_int,
                                        C.c_int, C.c_double, C.c_double, C.c_int, C.c_double]
        L.orc_run_trial.argtypes = [C.POINTER(Policy), C.POINTER(Env), C.c_uint64, C.c_int, C.c_int,
                                    C.POINTER(TrialRecord), _dp]
        L.orc_run_trial_noise.argtypes = [C.POINTER(Policy), C.POINTER(Env), C.c_uint64, C.c_int, C.c_int,
                                          C.c_double, C.c_double, C.c_double, C.POINTER(TrialRecord), _dp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def load_track(path=None, width=15.0):
    """48-point sub-sampled curve.csv centre line (car_racing_tracks.jl:14-34, sample_factor=20)."""
    if path is None:
        path = os.path.join(os.path.dirname(_HERE), "mpopis_amd", "data", "curve_sf20.csv")
    d = np.loadtxt(path, delimiter=",")
    return f64(d[:, 0]), f64(d[:, 1]), f64(np.full(d.shape[0], width))


def car_default_params():
    p = np.zeros(CP_N)
    lib().orc_car_default_params(_d(p))
    return p


def mountaincar_default_params():
    p = np.zeros(MP_N)
    lib().orc_mountaincar_default_params(_d(p))
    return p


def cartpole_default_params():
    p = np.zeros(11)
    lib().orc_cartpole_default_params(_d(p))
    return p


class OracleEnv:
    """Thin holder around orc_env (keeps the borrowed track arrays alive)."""

    def __init__(self, kind="car", ncars=1, params=None, track=None):
        self.track = track if track is not None else load_track()
        self.e = Env()
        k = {"car": ENV_CAR, "mountaincar": ENV_MOUNTAINCAR, "cartpole": ENV_CARTPOLE}[kind]
        pp = _d(f64(params)) if params is not None else None
        tx, ty, tw = self.track
        lib().orc_env_init(C.byref(self.e), k, ncars, pp, len(tx), _d(tx), _d(ty), _d(tw))

    @property
    def ss(self):
        return self.e.ss

    @property
    def as_(self):
        return self.e.as_

    @property
    def state(self):
        return np.array(self.e.state[: self.e.ss])

    @state.setter
    def state(self, s):
        s = f64(s)
        for i in range(self.e.ss):
            self.e.state[i] = s[i]

    @property
    def params(self):
        return np.array(self.e.params[:])

    def reset(self):
        lib().orc_env_reset(C.byref(self.e))

    def step(self, a):
        a = f64(a)
        return lib().orc_env_step(C.byref(self.e), _d(a))

    def reward(self):
        return lib().orc_env_reward(C.byref(self.e))

    def copy(self):
        o = OracleEnv.__new__(OracleEnv)
        o.track = self.track
        o.e = Env()
        C.memmove(C.byref(o.e), C.byref(self.e), C.sizeof(Env))
        return o


class OraclePolicy:
    def __init__(self, kind, env, K, T, lam=1.0, alpha=1.0, U0=None, cov=None, N=10, lam_ais=20.0,
                 elite_threshold=0.8, sigma_est="mle", cma_sigma=1.0, nthreads=1):
        self.env = env
        self.p = Policy()
        as_ = env.as_
        U0 = f64(np.zeros(as_) if U0 is None else U0)
        cov = f64(np.ones(as_) if cov is None else cov)
        is_vec = int(cov.ndim == 1)
        covc = np.asfortranarray(cov)
        rc = lib().orc_policy_create(C.byref(self.p), POL[kind], C.byref(env.e), K, T, lam, alpha,
                                     _d(U0), len(U0), covc.ctypes.data_as(_dp), covc.shape[0], is_vec,
                                     N, lam_ais, elite_threshold, SIGMA_EST[sigma_est], cma_sigma)
        if rc:
            raise ValueError("orc_policy_create failed: %d" % rc)
        self.p.nthreads = nthreads
        self.kind, self.K, self.T, self.as_, self.cs, self.N = kind, K, T, as_, self.p.cs, N

    def __del__(self):
        try:
            lib().orc_policy_free(C.byref(self.p))
        except Exception:
            pass

    @property
    def U(self):
        return np.ctypeslib.as_array(self.p.U, shape=(self.cs,)).copy()

    @U.setter
    def U(self, v):
        np.ctypeslib.as_array(self.p.U, shape=(self.cs,))[:] = f64(v)

    @property
    def Sigma(self):
        n = self.as_ if self.kind == "mppi" else self.cs
        return np.ctypeslib.as_array(self.p.Sigma, shape=(n, n)).T.copy()

    @Sigma.setter
    def Sigma(self, S):
        """pol.Σ = S (full matrix; what mpopis_set_Sigma does on the engine side)"""
        n = self.as_ if self.kind == "mppi" else self.cs
        S = f64(S).reshape(n, n)
        np.ctypeslib.as_array(self.p.Sigma, shape=(n, n))[:] = S.T

    @property
    def cma_ws(self):
        return np.ctypeslib.as_array(self.p.ws, shape=(self.K,)).copy()

    def n_iters(self):
        return 1 if self.kind in ("mppi", "gmppi") else self.N

    def noise_size(self):
        return self.K * self.T * self.as_ if self.kind == "mppi" else self.cs * self.K

    def simulate_model(self, Ucur, E, Sigma_inv=None, U_orig=None, log=False):
        """simulate_model(pol, env, E, Sigma_inv, U_orig); E is cs x K (numpy, column k = sample k)."""
        Ucur = f64(Ucur)
        U_orig = f64(Ucur if U_orig is None else U_orig)
        Ecm = np.ascontiguousarray(np.asarray(E, dtype=np.float64).T)  # (K, cs) C-order == cs x K col-major
        cost = np.zeros(self.K)
        Si = None if Sigma_inv is None else np.ascontiguousarray(np.asarray(Sigma_inv, dtype=np.float64).T)
        traj = np.zeros((self.K, self.T, self.env.ss)) if log else None
        lib().orc_simulate_model(C.byref(self.p), _d(Ucur), C.byref(self.env.e), _d(Ecm),
                                 _d(Si) if Si is not None else None, _d(U_orig), _d(cost),
                                 _d(traj) if log else None)
        return (cost, traj) if log else cost

    def __call__(self, env, Z, res_i0=None, res_u=None, want_extra=False):
        """pol(env) with injected noise.  Z: (N, K, cs) [row k = sample k] for G-variants, or
        (T, K, as) for mppi.  Returns dict(control, cost, weights, E, iters_run, status, ...)."""
        K, cs = self.K, self.cs
        Z = f64(Z)
        nz = Noise()
        nz.Z = _d(Z)
        if res_i0 is not None:
            res_i0 = np.ascontiguousarray(res_i0, dtype=np.int32)
            res_u = f64(res_u)
            nz.res_i0, nz.res_u = _i(res_i0), _d(res_u)
        out = StepOut()
        control = np.zeros(self.as_)
        cost = np.zeros(K)
        w = np.zeros(K)
        E = np.zeros(self.noise_size())
        nN = max(self.n_iters() - 1, 1)
        ridx = np.zeros(nN * K, dtype=np.int32)
        n = self.as_ if self.kind == "mppi" else cs
        Slast = np.zeros(n * n)
        Ulast = np.zeros(cs)
        out.control, out.cost, out.weights, out.E = _d(control), _d(cost), _d(w), _d(E)
        out.res_idx0 = _i(ridx)
        if self.kind != "mppi":
            out.Sigma_last, out.U_last = _d(Slast), _d(Ulast)
        st = lib().orc_policy_call(C.byref(self.p), C.byref(env.e), C.byref(nz), C.byref(out))
        r = dict(control=control, cost=cost, weights=w, iters_run=out.iters_run, status=st)
        if self.kind == "mppi":
            r["E"] = E.reshape(self.T, K, self.as_)
        else:
            r["E"] = E.reshape(K, cs).T.copy()        # cs x K
            r["Sigma_last"] = Slast.reshape(cs, cs).T.copy()
            r["U_last"] = Ulast
            r["res_idx0"] = ridx.reshape(nN, K)
        return r

    def run_trial(self, env, seed, num_steps=200, laps=2, log_actions=False, state_noise=(0.0, 0.0, 0.0)):
        rec = TrialRecord()
        acts = np.zeros((num_steps + 1, self.as_)) if log_actions else None
        st = lib().orc_run_trial_noise(C.byref(self.p), C.byref(env.e), seed, num_steps, laps,
                                       float(state_noise[0]), float(state_noise[1]), float(state_noise[2]), C.byref(rec),
                                       _d(acts) if log_actions else None)
        d = {f: getattr(rec, f) for f, _ in TrialRecord._fields_ if f != "lap_t"}
        d["lap_t"] = list(rec.lap_t)
        d["status"] = st
        if log_actions:
            d["actions"] = acts
        return d


def philox_normals(seed, stream_lo, stream_hi, n):
    out = np.zeros(n)
    lib().orc_philox_normals(seed, stream_lo, stream_hi, n, _d(out))
    return out


def philox_resample_draws(seed, stream_lo, stream_hi, K):
    i0 = np.zeros(K, dtype=np.int32)
    u = np.zeros(K)
    lib().orc_philox_resample_draws(seed, stream_lo, stream_hi, K, _i(i0), _d(u))
    return i0, u


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


def compute_weights(lam, cost):
    cost = f64(cost)
    w = np.zeros(len(cost))
    lib().orc_compute_weights(lam, _d(cost), len(cost), _d(w))
    return w


def make_alias_table(w, wsum=1.0):
    w = f64(w)
    a = np.zeros(len(w))
    al = np.zeros(len(w), dtype=np.int32)
    lib().orc_make_alias_table(_d(w), wsum, len(w), _d(a), _i(al))
    return a, al


def alias_sample(accept, alias0, di, du):
    accept, du = f64(accept), f64(du)
    alias0 = np.ascontiguousarray(alias0, dtype=np.int32)
    di = np.ascontiguousarray(di, dtype=np.int32)
    out = np.zeros(len(di), dtype=np.int32)
    lib().orc_alias_sample(_d(accept), _i(alias0), len(accept), _i(di), _d(du), len(di), _i(out))
    return out


def within_track(track, pos):
    tx, ty, tw = track
    pos = f64(pos)
    dist = C.c_double()
    w = lib().orc_within_track(len(tx), _d(tx), _d(ty), _d(tw), _d(pos), C.byref(dist))
    return bool(w), dist.value


def car_step(params, state, action):
    s = f64(state).copy()
    lib().orc_car_step(_d(f64(params)), _d(s), _d(f64(action)))
    return s


def cholesky_lower(A):
    A = np.asarray(A, dtype=np.float64)
    n = A.shape[0]
    Af = np.ascontiguousarray(A.T)
    L = np.zeros(n * n)
    rc = lib().orc_cholesky_lower(n, _d(Af), _d(L))
    return rc, L.reshape(n, n).T.copy()


def sym_pow(A, p):
    A = np.asarray(A, dtype=np.float64)
    n = A.shape[0]
    Af = np.ascontiguousarray(A.T)
    o = np.zeros(n * n)
    rc = lib().orc_sym_pow(n, _d(Af), p, _d(o))
    return rc, o.reshape(n, n).T.copy()


SIGMA_EST = {"mle": 0, "ss": 1, "lw": 2, "rblw": 3, "oas": 4}


def cov_estimate(X, est):
    """cov(Σ_est, X') for X of shape (cs, m) (columns = observations, like the elite matrix :464) -> (mean, S)."""
    X = np.asarray(X, dtype=np.float64)
    cs, m = X.shape
    Xf = np.ascontiguousarray(X.T)                       # cs x m column-major
    mean, S = np.zeros(cs), np.zeros(cs * cs)
    rc = lib().orc_cov_estimate(cs, m, _d(Xf), SIGMA_EST[est], _d(mean), _d(S))
    assert rc == 0
    return mean, S.reshape(cs, cs).T.copy()


def block_diagm(A, rep):
    A = np.asarray(A, dtype=np.float64)
    if A.ndim == 1:
        A = np.diag(A)
    r = A.shape[0]
    B = np.zeros((r * rep) ** 2)
    lib().orc_block_diagm(_d(np.ascontiguousarray(A.T)), r, rep, _d(B))
    return B.reshape(r * rep, r * rep).T.copy()


def m_elite(K, thr):
    return lib().orc_m_elite(K, thr)


def quantile_ci(x):
    x = f64(x)
    lo, med, hi = C.c_double(), C.c_double(), C.c_double()
    lib().orc_quantile_ci(_d(x), len(x), C.byref(lo), C.byref(med), C.byref(hi))
    return lo.value, med.value, hi.value
