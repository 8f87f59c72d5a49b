"""Host-side mirrors of the reference env types.  They hold the reference's fields (state, done,
t, params, track) and delegate every computation to the HIP engine through the C ABI: stepping and
reward are the device kernels (mpopis_env_step / mpopis_env_query); nothing is evaluated on the CPU.

  CarRacingEnv        src/envs/car_racing.jl:28-150
  MultiCarRacingEnv   src/envs/multi-car_racing.jl:2-64
  MountainCarEnv      RL.jl MountainCarEnv(continuous=true) + src/examples/mountaincar_example.jl:4-22
  CartPoleEnv         RL.jl CartPoleEnv(continuous=true) + src/examples/cartpole_example.jl:3-6
  Track               src/envs/car_racing_tracks/car_racing_tracks.jl:2-34
"""
import math
import os
import numpy as np

from .engine import Engine, default_track, BUNDLED_TRACKS, _f64
from ._lib import MPOPISError, ERR_ARG


def deg2rad(d):
    return d * (math.pi / 180.0)


class CarRacingEnvParams:
    """src/envs/car_racing.jl:2-21 (defaults :68-93)."""
    FIELDS = ("m", "Izz", "h_cm", "l_f", "l_r", "C_D0", "C_D1", "C_αf", "C_αr", "μ_f", "μ_r", "δ_max", "δ_dot_max",
              "Fx_max", "Fx_min", "λ_brake", "λ_drive", "β_limit")

    def __init__(self, m=2000.0, I_zz=3764.0, h_cm=0.3, l_f=1.53, l_r=1.23, C_D0=241.0, C_D1=25.1, C_αf=150000.0,
                 C_αr=280000.0, μ_f=0.9, μ_r=0.9, δ_max=deg2rad(18.0), δ_dot_max=deg2rad(90.0), Fx_max=7200.0,
                 Fx_min=22500.0, λ_brake=0.6, λ_drive=0.0, β_limit=deg2rad(45.0)):
        self.values = [float(v) for v in (m, I_zz, h_cm, l_f, l_r, C_D0, C_D1, C_αf, C_αr, μ_f, μ_r, δ_max, δ_dot_max,
                                          Fx_max, Fx_min, λ_brake, λ_drive, β_limit)]

    def __getattr__(self, name):
        if name in CarRacingEnvParams.FIELDS:
            return self.values[CarRacingEnvParams.FIELDS.index(name)]
        raise AttributeError(name)

    def vector(self, dt, δt):
        return np.array(self.values + [dt, δt], dtype=np.float64)


class Track:
    """Track(infile; width=15.0, sample_factor=20): x, y, lane_width and the sub-sampled x′, y′, lane_width′
    (car_racing_tracks.jl:14-34).  `infile=None` is the reference default curve.csv at sample_factor 20,
    shipped pre-sampled (mpopis_amd/data/curve_sf20.csv); the stem of any other reference track file
    ("curve1".."curve5", "cubic", "cubic1".."cubic5") selects its pre-sampled fixture the same way."""

    def __init__(self, infile=None, width=15.0, sample_factor=20):
        if infile is None or infile in BUNDLED_TRACKS:
            if sample_factor != 20 or not np.isscalar(width):
                raise MPOPISError(ERR_ARG, "the bundled tracks are pre-sampled at sample_factor=20 with a constant width; pass a CSV path for others")
            self.x = self.y = self.lane_width = None
            self.xp, self.yp, self.wp = default_track(width, infile or "curve")
        else:
            d = np.loadtxt(infile, delimiter=",")
            if d.ndim != 2 or d.shape[1] != 2:
                raise MPOPISError(ERR_ARG, "Can only have 2 columns for a track file")      # :16
            w = np.full(d.shape[0], width) if np.isscalar(width) else _f64(width)
            if len(w) != d.shape[0]:
                raise MPOPISError(ERR_ARG, "Supplied width vector does not match length of track file")   # :17
            self.x, self.y, self.lane_width = _f64(d[:, 0]), _f64(d[:, 1]), w
            self.xp, self.yp, self.wp = _f64(d[::sample_factor, 0]), _f64(d[::sample_factor, 1]), _f64(w[::sample_factor])
        self.sample_factor = sample_factor

    def arrays(self):
        return self.xp, self.yp, self.wp


class _EnvBase:
    """Shared plumbing: a private 1-slot engine that owns the resident copy of this env."""
    kind, ncars = "car", 1

    def _mk_engine(self, device=0):
        self._eng = Engine(self.kind, self.ncars, "gmppi", num_samples=1, horizon=1, batch=1, lam=1.0,
                           device=device, track=self.track.arrays() if self.kind == "car" else None,
                           env_params=self._param_vector())
        self._push()

    def _push(self):
        self._eng.set_state(self.state[None], [self.t], [int(self.done)])

    def _pull(self):
        x, t, d = self._eng.get_state()
        self.state, self.t, self.done = x[0].copy(), int(t[0]), bool(d[0])

    def __call__(self, a):
        """env(action): the functor of the reference (car_racing.jl:238-250; multi :200-216; mountaincar_example.jl:4-7)."""
        a = _f64(a).reshape(-1)
        if a.size != self.as_:
            raise MPOPISError(ERR_ARG, "Only implented for one step" if a.size > self.as_ else "Action space of each car is of size 2")
        self._push()
        self._last_reward = float(self._eng.env_step(a[None])[0])
        self._pull()
        return self

    def copy(self):
        import copy as _c
        o = _c.copy(self)
        o.state = self.state.copy()
        o._mk_engine(self._eng_device)
        return o


class CarRacingEnv(_EnvBase):
    def __init__(self, params=None, dt=0.1, δt=0.01, track=None, track_sample_factor=20, rng=None, device=0, **param_kw):
        self.params = params if params is not None else CarRacingEnvParams(**param_kw)
        self.dt, self.δt = dt, δt
        self.track = track if isinstance(track, Track) else Track(track, sample_factor=track_sample_factor)
        self.rng = rng
        self.kind, self.ncars, self.as_, self.ss = "car", 1, 2, 8
        self._eng_device = device
        self.reset(_make=False)
        self._mk_engine(device)

    def _param_vector(self):
        return self.params.vector(self.dt, self.δt)

    def reset(self, state=None, _make=True):
        """reset!(env) / reset!(env, state): car_racing.jl:215-230."""
        if state is None:
            self.state = np.zeros(8)
            self.state[2] = deg2rad(90.0)
            self.state[3] = 10.0
        else:
            self.state = _f64(state).copy()
        self.t, self.done = 0, False
        if _make:
            self._push()


class MultiCarRacingEnv(_EnvBase):
    def __init__(self, N=2, dt=0.1, δt=0.01, track=None, rng=None, device=0):
        if not 1 <= N <= 4:
            raise MPOPISError(ERR_ARG, "this engine supports 1..4 cars")
        self.N = N
        self.params = CarRacingEnvParams()
        self.dt, self.δt = dt, δt
        self.track = track if isinstance(track, Track) else Track(track)
        self.rng = rng
        self.kind, self.ncars, self.as_, self.ss = "car", N, 2 * N, 8 * N
        self._eng_device = device
        self.reset(_make=False)
        self._mk_engine(device)

    def _param_vector(self):
        return self.params.vector(self.dt, self.δt)

    @property
    def envs(self):
        """Per-car views (state slices), like env.envs[i].state."""
        return [self.state[8 * i:8 * i + 8] for i in range(self.N)]

    def reset(self, state=None, _make=True):
        """multi-car_racing.jl:160-188."""
        if state is None:
            s = np.zeros(8 * self.N)
            for c in range(self.N):
                ii = c + 1
                if ii >= 2:
                    s[8 * c] = ii / 2 * 5.0 if ii % 2 == 0 else (1 - ii) / 2 * 5.0
                s[8 * c + 2] = deg2rad(90.0)
                s[8 * c + 3] = 10.0
            self.state = s
        else:
            self.state = _f64(state).copy()
        self.t, self.done = 0, False
        if _make:
            self._push()


class MountainCarEnvParams:
    def __init__(self, min_pos=-1.2, max_pos=0.6, max_speed=0.07, goal_pos=0.45, goal_velocity=0.0, power=0.0015,
                 gravity=0.0025, max_steps=200):
        self.min_pos, self.max_pos, self.max_speed, self.goal_pos = min_pos, max_pos, max_speed, goal_pos
        self.goal_velocity, self.power, self.gravity, self.max_steps = goal_velocity, power, gravity, max_steps

    def vector(self):
        return np.array([self.min_pos, self.max_pos, self.max_speed, self.goal_pos, self.goal_velocity, self.power,
                         self.gravity, float(self.max_steps)])


class MountainCarEnv(_EnvBase):
    """MountainCarEnv(continuous=true, rng=...).  The reference's constructor draws x0 ~ U(-0.6,-0.4) from an
    unseeded RNG; here x0 is explicit (default -0.5) or drawn from `rng` if given."""

    def __init__(self, continuous=True, rng=None, x0=None, device=0, **param_kw):
        if not continuous:
            raise MPOPISError(ERR_ARG, "only the continuous MountainCar of the reference examples is supported")
        self.params = MountainCarEnvParams(**param_kw)
        self.rng = rng
        self.kind, self.ncars, self.as_, self.ss = "mountaincar", 0, 1, 2
        self.track = None
        self._x0 = x0
        self._eng_device = device
        self.reset(_make=False)
        self._mk_engine(device)

    def _param_vector(self):
        return self.params.vector()

    def reset(self, state=None, _make=True):
        if state is None:
            x0 = self._x0 if self._x0 is not None else (0.2 * self.rng.random() - 0.6 if self.rng is not None else -0.5)
            self.state = np.array([x0, 0.0])
        else:
            self.state = _f64(state).copy()
        self.t, self.done = 0, False
        if _make:
            self._push()


class CartPoleEnvParams:
    """RL.jl CartPoleEnvParams (third-party, recalled)."""

    def __init__(self, gravity=9.8, masscart=1.0, masspole=0.1, halflength=0.5, forcemag=10.0, dt=0.02,
                 thetathreshold=12 * 2 * math.pi / 360, xthreshold=2.4, max_steps=200):
        self.gravity, self.masscart, self.masspole, self.halflength = gravity, masscart, masspole, halflength
        self.totalmass, self.polemasslength = masspole + masscart, masspole * halflength
        self.forcemag, self.dt, self.thetathreshold, self.xthreshold, self.max_steps = forcemag, dt, thetathreshold, xthreshold, max_steps

    def vector(self):
        return np.array([self.gravity, self.masscart, self.masspole, self.totalmass, self.halflength, self.polemasslength,
                         self.forcemag, self.dt, self.thetathreshold, self.xthreshold, float(self.max_steps)])


class CartPoleEnv(_EnvBase):
    """CartPoleEnv(continuous=true, rng=...) driven by the functor of src/examples/cartpole_example.jl:3-6.
    State [x, xdot, theta, thetadot]; the reference's reset draws 0.1*rand(4) - 0.05 from `rng`
    (here: from `rng` if given, else the centre of that box, or an explicit `x0`)."""

    def __init__(self, continuous=True, rng=None, x0=None, device=0, **param_kw):
        if not continuous:
            raise MPOPISError(ERR_ARG, "only the continuous CartPole of the reference examples is supported")
        self.params = CartPoleEnvParams(**param_kw)
        self.rng = rng
        self.kind, self.ncars, self.as_, self.ss = "cartpole", 0, 1, 4
        self.track = None
        self._x0 = x0
        self._eng_device = device
        self.reset(_make=False)
        self._mk_engine(device)

    def _param_vector(self):
        return self.params.vector()

    def reset(self, state=None, _make=True):
        if state is None:
            if self._x0 is not None:
                self.state = _f64(self._x0).copy()
            else:
                self.state = 0.1 * self.rng.random(4) - 0.05 if self.rng is not None else np.zeros(4)
        else:
            self.state = _f64(state).copy()
        self.t, self.done = 0, False
        if _make:
            self._push()


# ---- RLBase-style free functions used by the reference's callers ------------------------------------
def state(env):
    return env.state


def is_terminated(env):
    return env.done


def action_space(env):
    """(leftendpoint, rightendpoint): car_racing.jl:156-159; multi-car_racing.jl:75-84; RL.jl -1.0..1.0"""
    return -np.ones(env.as_), np.ones(env.as_)


def reward(env):
    """reward(env) of the CURRENT state (car_racing.jl:201-213; multi :145-158; mountaincar_example.jl:10-22)."""
    env._push()
    return float(env._eng.env_query()[0][0])


def within_track(env):
    """CarRacingEnv: (within, dist) NamedTuple-like (car_racing.jl:178-180); MultiCarRacingEnv: Bool (:122-128)."""
    env._push()
    _, w, d, _ = env._eng.env_query()
    return (bool(w[0]), float(d[0, 0])) if env.ncars == 1 else bool(w[0])


def calculate_β(env):
    return math.atan2(env.state[4], env.state[3])                 # car_racing.jl:181-183 (pure accessor)


def exceed_β(env):
    if env.ncars == 1:
        return abs(calculate_β(env)) > env.params.β_limit         # :184-189
    return any(abs(math.atan2(s[4], s[3])) > env.params.β_limit for s in env.envs)   # multi :130-136
