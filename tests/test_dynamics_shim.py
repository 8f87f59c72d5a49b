"""CPU check (no GPU) of the engine's model step -- mpopis_amd/csrc/car_dynamics.h: car_action_step = ten car_substep's with the HOT force rules (forward, front
slip in the forward half plane) and the GENERAL rules (stopped, sliding / rolling backwards, NaN) sharing one integration tail -- against the oracle's literal
CarRacingEnv functor + _step! (src/envs/car_racing.jl:238-250,282-344: atan2 / tan / sincos per Euler sub-step).  The header is compiled for the host
(tests/shim/host_shim.cpp, test infrastructure only); the device build differs from it only in the rcp / rsq seeds of its divisions and square roots.
What this pins without a GPU: a regression of the general force rules (the cold lanes of a rollout that brakes to a standstill) shows up here as a state
deviation far above rounding -- tests/test_gpu_standstill.py and the fuzz sweep only see it through costs and controls.  States: driving, crawling, exactly
stopped, rolling backwards, spinning (|beta| large), full brake through Vx = 0, steering at the stops; actions incl. the clamp values."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM_SRC = os.path.join(HERE, "shim", "host_shim.cpp")
SHIM_SO = os.path.join(HERE, "shim", "libhost_shim.so")
HDR = os.path.join(os.path.dirname(HERE), "mpopis_amd", "csrc", "car_dynamics.h")
dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def shim():
    if (not os.path.exists(SHIM_SO)) or os.path.getmtime(SHIM_SO) < max(os.path.getmtime(SHIM_SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-o", SHIM_SO, SHIM_SRC])
    L = C.CDLL(SHIM_SO)
    L.shim_car_action_step.argtypes = [dp, dp, C.c_double, C.c_double]
    L.shim_car_action_step.restype = None
    return L


def _states(rng, n, regime):
    s = np.zeros((n, 8))
    s[:, 0] = rng.uniform(-50, 50, n); s[:, 1] = rng.uniform(-50, 50, n); s[:, 2] = rng.uniform(-np.pi, np.pi, n)
    s[:, 6] = rng.uniform(-0.45, 0.45, n); s[:, 7] = rng.uniform(-1, 1, n)
    if regime == "driving":
        s[:, 3] = rng.uniform(2.0, 35.0, n); s[:, 4] = rng.uniform(-1.5, 1.5, n); s[:, 5] = rng.uniform(-0.8, 0.8, n)
    elif regime == "crawling":
        s[:, 3] = rng.uniform(1e-3, 1.2, n); s[:, 4] = rng.uniform(-0.3, 0.3, n); s[:, 5] = rng.uniform(-0.3, 0.3, n)
    elif regime == "stopped":
        s[:, 3] = 0.0; s[:, 4] = np.where(rng.random(n) < 0.5, 0.0, rng.uniform(-0.2, 0.2, n)); s[:, 5] = np.where(rng.random(n) < 0.5, 0.0, rng.uniform(-0.2, 0.2, n))
    elif regime == "backwards":
        s[:, 3] = -rng.uniform(1e-3, 6.0, n); s[:, 4] = rng.uniform(-1.0, 1.0, n); s[:, 5] = rng.uniform(-0.6, 0.6, n)
    elif regime == "spinning":
        s[:, 3] = rng.uniform(-3.0, 8.0, n); s[:, 4] = rng.uniform(-8.0, 8.0, n); s[:, 5] = rng.uniform(-3.0, 3.0, n)
    return s


@pytest.mark.parametrize("regime", ["driving", "crawling", "stopped", "backwards", "spinning"])
def test_model_step_matches_the_literal_reference_step(shim, oracle, regime):
    rng = np.random.default_rng({"driving": 1, "crawling": 2, "stopped": 3, "backwards": 4, "spinning": 5}[regime])
    p = oracle.car_default_params()
    n = 1500
    S = _states(rng, n, regime)
    A = rng.uniform(-1, 1, (n, 2))
    A[rng.random(n) < 0.15, 1] = -1.0                           # full brake (through Vx = 0 from the crawling states)
    A[rng.random(n) < 0.10, 1] = 1.0
    A[rng.random(n) < 0.10, 0] = rng.choice([-1.0, 1.0])
    A[rng.random(n) < 0.05] = 0.0
    worst, flips = 0.0, 0
    scale = np.array([1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0])
    for i in range(n):
        ref = oracle.car_step(p, S[i], A[i])
        got = S[i].copy()
        shim.shim_car_action_step(p.ctypes.data_as(dp), got.ctypes.data_as(dp), float(A[i, 0]), float(A[i, 1]))
        # three or four sub-steps later the two implementations may disagree about sign(Vx) when Vx passes within rounding of zero: the
        # reference's brake force flips there (car_racing.jl:311) and the step's outcome differs by a whole sub-step's impulse -- counted, bounded
        d = np.abs(got - ref) / np.maximum(scale, np.abs(ref))
        d[2] = min(d[2], abs(abs(got[2] - ref[2]) - 2 * np.pi))  # the heading's wrap to (-pi, pi] may land on either side at exactly +-pi
        if d.max() > 1e-9:
            flips += 1
            continue
        worst = max(worst, float(d.max()))
    print("\n[dynamics shim] %s: worst relative state deviation %.2e over %d states, %d set aside (sign(Vx) decided within rounding of zero)" % (regime, worst, n - flips, flips))
    assert worst < 1e-11
    assert flips <= (n // 100 if regime in ("crawling", "stopped") else 0)


def test_nan_action_poisons_the_state_like_the_reference(shim, oracle):
    """a NaN action is the reference's "Action is not in action space" error (car_racing.jl:239); in a rollout it must propagate into the state (and from
    there into the cost) instead of being clamped away"""
    p = oracle.car_default_params()
    s = np.array([0.0, 0.0, 0.5, 10.0, 0.1, 0.05, 0.02, 0.3])
    for a in ((float("nan"), 0.2), (0.1, float("nan"))):
        got = s.copy()
        shim.shim_car_action_step(p.ctypes.data_as(dp), got.ctypes.data_as(dp), a[0], a[1])
        assert np.isnan(got[[0, 1, 3, 4, 5]]).any()


def test_reward_matches_the_reference_reward(shim, oracle):
    """car_reward (ratio test for |beta| > beta_limit, squared-distance neighbour choice, rsq / rcp-free on the host) against the oracle's literal reward
    (src/envs/car_racing.jl:201-213: atan2, findmin, norms) on the default track: on the road, off the road (-1e6), beyond the slip-angle limit (-5000),
    slow and reversed cars."""
    shim.shim_car_reward.argtypes = [dp, C.c_int, dp, dp, dp, dp]
    shim.shim_car_reward.restype = C.c_double
    p = oracle.car_default_params()
    track = oracle.load_track()
    tx, ty, tw = (np.ascontiguousarray(a, dtype=np.float64) for a in track)
    env = oracle.OracleEnv("car", 1, track=track)
    rng = np.random.default_rng(9)
    worst, seen = 0.0, set()
    for i in range(4000):
        j = int(rng.integers(0, len(tx)))
        off = rng.normal(0.0, 6.0, 2) if i % 4 else rng.normal(0.0, 25.0, 2)          # mostly inside the lane, a quarter far off
        s = np.array([tx[j] + off[0], ty[j] + off[1], rng.uniform(-np.pi, np.pi), rng.uniform(-3, 30), rng.uniform(-6, 6), rng.uniform(-1, 1), rng.uniform(-0.4, 0.4), rng.uniform(-1, 1)])
        env.state = s
        ref = env.reward()
        got = shim.shim_car_reward(p.ctypes.data_as(dp), len(tx), tx.ctypes.data_as(dp), ty.ctypes.data_as(dp), tw.ctypes.data_as(dp), s.ctypes.data_as(dp))
        seen.add((ref < -9e5, -9e5 <= ref < -4000))
        worst = max(worst, abs(got - ref) / max(1.0, abs(ref)))
    assert worst < 1e-12, worst
    assert len(seen) >= 3                                              # on the road, off the road and beyond the slip limit all occurred


def test_mountaincar_step_and_reward_match_the_oracle(shim, oracle):
    """mc_step / mc_reward (car_dynamics.h) against the oracle's MountainCar (RL.jl MountainCarEnv(continuous = true) + the reward override of
    src/examples/mountaincar_example.jl:4-22) along closed-loop runs with random forces, through the goal and the step limit."""
    shim.shim_mc_step.argtypes = [dp, dp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_double]
    shim.shim_mc_step.restype = None
    shim.shim_mc_reward.argtypes = [dp, dp, C.c_int]
    shim.shim_mc_reward.restype = C.c_double
    p = oracle.mountaincar_default_params()
    rng = np.random.default_rng(4)
    done_seen = 0
    for run in range(40):
        env = oracle.OracleEnv("mountaincar")
        s = env.state.copy()
        t, done = C.c_int(0), C.c_int(0)
        for step in range(250):
            f = float(np.clip(rng.normal(0.6 if run % 2 else 0.0, 0.8), -1, 1))
            env.step([f])
            shim.shim_mc_step(p.ctypes.data_as(dp), s.ctypes.data_as(dp), C.byref(t), C.byref(done), f)
            assert np.max(np.abs(s - env.state)) < 1e-14
            r = shim.shim_mc_reward(p.ctypes.data_as(dp), s.ctypes.data_as(dp), done.value)
            assert abs(r - env.reward()) <= 1e-12 * max(1.0, abs(env.reward()))
            if done.value:
                done_seen += 1
                break
    assert done_seen >= 5


def test_model_step_with_random_car_parameters(shim, oracle):
    """mpopis_set_env_params takes any CarRacingEnvParams / dt / δt (car_racing.jl:2-21,33-34): the reformulated step (hoisted tyre constants, carried sin / cos,
    small-angle rotations with a library path beyond 1/32 rad, the 2-per-trip sub-step loop) against the literal step under RANDOM parameters -- masses, axle
    geometry, tyre stiffness and friction, steering limits and rates up to the library path, brake / drive splits, slip limits, 1 ... 20 sub-steps incl. odd
    counts -- in all five state regimes."""
    rng = np.random.default_rng(77)
    base = oracle.car_default_params()
    worst, n_total = 0.0, 0
    for trial in range(300):
        p = base.copy()
        p[0] *= rng.uniform(0.6, 1.6); p[1] *= rng.uniform(0.6, 1.6)                    # m, Izz
        p[2] *= rng.uniform(0.5, 1.5)                                                    # h
        p[3] *= rng.uniform(0.8, 1.25); p[4] *= rng.uniform(0.8, 1.25)                   # lf, lr
        p[5] *= rng.uniform(0.0, 2.0); p[6] *= rng.uniform(0.0, 2.0)                     # CD0, CD1
        p[7] *= rng.uniform(0.5, 1.8); p[8] *= rng.uniform(0.5, 1.8)                     # Caf, Car
        p[9] = rng.uniform(0.4, 1.2); p[10] = rng.uniform(0.4, 1.2)                      # mu_f, mu_r
        p[11] = np.deg2rad(rng.uniform(10.0, 45.0))                                      # delta_max
        p[12] = np.deg2rad(rng.choice([30.0, 90.0, 150.0, 400.0, 900.0]))                # delta_dot_max (the last two: beyond the small-angle range per sub-step)
        p[13] *= rng.uniform(0.5, 1.5); p[14] *= rng.uniform(0.5, 1.5)                   # Fx_max, Fx_min
        p[15] = rng.uniform(0.3, 0.9); p[16] = rng.uniform(0.0, 1.0)                     # lambda_brake, lambda_drive
        p[17] = np.deg2rad(rng.uniform(15.0, 80.0))                                      # beta_limit
        nsub = int(rng.choice([1, 2, 3, 5, 7, 10, 13, 20]))
        p[19] = float(rng.choice([0.005, 0.01, 0.02])); p[18] = nsub * p[19]             # delta_t, dt
        for regime in ("driving", "crawling", "stopped", "backwards", "spinning"):
            S = _states(rng, 6, regime)
            S[:, 6] = rng.uniform(-0.9, 0.9, 6) * p[11]                                  # steering angle inside its limits
            A = rng.uniform(-1, 1, (6, 2))
            for i in range(6):
                ref = oracle.car_step(p, S[i], A[i])
                got = S[i].copy()
                shim.shim_car_action_step(p.ctypes.data_as(dp), got.ctypes.data_as(dp), float(A[i, 0]), float(A[i, 1]))
                d = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
                d[2] = min(d[2], abs(abs(got[2] - ref[2]) - 2 * np.pi))
                if d.max() > 1e-9 and regime in ("crawling", "stopped"):                 # sign(Vx) decided within rounding of zero (see above)
                    continue
                worst = max(worst, float(d.max())); n_total += 1
    print("\n[dynamics shim] random parameters: worst relative state deviation %.2e over %d steps" % (worst, n_total))
    assert worst < 1e-10 and n_total > 8500
