"""Engine (through the C ABI) vs the oracle at the EDGES of the parameter ranges -- what the randomised sweep does not draw: degenerate elite sets (none, one,
all of the samples), a single sample, extreme lambda, alpha != 1 under adaptive policies, a one-step horizon.  The two must agree on the outcome class
(status code: -2 where the reference's MvNormal throws PosDefException, -1 where its delta_s[order[ii]] indexing throws BoundsError, :593) and, where the call
succeeds, on iteration count and control."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    dict(kind="cemppi", K=150, T=10, N=4, elite_threshold=1.0),                 # no elite sample: mean of nothing -> NaN covariance -> PosDefException
    dict(kind="cemppi", K=150, T=10, N=4, elite_threshold=0.995),               # one elite sample
    dict(kind="cemppi", K=150, T=10, N=4, elite_threshold=0.0),                 # every sample is elite
    dict(kind="cemppi", K=150, T=10, N=4, elite_threshold=0.5, sigma_est="oas"),
    dict(kind="cmamppi", K=150, T=10, N=3, elite_threshold=0.99),               # cs * m_elite < K: BoundsError (:593)
    dict(kind="cmamppi", K=150, T=10, N=3, elite_threshold=0.0),
    dict(kind="musigmaaismppi", K=2, T=10, N=3),
    dict(kind="musigmaaismppi", K=1, T=10, N=3),
    dict(kind="pmcmppi", K=2, T=10, N=3),
    dict(kind="pmcmppi", K=1, T=5, N=3),                                        # corrected covariance of one column: 0/0 -> PosDefException
    dict(kind="musigmaaismppi", K=256, T=10, N=3, lam=1e-6, lam_ais=1e-6),      # the weights collapse onto the best sample
    dict(kind="musigmaaismppi", K=256, T=10, N=3, lam=1e12, lam_ais=1e12),      # uniform weights
    dict(kind="muaismppi", K=256, T=10, N=3, alpha=0.3),                        # control-cost term (gamma != 0) through the AIS iterations
    dict(kind="cemppi", K=256, T=10, N=3, alpha=0.0),
    dict(kind="gmppi", K=64, T=1, N=1),                                         # horizon 1: the roll of pol.U degenerates (utils.jl:98)
    dict(kind="imppi", K=64, T=1, N=3),
]


@pytest.mark.parametrize("c", CASES, ids=["%s-%s" % (c["kind"], "-".join("%s=%s" % kv for kv in c.items() if kv[0] not in ("kind", "T", "N"))) for c in CASES])
def test_edge_configuration_agrees_with_the_oracle(oracle, track, c):
    from mpopis_amd import build
    build.build()
    from mpopis_amd.engine import Engine
    from mpopis_amd._lib import MPOPISError
    kind, K, T, N = c["kind"], c["K"], c["T"], c["N"]
    kw = {k: v for k, v in c.items() if k not in ("kind", "K", "T", "N")}
    lam, lam_ais, alpha = kw.pop("lam", 10.0), kw.pop("lam_ais", 20.0), kw.pop("alpha", 1.0)
    cs = 2 * T
    rng = np.random.default_rng(1000 + K + T)
    Z = rng.standard_normal((1, N if kind != "gmppi" else 1, K, cs))
    di = rng.integers(0, K, (1, max(N - 1, 1), K)).astype(np.int32)
    du = rng.random((1, max(N - 1, 1), K))
    env = oracle.OracleEnv("car", 1, track=track)
    pol = oracle.OraclePolicy(kind, env, K, T, lam=lam, alpha=alpha, U0=np.zeros(2), cov=[0.0625, 0.1], N=N, lam_ais=lam_ais, cma_sigma=0.75, **kw)
    ref = pol(env, Z[0], di[0], du[0])
    try:
        eng = Engine("car", 1, kind, K, T, batch=1, lam=lam, alpha=alpha, ais_its=N, lam_ais=lam_ais, cma_sigma=0.75, cov=[0.0625, 0.1], track=track, **kw)
    except MPOPISError as e:
        # the one configuration the engine refuses up front: the reference would throw at its first update (BoundsError), the oracle reports it there
        assert e.code == -1 and "BoundsError" in str(e) and ref["status"] == -1
        return
    try:
        got = eng.policy_step(Z, di, du)
        code = 0
    except MPOPISError as e:
        got, code = None, e.code
    eng.close()
    assert code == ref["status"], (code, ref["status"])
    if code == 0:
        assert int(got["iters_run"][0]) == ref["iters_run"]
        assert np.all(np.isfinite(got["control"][0])) and np.max(np.abs(got["control"][0] - ref["control"])) < 1e-9
        assert np.max(np.abs(got["cost"][0] - ref["cost"]) / (np.abs(ref["cost"]) + 1e-9)) < 1e-7


@pytest.mark.parametrize("kind,ncars", [("gmppi", 1), ("musigmaaismppi", 1), ("cemppi", 3)])
def test_custom_action_bounds_reach_rollouts_and_control(oracle, track, kind, ncars):
    """mpopis_set_action_bounds = the bounds get_model_controls and the final clamp take from action_space(env) (src/utils.jl:103-116, :91): narrower, asymmetric
    bounds must clamp every rollout's controls and the returned control exactly as the oracle's do (per car for the multi-car env)."""
    from mpopis_amd import build
    build.build()
    from mpopis_amd.engine import Engine
    K, T, N = 256, 12, 3
    cs = 2 * ncars * T
    cov = np.tile([0.25, 0.4], ncars)                          # wide noise: most samples hit a bound somewhere
    lo = np.tile([-0.5, -0.2], ncars) - 0.05 * np.arange(2 * ncars)
    hi = np.tile([0.3, 0.9], ncars) - 0.03 * np.arange(2 * ncars)
    eng = Engine("car", ncars, kind, K, T, batch=2, lam=10.0, ais_its=N, lam_ais=20.0, elite_threshold=0.8, cov=cov, track=track, seed=3)
    eng.set_action_bounds(lo, hi)
    rng = np.random.default_rng(21)
    Z = rng.standard_normal((2, N if kind != "gmppi" else 1, K, cs))
    got = eng.policy_step(Z)
    for b in range(2):
        env = oracle.OracleEnv("car", ncars, track=track)
        pol = oracle.OraclePolicy(kind, env, K, T, lam=10.0, U0=np.zeros(2 * ncars), cov=cov, N=N, lam_ais=20.0, elite_threshold=0.8)
        for i in range(2 * ncars):
            pol.p.lo[i] = lo[i]; pol.p.hi[i] = hi[i]
        ref = pol(env, Z[b])
        assert ref["status"] == 0 and int(got["iters_run"][b]) == ref["iters_run"]
        assert np.max(np.abs(got["cost"][b] - ref["cost"]) / (np.abs(ref["cost"]) + 1e-9)) < 1e-7
        assert np.max(np.abs(got["control"][b] - ref["control"])) < 1e-9
        assert np.all(got["control"][b] >= lo - 1e-15) and np.all(got["control"][b] <= hi + 1e-15)
    # and the bounds bite: with the default [-1, 1] the same noise gives other costs
    eng.set_action_bounds(-np.ones(2 * ncars), np.ones(2 * ncars))
    eng.reset(); eng.set_U(np.zeros((2, cs)))
    assert not np.allclose(eng.policy_step(Z)["cost"], got["cost"])
    eng.close()


@pytest.mark.parametrize("env_kind", ["mountaincar", "cartpole"])
def test_nan_control_in_the_scalar_action_envs(oracle, env_kind):
    """the same for the RL.jl adapters (k_rollout_simple): a NaN force reaches every rollout, the reference's act! rejects it; engine and oracle report -3"""
    from mpopis_amd import build
    build.build()
    from mpopis_amd.engine import Engine
    from mpopis_amd._lib import MPOPISError
    K, T = 20, 15
    eng = Engine(env_kind, 1, "gmppi", K, T, batch=2, lam=0.1, cov=[1.5], seed=2)
    assert np.all(np.isfinite(eng.policy_step(None)["cost"]))
    U = np.zeros((2, T)); U[0, 3] = np.nan
    eng.set_U(U)
    with pytest.raises(MPOPISError) as ei:
        eng.policy_step(None)
    assert ei.value.code == -3
    eng.close()
    env = oracle.OracleEnv(env_kind)
    pol = oracle.OraclePolicy("gmppi", env, K, T, lam=0.1, U0=np.zeros(1), cov=[1.5])
    pol.U = U[0]
    assert pol(env, np.random.default_rng(0).standard_normal((1, K, T)))["status"] == -3


@pytest.mark.parametrize("kind,ncars,cov,lam,N", [("gmppi", 1, [1.0, 1.0], 1e4, 1), ("gmppi", 3, [0.6, 0.8], 1e4, 1), ("musigmaaismppi", 2, [0.5, 0.7], 1e3, 3),
                                               ("cemppi", 1, [0.8, 0.9], 1e4, 3)])
def test_run_trials_bookkeeping_when_things_go_wrong(oracle, track, kind, ncars, cov, lam, N):
    """The trial loop's statistics (src/examples/car_example.jl:238-302) in closed loops that do NOT go well: wide proposals and a large lambda make the
    controller drive erratically, cars leave the track, exceed the slip-angle limit and (multi-car) run into each other -- the counters of track / beta / crash
    violations, the per-car means and maxima and the early termination (`T_viol > 10 || beta_viol > 50`, :277-279) must match the oracle's restatement field
    by field, and so must the actions step by step (same Philox streams).  The calm closed loops of tests/test_gpu_parity.py never enter these branches."""
    from mpopis_amd import build
    build.build()
    from mpopis_amd.engine import Engine
    K, T, B, steps, seed = 96, 12, 3, 70, 4100
    covf = np.tile(cov, ncars)
    eng = Engine("car", ncars, kind, K, T, batch=B, lam=lam, ais_its=N, lam_ais=lam, elite_threshold=0.8, cov=covf, track=track, seed=seed)
    # start states that are already in trouble: near the lane's edge heading outwards, sliding (slip angle beyond the limit), the cars of a multi-car env
    # within collision distance of each other
    tx, ty, tw = track
    rng = np.random.default_rng(seed)
    x0 = np.zeros((B, 8 * ncars))
    for b in range(B):
        j = int(rng.integers(0, len(tx)))
        jn = (j + 1) % len(tx)
        head = np.arctan2(ty[jn] - ty[j], tx[jn] - tx[j])
        nrm = np.array([-np.sin(head), np.cos(head)])
        for c in range(ncars):
            off = (0.6 + 0.25 * b) * 0.5 * tw[j] * (1 if (b + c) % 2 else -1)          # 0.3 ... 0.55 lane widths off the centre line
            pos = np.array([tx[j], ty[j]]) + off * nrm + 2.5 * c * np.array([np.cos(head), np.sin(head)])
            x0[b, 8 * c:8 * c + 8] = [pos[0], pos[1], head + 0.5 * np.sign(off), 9.0 + 3 * b, 8.0 * np.sign(off) * (1 if b != 1 else 0.3), 0.4, 0.1, 0.0]
    eng.set_state(x0)
    rec, acts = eng.run_trials(num_steps=steps, laps=2, log_actions=True)
    eng.close()
    viol = np.zeros(3)
    for b in range(B):
        env = oracle.OracleEnv("car", ncars, track=track)
        env.state = x0[b]
        pol = oracle.OraclePolicy(kind, env, K, T, lam=lam, U0=np.zeros(2 * ncars), cov=covf, N=N, lam_ais=lam, elite_threshold=0.8, nthreads=8)
        r = pol.run_trial(env, seed + b + 1, num_steps=steps, laps=2, log_actions=True)
        assert r["status"] == 0 and rec[b, 15] == 0
        n = int(r["steps"])
        assert rec[b, 1] == r["steps"], (rec[b, 1], r["steps"])                      # incl. the early termination on violations
        assert rec[b, 14] == r["rollouts"]
        assert np.max(np.abs(acts[b][:n] - r["actions"][:n])) < 1e-6
        ref = np.array([r["rew"], r["steps"], r["rew_per_step"]] + r["lap_t"] + [r["mean_v"], r["max_v"], r["mean_beta"], r["max_beta"], r["beta_viol"], r["trk_viol"], r["crash_viol"]])
        assert np.array_equal(rec[b, 11:14], ref[11:14]), (rec[b, 11:14], ref[11:14])  # violation counters: exact
        assert np.max(np.abs(rec[b, :11] - ref[:11]) / np.maximum(1.0, np.abs(ref[:11]))) < 1e-6, (rec[b, :11], ref[:11])
        viol += ref[11:14]
    print("\n[run_trials, rough] %s cars=%d: beta / track / crash violations over %d trials: %s" % (kind, ncars, B, viol))
    assert viol.sum() > 0                                                            # the case really enters the violation branches


@pytest.mark.parametrize("ncars", [1, 2])
def test_run_trials_lap_counting_and_lap_termination(oracle, track, ncars):
    """Lap bookkeeping (car_example.jl:273-279): a lap is counted when the rearmost car's y crosses 0 upwards within 15 m of the origin, its step number goes to
    lap_t, and the trial ends after `laps` laps.  Start states a few metres before the line (the calm 30-step loops elsewhere never get there): engine and
    oracle must count the same laps at the same steps and stop at the same step."""
    from mpopis_amd import build
    build.build()
    from mpopis_amd.engine import Engine
    K, T, B, seed = 128, 12, 3, 808
    cov = np.tile([0.0625, 0.1], ncars)
    x0 = np.zeros((B, 8 * ncars))
    for b in range(B):
        for c in range(ncars):
            x0[b, 8 * c:8 * c + 8] = [3.0 * c - 1.0, -2.0 - 3.0 * b - 1.5 * c, np.pi / 2, 10.0 + b, 0.0, 0.0, 0.0, 0.0]     # heading +y, 2 ... 11 m before the line
    for laps in (1, 2):
        eng = Engine("car", ncars, "gmppi", K, T, batch=B, lam=10.0, cov=cov, track=track, seed=seed)
        eng.set_state(x0)
        rec, acts = eng.run_trials(num_steps=14, laps=laps, log_actions=True)
        eng.close()
        for b in range(B):
            env = oracle.OracleEnv("car", ncars, track=track)
            env.state = x0[b]
            pol = oracle.OraclePolicy("gmppi", env, K, T, lam=10.0, U0=np.zeros(2 * ncars), cov=cov, nthreads=8)
            r = pol.run_trial(env, seed + b + 1, num_steps=14, laps=laps, log_actions=True)
            assert rec[b, 1] == r["steps"], (laps, b, rec[b, 1], r["steps"])
            assert list(rec[b, 3:7]) == list(r["lap_t"]), (laps, b, rec[b, 3:7], r["lap_t"])
            assert r["lap_t"][0] > 0                                              # the line really was crossed
            if laps == 1:
                assert r["steps"] < 14                                            # ... and one lap ended the trial early
            n = int(r["steps"])
            assert np.max(np.abs(acts[b][:n] - r["actions"][:n])) < 1e-6
