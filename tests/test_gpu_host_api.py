"""The Python host mirror of the reference's operator API, exercised the way the reference's own
callers use it (src/examples/car_example.jl:170-281): env + get_policy + pol(env) + env(act) + reward."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    from mpopis_amd import build
    build.build()
    import mpopis_amd
    return mpopis_amd


def test_closed_loop_like_reference_harness(M, oracle, track):
    rng = np.random.default_rng(21)
    K, T, N = 150, 20, 4
    env = M.CarRacingEnv()
    pol = M.get_policy(":cemppi", env, K, T, 10.0, 1.0, np.zeros(2), np.array([0.0625, 0.1]), False, N, 20.0, 0.8, "mle", 0.75, 0.8)
    oenv = oracle.OracleEnv("car", 1, track=track)
    opol = oracle.OraclePolicy("cemppi", oenv, K, T, lam=10.0, U0=[0.0, 0.0], cov=[0.0625, 0.1], N=N, elite_threshold=0.8)
    assert np.allclose(env.state, oenv.state)
    rew = 0.0
    for cnt in range(6):
        Z = rng.standard_normal((N, K, 2 * T))
        act = pol(env, Z=Z)                       # act = pol(env)
        ref = opol(oenv, Z)
        assert np.max(np.abs(act - ref["control"])) < 1e-8
        env(act)                                  # env(act)
        oenv.step(ref["control"])
        assert np.max(np.abs(env.state - oenv.state)) < 1e-8
        r = M.reward(env)
        assert abs(r - oenv.reward()) < 1e-8 * abs(oenv.reward())
        rew += r
        w, d = M.within_track(env)
        ow, od = oracle.within_track(track, oenv.state[:2])
        assert w == ow and abs(d - od) < 1e-9
        assert not M.exceed_β(env)
    assert np.max(np.abs(pol.U - opol.U)) < 1e-8
    pol.close()


def test_policy_constructors_and_errors(M):
    env = M.MultiCarRacingEnv(3)
    assert list(env.state[[0, 8, 16]]) == [0.0, 5.0, -5.0]
    lo, hi = M.action_space(env)
    assert len(lo) == 6 and np.all(lo == -1) and np.all(hi == 1)
    pol = M.CMAMPPI_Policy(env, num_samples=256, horizon=10, λ=10.0, U0=np.zeros(6), cov_mat=M.block_diagm([0.0625, 0.1], 3), opt_its=3, σ=0.75)
    assert pol.Σ.shape == (60, 60) and pol.params.cs == 60
    a = pol(env)
    assert a.shape == (6,) and np.all(np.abs(a) <= 1.0)
    env(a)
    assert env.t == 1 and np.isfinite(M.reward(env))
    pol.close()
    with pytest.raises(M.MPOPISError):
        M.GMPPI_Policy(env, num_samples=8, horizon=4, U0=np.zeros(5))          # "U₀ must be length of action space or control space"
    with pytest.raises(M.MPOPISError):
        M.get_policy(":nope", env, 8, 4, 1.0, 1.0, np.zeros(6), np.ones(6), False, 2, 1.0, 0.8, "mle", 1.0, 0.8)
    with pytest.raises(M.MPOPISError):
        M.CEMPPI_Policy(env, Σ_est="bogus")
    e1 = M.CarRacingEnv()
    with pytest.raises(M.MPOPISError):
        e1([2.0, 0.0])                                                          # Action is not in action space


def test_logger_and_calculate_trajectory_costs(M, oracle, track):
    env = M.CarRacingEnv()
    pol = M.GMPPI_Policy(env, num_samples=64, horizon=8, λ=10.0, U0=np.zeros(2), cov_mat=[0.0625, 0.1], log=True)
    rng = np.random.default_rng(5)
    Z = rng.standard_normal((1, 64, 16))
    cost, E, w = M.calculate_trajectory_costs(pol, env, Z=Z)
    assert E.shape == (16, 64) and abs(w.sum() - 1) < 1e-12
    assert len(pol.logger.trajectories) == 64 and pol.logger.trajectories[0].shape == (8, 8)
    oenv = oracle.OracleEnv("car", 1, track=track)
    opol = oracle.OraclePolicy("gmppi", oenv, 64, 8, lam=10.0, U0=[0.0, 0.0], cov=[0.0625, 0.1])
    c2, tr = opol.simulate_model(np.zeros(16), E, log=True)
    assert np.max(np.abs(cost - c2) / np.abs(c2)) < 1e-8
    assert np.max(np.abs(np.array(pol.logger.trajectories) - tr)) < 1e-8
    pol.U = np.zeros(16)
    c3 = M.simulate_model(pol, env, E)
    assert np.max(np.abs(c3 - c2) / np.abs(c2)) < 1e-8
    pol.close()


def test_simulate_car_racing_and_mountaincar(M):
    rec, summ = M.simulate_car_racing(num_trials=4, num_steps=20, policy_type=":μΣaismppi", num_samples=256, horizon=20, ais_its=3,
                                      seed=123, quiet=True)
    assert rec.shape[0] == 4 and np.all(rec[:, 2] == 20) and np.all(rec[:, 13] == 0)     # steps, no track violations
    assert np.all(rec[:, 8] > 9.0)                                                       # mean speed
    assert summ["AVE"].shape == summ["MAX"].shape
    rec2, _ = M.simulate_car_racing(num_trials=4, num_steps=20, policy_type=":μΣaismppi", num_samples=256, horizon=20, ais_its=3,
                                    seed=123, quiet=True)
    assert np.array_equal(rec[:, :15], rec2[:, :15])                                    # deterministic for a fixed seed
    recm, _ = M.simulate_mountaincar(num_trials=3, num_steps=200, policy_type=":mppi", x0=[-0.5, -0.45, -0.55], seed=9, quiet=True)
    assert recm.shape == (3, 16) and np.all(recm[:, 1] >= 1)


def test_cartpole_env_and_simulate(M, oracle):
    """CartPoleEnv mirror (cartpole_example.jl:3-6) + simulate_cartpole (same kwargs/defaults as the reference)."""
    x0 = [0.02, 0.0, -0.03, 0.05]
    env = M.CartPoleEnv(x0=x0)
    oenv = oracle.OracleEnv("cartpole"); oenv.state = x0
    pol = M.get_policy(":cemppi", env, 20, 15, 0.1, 1.0, [0.0], [1.5], False, 5, 0.1, 0.8, ":mle", 0.75, 0.8)
    for _ in range(5):
        act = pol(env)
        assert act.shape == (1,) and -1.0 <= act[0] <= 1.0
        env(act); oenv.step(act)
        assert np.max(np.abs(M.state(env) - oenv.state)) < 1e-13
        assert M.reward(env) == oenv.reward() and bool(M.is_terminated(env)) == bool(oenv.e.done)
    pol.close()
    rec, summ = M.simulate_cartpole(num_trials=3, num_steps=60, seed=11, quiet=True)
    assert rec.shape == (3, 16) and np.all(rec[:, 1] >= 1) and np.all(rec[:, 15] == 0)
    rec2, _ = M.simulate_cartpole(num_trials=3, num_steps=60, seed=11, quiet=True)
    assert np.array_equal(rec[:, :15], rec2[:, :15])
    with pytest.raises(TypeError):
        M.simulate_cartpole(num_trails=1)


def test_sharded_trials_in_one_handle_match_full_run(M):
    """SURVEY 8e partitioning: a rank holding trials {1,3,5,7} (rank 0 of 2) keeps them in ONE handle via
    mpopis_seed_slots and must reproduce exactly the same trials of the unsharded num_trials=8 run
    (trial k draws from seed + k, car_example.jl:187-188)."""
    from mpopis_amd.engine import Engine
    from mpopis_amd.examples import shard_trials
    kw = dict(lam=10.0, ais_its=3, lam_ais=20.0, cov=[0.0625, 0.1])
    seed = 777
    full = Engine("car", 1, "musigmaaismppi", 256, 20, batch=8, seed=seed, **kw)
    rec_full, act_full = full.run_trials(15, 2, log_actions=True)
    full.close()
    for rank in (0, 1):
        mine = shard_trials(8, rank, 2)
        assert mine == ([1, 3, 5, 7] if rank == 0 else [2, 4, 6, 8])
        part = Engine("car", 1, "musigmaaismppi", 256, 20, batch=len(mine), seed=seed, **kw)
        part.seed_slots([seed + k for k in mine])
        rec, act = part.run_trials(15, 2, log_actions=True)
        part.close()
        idx = [k - 1 for k in mine]
        assert np.array_equal(act, act_full[idx])                      # bit-identical actions ...
        assert np.array_equal(rec[:, :15], rec_full[idx][:, :15])      # ... and records
    # different seeds really give different trials
    assert not np.array_equal(act_full[0], act_full[1])


def test_gather_summary_through_the_abi(M):
    """mpopis_gather_summary: world == 1 copy path, and a real one-rank RCCL communicator (dlopen of librccl,
    ncclGetUniqueId, ncclCommInitRank, ncclGather/ncclAllGather on the handle's stream)."""
    from mpopis_amd.engine import Engine
    from mpopis_amd._lib import RECORD_LEN
    eng = Engine("car", 1, "gmppi", 64, 5, batch=3, lam=10.0, cov=[0.0625, 0.1])
    rec = np.arange(3 * RECORD_LEN, dtype=np.float64).reshape(3, RECORD_LEN)
    eng.comm_init(0, 1)
    assert eng.comm_count() == 0                                        # no RCCL communicator behind the copy path
    parts = eng.gather_summary(rec, 5)                                  # n_max > n_local: padded rows are dropped again
    assert len(parts) == 1 and np.array_equal(parts[0], rec)
    eng.close()
    eng = Engine("car", 1, "gmppi", 64, 5, batch=3, lam=10.0, cov=[0.0625, 0.1])
    uid = Engine.comm_unique_id()
    assert len(uid) == 128
    eng.comm_init(0, 1, uid)
    assert eng.comm_count() == 1                                        # ncclCommCount of the real communicator
    parts = eng.gather_summary(rec, 3)
    assert len(parts) == 1 and np.array_equal(parts[0], rec)
    parts = eng.gather_summary(rec[:0], 3)                              # a rank without trials
    assert parts[0].shape == (0, RECORD_LEN)
    eng.close()


def test_debug_launch_checks_and_two_handles_one_device(M, monkeypatch):
    """MPOPIS_DEBUG_LAUNCH=1 checks hipGetLastError + syncs after every kernel class (same results as a release run); two
    handles on device 0 used in interleaved order get their large-LDS kernels configured per device, not per process."""
    from mpopis_amd.engine import Engine
    kw = dict(lam=10.0, ais_its=3, lam_ais=20.0, cov=[0.0625, 0.1], seed=5)
    a = Engine("car", 1, "musigmaaismppi", 512, 50, batch=2, **kw)          # cs = 100: 100 KiB LDS Cholesky, 96 KiB scatter
    monkeypatch.setenv("MPOPIS_DEBUG_LAUNCH", "1")
    b = Engine("car", 1, "musigmaaismppi", 512, 50, batch=2, **kw)
    monkeypatch.delenv("MPOPIS_DEBUG_LAUNCH")
    c = Engine("car", 1, "pmcmppi", 512, 50, batch=2, **kw)
    rb = b.policy_step(None)
    rc_ = c.policy_step(None)
    ra = a.policy_step(None)
    assert np.array_equal(ra["control"], rb["control"]) and np.array_equal(ra["cost"], rb["cost"])
    assert np.all(np.isfinite(rc_["control"]))
    for e in (a, b, c):
        e.close()


def test_plain_c_client_gives_the_same_numbers_as_the_python_mirror(M, tmp_path):
    """tests/abi_client.c (C99, links only libmpopis_hip.so) runs a MountainCar :cemppi closed loop on the device RNG; the same calls
    through the Python mirror must print the same 17-digit text: the C ABI is the product, Python is one of its hosts."""
    import os, shutil, subprocess
    from mpopis_amd import engine as eng_mod
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "mpopis_amd", "lib")
    exe = str(tmp_path / "abi_client")
    r = subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "abi_client.c"), "-L" + libdir, "-lmpopis_hip",
                        "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("step ")]
    assert len(lines) == 5
    eng = eng_mod.Engine("mountaincar", 0, "cemppi", 64, 15, batch=2, lam=0.1, alpha=1.0, ais_its=4, lam_ais=0.0, elite_threshold=0.8, sigma_est="mle",
                         cma_sigma=1.0, seed=1234, cov=[1.0])
    eng.reset()
    for step in range(5):
        got = eng.policy_step(None)
        rew = eng.env_step(got["control"])
        c = got["control"].reshape(-1)
        want = "step %d control %.17g %.17g reward %.17g %.17g cost0 %.17g iters %d %d" % (step, c[0], c[1], rew[0], rew[1], got["cost"][0][0],
                                                                                           got["iters_run"][0], got["iters_run"][1])
        assert lines[step] == want, (lines[step], want)
    eng.close()
    # part 2 of the client: the reference harness' order (construct, seed!(pol, seed + k), first pol(env)) through a lazily created handle
    # and mpopis_policy_call; the program itself checks that two runs repeat bit for bit.  Here: the printed controls equal the four-call
    # composition (set_state + set_U + policy_step + get_U) on an eagerly created, then seeded handle.
    assert "harness reproducible 1" in out.stdout
    hl = [l for l in out.stdout.splitlines() if l.startswith("harness trial ")]
    assert len(hl) == 6
    for k in (1, 2):
        envh = eng_mod.Engine("mountaincar", 0, "gmppi", 1, 1, batch=1, lam=1.0, alpha=1.0)
        pol = eng_mod.Engine("mountaincar", 0, "cemppi", 64, 15, batch=1, lam=0.1, alpha=1.0, ais_its=4, lam_ais=0.0, elite_threshold=0.8, sigma_est="mle",
                             cma_sigma=1.0, seed=999, cov=[1.0])
        pol.seed(777 + k - 1)                                   # seed!(pol, seed + k) on a one-slot handle (slot 0 draws from arg + 1)
        U = np.zeros((1, 15))
        for s_ in range(3):
            x, t, done = envh.get_state()
            pol.set_state(x, t, done); pol.set_U(U)
            got = pol.policy_step(None, minimal=True)
            U = pol.get_U()
            envh.env_step(got["control"])
            want = "harness trial %d step %d x %.17g %.17g control %.17g U0 %.17g" % (k, s_, x[0, 0], x[0, 1], got["control"][0, 0], U[0, 0])
            assert hl[(k - 1) * 3 + s_] == want, (hl[(k - 1) * 3 + s_], want)
        envh.close(); pol.close()
    # the round-3 binding logic (a seed issued before the handle exists is dropped, the handle is created with a clock seed) must FAIL this program
    exe_old = str(tmp_path / "abi_client_old")
    r = subprocess.run(["gcc", "-std=c99", "-DBINDING_DROPS_EARLY_SEED", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "abi_client.c"),
                        "-L" + libdir, "-lmpopis_hip", "-Wl,-rpath," + libdir, "-o", exe_old], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    old = subprocess.run([exe_old], capture_output=True, text=True, timeout=120)
    assert old.returncode == 2 and "not reproducible" in old.stderr, (old.returncode, old.stderr)


def test_policy_call_equals_the_four_call_composition(M):
    """mpopis_policy_call (one host wait, mailbox in pinned mapped memory) against set_state + set_U + policy_step + get_U on a twin handle:
    controls, rolled U, costs, weights and iters bit-identical over a short closed loop; car env, 3 slots, :μΣaismppi and :cemppi;
    a third handle takes the same loop with want_cost = False (the mailbox-spin wait instead of the stream wait): same controls and U;
    x = None keeps the resident state; U = None rolls the resident pol.U."""
    from mpopis_amd import engine as eng_mod
    for pol, kw in (("μΣaismppi", {}), ("cemppi", dict(sigma_est="ss", elite_threshold=0.8)), ("gmppi", {})):
        mk = lambda: eng_mod.Engine("car", 1, pol, 256, 20, batch=3, lam=10.0, alpha=1.0, ais_its=3, lam_ais=20.0, cov=[0.0625, 0.1], seed=41, **kw)
        a, b, c = mk(), mk(), mk()
        Ua, Ub, Uc = np.zeros((3, 40)), np.zeros((3, 40)), np.zeros((3, 40))
        for step in range(4):
            x, t, done = b.get_state()
            x[:, 3] += 0.25 * step                               # the host owns the state: hand over something the resident copy does not hold
            ra = a.policy_call(x, t, done, Ua, want_cost=True)   # Ua rolled in place
            rc_ = c.policy_call(x, t, done, Uc, want_cost=False) # the production form: nothing but the mailbox comes back, the host spins on its sequence word
            assert rc_["cost"] is None and np.array_equal(rc_["control"], ra["control"]) and np.array_equal(rc_["iters_run"], ra["iters_run"]), (pol, step)
            assert np.array_equal(Uc, Ua), (pol, step)
            b.set_state(x, t, done); b.set_U(Ub)
            rb = b.policy_step(None)
            Ub = b.get_U()
            for k in ("control", "cost", "weights", "iters_run"):
                assert np.array_equal(ra[k], rb[k]), (pol, step, k)
            assert np.array_equal(Ua, Ub), (pol, step)
            assert np.array_equal(a.get_U(), Ub) and np.array_equal(a.get_state()[0], x)
            b.env_step(rb["control"])
        # resident forms: no state / no U handed over
        xa = a.get_state()[0]
        b.set_state(xa, *a.get_state()[1:])
        r1 = a.policy_call()
        r2 = b.policy_step(None, minimal=True)
        assert np.array_equal(r1["control"], r2["control"]) and np.array_equal(a.get_U(), b.get_U())
        with pytest.raises(Exception):
            a.policy_call(None, [0, 0, 0], None)                 # t without x
        a.close(); b.close(); c.close()


def test_concurrent_handles_with_cooperative_kernels(M):
    """Four handles on one device, driven from four host threads at once, all on 3-car configurations whose Cholesky / Lanczos kernels run as
    multi-workgroup clusters with bounded spin-waits (cs = 300): every trial must finish with status 0 -- in particular never -4, the
    code a cluster reports when a partner workgroup did not show up within its wait bound."""
    import threading
    res = {}

    def worker(i, pt):
        try:
            rec, _ = M.simulate_car_racing(num_trials=4, num_steps=40, num_cars=3, policy_type=pt, num_samples=512, horizon=50, ais_its=4, seed=300 + i, quiet=True)
            res[i] = (float(rec[:, 16].min()), int(rec[:, 2].min()))
        except Exception as e:                                   # :cmamppi may legitimately end in the reference's PosDefException (-2)
            res[i] = repr(e)
    ths = [threading.Thread(target=worker, args=(i, pt)) for i, pt in enumerate([":μΣaismppi", ":cemppi", ":cmamppi", ":pmcmppi"])]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=300)
    assert len(res) == 4
    for i, v in res.items():
        assert "-4" not in str(v), (i, v)
        if isinstance(v, tuple):
            assert v[0] in (0.0, -2.0) and v[1] >= 1, (i, v)
