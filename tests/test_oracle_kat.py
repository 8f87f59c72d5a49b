"""Known-answer tests that pin the CPU oracle (SURVEY 8c list).  The reference ships no vectors,
so these are hand-derivable values plus an independent NumPy re-derivation (oracle/np_rederive.py)."""
import math
import numpy as np
import pytest
from oracle import np_rederive as NP

RTOL = 1e-12


def test_philox_kat(oracle):
    # Random123 kat_vectors, philox4x32-10
    assert oracle.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    assert oracle.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]


def test_philox_normals_moments_and_streams(oracle):
    z = oracle.philox_normals(7, 3, 1, 400000)
    assert abs(z.mean()) < 6e-3 and abs(z.std() - 1) < 5e-3 and abs((z ** 4).mean() - 3) < 0.06
    assert np.array_equal(z[:1001], oracle.philox_normals(7, 3, 1, 1001))       # prefix-stable, odd n
    assert not np.allclose(z[:100], oracle.philox_normals(7, 3, 2, 100))


def test_philox_normals_four_per_call(oracle):
    """The stream's definition (mpopis_amd/csrc/philox.h, orc_philox_normals): call q yields normals 4q .. 4q+3 = Box-Muller on (w0, w1) and on
    (w2, w3) with 32-bit uniforms (w + 0.5) 2^-32 -- restated here from the Philox words -- and, as a distribution: Kolmogorov-Smirnov against the
    normal CDF, independence of the four outputs of a call (the two Box-Muller outputs of a pair share their radius: uncorrelated, and so are
    their squares only through the angle), tails at 3 and 4 sigma at the binomial level, nothing beyond sqrt(-2 log 2^-33) = 6.76."""
    from scipy import stats
    n = 1 << 20
    z = oracle.philox_normals(12345, 2, 5, n)
    for q in (0, 1, 77, n // 4 - 1):
        w = oracle.philox4x32_10([q & 0xffffffff, q >> 32, 2, 5], [12345, 0])
        ref = []
        for wr, wa in ((w[0], w[1]), (w[2], w[3])):
            u1, u2 = (wr + 0.5) / 2.0 ** 32, (wa + 0.5) / 2.0 ** 32
            R = math.sqrt(-2.0 * math.log(u1))
            ref += [R * math.cos(2 * math.pi * u2), R * math.sin(2 * math.pi * u2)]
        np.testing.assert_allclose(z[4 * q:4 * q + 4], ref, rtol=0, atol=1e-13)
    assert stats.kstest(z, "norm").pvalue > 1e-3
    Q = z.reshape(-1, 4)
    C = np.corrcoef(Q.T)
    assert np.max(np.abs(C - np.eye(4))) < 5.0 / math.sqrt(Q.shape[0])                      # 5 sigma of a sample correlation
    C2 = np.corrcoef((Q ** 2).T)
    assert np.max(np.abs(C2 - np.eye(4))) < 5.0 / math.sqrt(Q.shape[0])
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 5.0 / math.sqrt(n)
    for thr in (3.0, 4.0):
        p = 2 * stats.norm.sf(thr)
        k = int(np.sum(np.abs(z) > thr))
        assert abs(k - n * p) < 5 * math.sqrt(n * p) + 1
    assert np.max(np.abs(z)) < 6.76


def test_compute_weights_kat(oracle):
    w = oracle.compute_weights(1.0, [1.0, 2.0, 3.0])
    e = np.array([1.0, math.exp(-1), math.exp(-2)])
    np.testing.assert_allclose(w, e / e.sum(), rtol=RTOL)
    w2 = oracle.compute_weights(10.0, [5.0, 5.0])
    np.testing.assert_allclose(w2, [0.5, 0.5], rtol=0)
    np.testing.assert_allclose(oracle.compute_weights(0.1, [3.0, 1.0, 7.0]), NP.compute_weights(0.1, np.array([3.0, 1.0, 7.0])), rtol=RTOL)


def test_block_diagm(oracle):
    A = np.array([[1.0, 2.0], [3.0, 4.0]])
    B = oracle.block_diagm(A, 3)
    ref = np.zeros((6, 6))
    for b in range(3):
        ref[2 * b:2 * b + 2, 2 * b:2 * b + 2] = A
    assert np.array_equal(B, ref)
    assert np.array_equal(oracle.block_diagm(np.array([0.0625, 0.1]), 2), np.diag([0.0625, 0.1, 0.0625, 0.1]))


def test_m_elite_rounding(oracle):
    assert oracle.m_elite(150, 0.8) == 30
    assert oracle.m_elite(4096, 0.8) == 819
    assert oracle.m_elite(20, 0.8) == 4
    assert oracle.m_elite(10, 0.75) == 2      # 2.5 -> 2 (half to even)


def test_tire_fz(oracle):
    p = oracle.car_default_params()
    import ctypes as C
    dp = p.ctypes.data_as(C.POINTER(C.c_double))
    L = 1.53 + 1.23
    fzf = oracle.lib().orc_calc_tire_fz(dp, 0.0, b"f")
    fzr = oracle.lib().orc_calc_tire_fz(dp, 0.0, b"r")
    assert fzf == pytest.approx(2000 * 1.23 * 9.81 / L, rel=RTOL)
    assert fzr == pytest.approx(2000 * 1.53 * 9.81 / L, rel=RTOL)
    for fx in (-22500.0, 7200.0):
        s = oracle.lib().orc_calc_tire_fz(dp, fx, b"f") + oracle.lib().orc_calc_tire_fz(dp, fx, b"r")
        assert s == pytest.approx(2000 * 9.81, rel=1e-13)                           # load transfer conserves m g
        assert oracle.lib().orc_calc_tire_fz(dp, fx, b"f") == pytest.approx((2000 * 1.23 * 9.81 - 0.3 * fx) / L, rel=RTOL)


def test_tire_fy_branches_and_continuity(oracle):
    fy = oracle.lib().orc_calc_tire_fy
    mu, Ca, fz, fx = 0.9, 150000.0, 8000.0, 1000.0
    fymax = math.sqrt((mu * fz) ** 2 - fx ** 2)
    a_sw = math.atan(3 * fymax / Ca)
    assert fy(0.0, mu, Ca, fz, fx) == 0.0
    assert fy(0.5, mu, Ca, fz, fx) == pytest.approx(-fymax, rel=RTOL)             # saturated
    assert fy(-0.5, mu, Ca, fz, fx) == pytest.approx(fymax, rel=RTOL)
    a = 0.01
    ta = math.tan(a)
    assert fy(a, mu, Ca, fz, fx) == pytest.approx(-Ca * ta + Ca ** 2 / (3 * fymax) * abs(ta) * ta - Ca ** 3 / (27 * fymax ** 2) * ta ** 3, rel=RTOL)
    assert fy(a_sw * (1 - 1e-9), mu, Ca, fz, fx) == pytest.approx(-fymax, rel=1e-7)  # continuity at the switch
    assert fy(0.2, mu, Ca, 100.0, 5000.0) == pytest.approx(-math.sqrt(1e-8), rel=RTOL)  # fy_max floor 1e-8


def test_car_step_coast_analytic(oracle):
    """a=(0,0) from the reset state: straight-line coast, Vx' = -(C_D0 + C_D1 Vx)/m, Euler 10 x 0.01 s."""
    p = oracle.car_default_params()
    s0 = np.array([0.0, 0.0, math.pi / 2, 10.0, 0.0, 0.0, 0.0, 0.0])
    s1 = oracle.car_step(p, s0, [0.0, 0.0])
    Vx, y = 10.0, 0.0
    for _ in range(10):
        Vx += -(241.0 + 25.1 * Vx) / 2000.0 * 0.01
        y += Vx * 0.01                                                               # sin(pi/2)=1
    assert s1[3] == pytest.approx(Vx, rel=1e-14)
    assert s1[1] == pytest.approx(y, rel=1e-14)
    assert abs(s1[0]) < 1e-15 and s1[4] == 0.0 and s1[5] == 0.0 and s1[6] == 0.0 and s1[7] == 0.0
    assert s1[2] == pytest.approx(math.pi / 2, rel=1e-15)


def test_car_step_steering_rate_limit(oracle):
    p = oracle.car_default_params()
    s0 = np.array([0.0, 0.0, 0.0, 10.0, 0.0, 0.0, 0.0, 0.0])
    s1 = oracle.car_step(p, s0, [1.0, 0.5])
    # commanded 18deg/0.1s = 180deg/s > 90deg/s limit -> delta = 90deg/s*0.1s = 9deg
    assert s1[6] == pytest.approx(9.0 * math.pi / 180.0, rel=1e-13)
    assert s1[7] == 0.5
    s2 = oracle.car_step(p, s0, [0.25, 0.0])     # 4.5deg/0.1s = 45deg/s < limit -> reaches target exactly
    assert s2[6] == pytest.approx(4.5 * math.pi / 180.0, rel=1e-13)


def test_car_step_vs_numpy_rederivation(oracle):
    rng = np.random.default_rng(0)
    p = oracle.car_default_params()
    pn = NP.car_params()
    K = 256
    S = np.zeros((K, 8))
    S[:, 0:2] = rng.uniform(-50, 50, (K, 2))
    S[:, 2] = rng.uniform(-3.1, 3.1, K)
    S[:, 3] = rng.uniform(-2, 35, K)
    S[:, 4] = rng.uniform(-6, 6, K)
    S[:, 5] = rng.uniform(-1.5, 1.5, K)
    S[:, 6] = rng.uniform(-0.3, 0.3, K)
    A = rng.uniform(-1, 1, (K, 2))
    A[:8, 1] = 0.0
    ref = NP.car_step(pn, S, A)
    got = np.stack([oracle.car_step(p, S[k], A[k]) for k in range(K)])
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9)


def test_within_track_square(oracle):
    # synthetic closed square, 4 corners, width 1.5
    tx = np.array([0.0, 10.0, 10.0, 0.0]); ty = np.array([0.0, 0.0, 10.0, 10.0]); tw = np.full(4, 1.5)
    tr = (tx, ty, tw)
    w, d = oracle.within_track(tr, [4.0, 1.0])        # nearest (0,0); neighbours (0,10) [prev, wraps] d=9.85 vs (10,0) d=6.08 -> next
    assert w and d == pytest.approx(1.0, rel=RTOL)
    w, d = oracle.within_track(tr, [4.0, -2.0])
    assert (not w) and d == pytest.approx(2.0, rel=RTOL)
    w, d = oracle.within_track(tr, [-3.0, -1.0])      # beyond the corner: projection onto the INFINITE line (t<0)
    # nearest (0,0); prev=(0,10): dist sqrt(9+121)=11.40; next=(10,0): sqrt(169+1)=13.04 -> prev, line x=0 -> dist 3
    assert (not w) and d == pytest.approx(3.0, rel=RTOL)
    w, d = oracle.within_track(tr, [1.0, 1.0])        # tie dist_m1 == dist_p1 -> prefers previous (<=): line x=0 -> 1.0
    assert w and d == pytest.approx(1.0, rel=RTOL)
    w, d = oracle.within_track(tr, [1.0, 9.0])        # nearest idx 4 (0,10): wrap-around next = idx 1
    assert w and d == pytest.approx(1.0, rel=RTOL)
    w, d = oracle.within_track(tr, [5.0, 0.5])        # equidistant to pts 1 and 2 -> findmin picks the first
    assert w and d == pytest.approx(0.5, rel=RTOL)


def test_within_track_vs_numpy(oracle, track):
    rng = np.random.default_rng(1)
    pos = np.stack([rng.uniform(-20, 270, 500), rng.uniform(-170, 160, 500)], 1)
    wn, dn = NP.within_track(track, pos)
    for k in range(500):
        w, d = oracle.within_track(track, pos[k])
        assert w == bool(wn[k]) and d == pytest.approx(dn[k], rel=1e-11, abs=1e-11)


def test_reward_reset_state(oracle, track):
    env = oracle.OracleEnv("car", 1, track=track)
    w, d = oracle.within_track(track, [0.0, 0.0])
    assert env.reward() == pytest.approx(-d + 20.0, rel=RTOL)
    s = env.state; s[3], s[4] = 1.0, 2.0; env.state = s                              # beta = atan2(2,1) > 45deg
    assert env.reward() == pytest.approx(-5000.0 - d + 2 * math.sqrt(5.0), rel=RTOL)
    s[0], s[1] = 1000.0, 1000.0; env.state = s
    _, d2 = oracle.within_track(track, [1000.0, 1000.0])
    assert env.reward() == pytest.approx(-1e6 - 5000.0 - d2 + 2 * math.sqrt(5.0), rel=RTOL)


def test_multicar_reset_and_reward(oracle, track):
    env = oracle.OracleEnv("car", 3, track=track)
    s = env.state.reshape(3, 8)
    assert list(s[:, 0]) == [0.0, 5.0, -5.0]                                         # multi-car_racing.jl:163-174
    assert np.all(s[:, 3] == 10.0) and np.allclose(s[:, 2], math.pi / 2)
    single = []
    for c in range(3):
        e1 = oracle.OracleEnv("car", 1, track=track); st = e1.state; st[:] = s[c]; e1.state = st
        single.append(e1.reward())
    # pairs: |0-5|=5, |0+5|=5, |5+5|=10 ; none <= 4
    assert env.reward() == pytest.approx(sum(single) - 20.0, rel=RTOL)
    s[1, 0] = 3.0; env.state = s.reshape(-1)                                         # car2 3 m from car1 -> -11000
    e1 = oracle.OracleEnv("car", 1, track=track); st = e1.state; st[:] = s[1]; e1.state = st
    assert env.reward() == pytest.approx(single[0] + e1.reward() + single[2] - 3.0 - 11000.0 - 5.0 - 8.0, rel=RTOL)


def test_mountaincar_step_and_reward(oracle):
    env = oracle.OracleEnv("mountaincar")
    env.state = [-0.5, 0.0]
    assert env.step([1.0]) == 0
    v = 1.0 * 0.0015 + math.cos(3 * -0.5) * (-0.0025)
    assert env.state[1] == pytest.approx(v, rel=RTOL) and env.state[0] == pytest.approx(-0.5 + v, rel=RTOL)
    assert env.reward() == pytest.approx(abs(v) - 1.0, rel=RTOL)
    env.state = [-1.19, -0.07]                                                       # wall clamp: x -> -1.2, v -> 0
    env.step([-1.0])
    assert env.state[0] == -1.2 and env.state[1] == 0.0
    env.state = [0.44, 0.07]                                                         # reach goal 0.45
    env.step([1.0])
    assert env.e.done == 1 and env.reward() == pytest.approx(100000 + env.state[1], rel=RTOL)
    assert env.step([1.5]) == -3                                                     # not in action space


def _cartpole_step_py(s, t, a):
    """independent scalar re-evaluation of RL.jl's CartPoleEnv _step! (recalled; see oracle header [3P])"""
    x, xd, th, thd = s
    force = a * 10.0
    tmp = (force + 0.05 * thd ** 2 * math.sin(th)) / 1.1
    thacc = (9.8 * math.sin(th) - math.cos(th) * tmp) / (0.5 * (4 / 3 - 0.1 * math.cos(th) ** 2 / 1.1))
    xacc = tmp - 0.05 * thacc * math.cos(th) / 1.1
    s2 = [x + 0.02 * xd, xd + 0.02 * xacc, th + 0.02 * thd, thd + 0.02 * thacc]
    t += 1
    done = abs(s2[0]) > 2.4 or abs(s2[2]) > 12 * 2 * math.pi / 360 or t > 200
    return s2, t, done


def test_cartpole_step_and_reward(oracle):
    env = oracle.OracleEnv("cartpole")
    assert env.ss == 4 and env.as_ == 1 and np.all(env.state == 0.0)
    p = oracle.cartpole_default_params()
    assert p[3] == 1.1 and p[5] == 0.05 and p[8] == pytest.approx(0.20943951023931953, rel=1e-15)
    assert env.step([1.0]) == 0
    # from rest at the origin with full push: tmp = 10/1.1, thetaacc = -tmp / (0.5 (4/3 - 0.1/1.1))
    tmp = 10 / 1.1
    thacc = -tmp / (0.5 * (4 / 3 - 0.1 / 1.1))
    xacc = tmp - 0.05 * thacc / 1.1
    np.testing.assert_allclose(env.state, [0.0, 0.02 * xacc, 0.0, 0.02 * thacc], rtol=RTOL)
    assert env.e.done == 0 and env.reward() == 1.0
    # a generic state against the scalar re-evaluation, 30 steps of bang-bang control
    s, t = [0.01, -0.02, 0.03, 0.04], 0
    env.reset(); env.state = s
    for i in range(30):
        a = 1.0 if i % 3 else -0.7
        env.step([a]); s, t, done = _cartpole_step_py(s, t, a)
        np.testing.assert_allclose(env.state, s, rtol=1e-13, atol=1e-16)
        assert env.e.t == t and bool(env.e.done) == done and env.reward() == (0.0 if done else 1.0)
    env.reset(); env.state = [2.39, 1.0, 0.0, 0.0]                                  # x leaves +-2.4
    env.step([0.0]); assert env.e.done == 1 and env.reward() == 0.0
    env.reset(); env.state = [0.0, 0.0, 0.2, 1.0]                                   # theta leaves +-12 deg
    env.step([0.0]); assert env.e.done == 1
    env.reset()                                                                     # t > max_steps (200) -> done at step 201
    for i in range(201):
        env.state = [0.0, 0.0, 0.0, 0.0]; env.step([0.0])
        assert env.e.done == (1 if i == 200 else 0)
    assert env.step([1.5]) == -3


def test_cemppi_cartpole_defaults(oracle):
    """simulate_cartpole's defaults (cartpole_example.jl:35-50): :cemppi K=20 H=15 lambda=0.1 Sigma=[1.5] N=5 elite 0.8, :mle."""
    env = oracle.OracleEnv("cartpole"); env.state = [0.01, 0.0, -0.02, 0.0]
    K, T, N = 20, 15, 5
    pol = oracle.OraclePolicy("cemppi", env, K, T, lam=0.1, U0=[0.0], cov=[1.5], N=N, elite_threshold=0.8)
    Z = np.random.default_rng(4).standard_normal((N, K, T))
    r = pol(env, Z)
    assert r["status"] == 0 and 1 <= r["iters_run"] <= N
    # costs are minus the number of not-done steps: integers in [-T, 0]
    assert np.all(r["cost"] == np.round(r["cost"])) and r["cost"].min() >= -T and r["cost"].max() <= 0
    # independent re-evaluation of the final-iteration costs from the returned E (E holds V - U_orig, U_orig = 0)
    E = r["E"]
    for k in range(K):
        s, t, c = [0.01, 0.0, -0.02, 0.0], 0, 0.0
        for tt in range(T):
            s, t, done = _cartpole_step_py(s, t, min(max(E[tt, k], -1.0), 1.0))
            c -= 0.0 if done else 1.0
        assert c == r["cost"][k]
    r2 = pol.run_trial(env, 7, num_steps=50)
    assert r2["status"] == 0 and r2["steps"] >= 1


def test_get_model_controls_and_rollout(oracle, track):
    env = oracle.OracleEnv("car", 1, track=track)
    pol = oracle.OraclePolicy("gmppi", env, 3, 4, lam=10.0, U0=[0.0, 0.0], cov=[0.0625, 0.1])
    E = np.zeros((8, 3)); E[:, 1] = 5.0; E[:, 2] = -5.0                              # clamps to +1 / -1
    cost = pol.simulate_model(np.zeros(8), E)
    for k, a in enumerate([0.0, 1.0, -1.0]):
        e = env.copy(); c = 0.0
        for t in range(4):
            e.step([a, a]); c -= e.reward()
        assert cost[k] == pytest.approx(c, rel=RTOL)


def test_roll_U_alias_quirk(oracle, track):
    """get_controls_roll_U! with pol.U === params.U0: tail (last `as`) never changes; rest shifts; not clamped."""
    env = oracle.OracleEnv("car", 1, track=track)
    T = 3
    U0 = np.array([0.1, 0.2, 0.3, 0.4, 0.5, 0.6])
    pol = oracle.OraclePolicy("gmppi", env, 4, T, lam=10.0, U0=U0, cov=[0.0625, 0.1])
    Z = np.zeros((1, 4, 6))
    r = pol(env, Z)                                # zero noise: wc == U
    assert np.allclose(r["control"], [0.1, 0.2])
    assert np.allclose(pol.U, [0.3, 0.4, 0.5, 0.6, 0.5, 0.6])
    r = pol(env, Z)
    assert np.allclose(pol.U, [0.5, 0.6, 0.5, 0.6, 0.5, 0.6])


def test_alias_table_kat(oracle):
    a, al = oracle.make_alias_table([0.5, 0.25, 0.25])
    # a = [1.5,.75,.75]; larges=[0], smalls=[1,2]; pop s=2,l=0: alias[2]=0, a[0]=1.25 -> large; pop s=1: alias[1]=0, a[0]=1.0 -> small; end: a[0]=1
    assert list(a) == [1.0, 0.75, 0.75] and al[1] == 0 and al[2] == 0
    idx = oracle.alias_sample(a, al, [0, 1, 1, 2, 2], [0.99, 0.74, 0.75, 0.1, 0.9])
    assert list(idx) == [0, 1, 0, 2, 0]
    rng = np.random.default_rng(3)
    w = rng.random(257); w /= w.sum()
    a, al = oracle.make_alias_table(w)
    an, aln = NP.alias_table(w)
    assert np.array_equal(a, an) and np.array_equal(al[a < 1.0], aln[an < 1.0])
    # the table reproduces the distribution exactly: p_i = (a_i + sum_{j: alias_j = i} (1-a_j)) / n
    p = a.copy()
    for j in range(257):
        if a[j] < 1.0:
            p[al[j]] += 1.0 - a[j]
    np.testing.assert_allclose(p / 257, w, rtol=1e-9, atol=1e-15)


def test_cma_constants_vs_numpy(oracle, track):
    for (K, T, ncars) in [(4096, 50, 3), (150, 50, 1)]:
        env = oracle.OracleEnv("car", ncars, track=track)
        pol = oracle.OraclePolicy("cmamppi", env, K, T, lam=10.0, U0=np.zeros(2 * ncars),
                                  cov=np.tile([0.0625, 0.1], ncars), N=10, cma_sigma=0.75)
        c = NP.cma_constants(K, 2 * ncars * T, 0.8)
        assert pol.p.m_elite == c["m_elite"]
        for f in ("mu_eff", "c_sigma", "d_sigma", "c_Sigma", "c1", "c_mu", "E_cma"):
            assert getattr(pol.p, f) == pytest.approx(c[f], rel=1e-12), f
        np.testing.assert_allclose(pol.cma_ws, c["ws"], rtol=1e-11)
        assert abs(pol.cma_ws[:c["m_elite"]].sum() - 1) < 1e-12


def test_linear_algebra_standins(oracle):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((12, 12)); S = A @ A.T + 0.5 * np.eye(12)
    rc, L = oracle.cholesky_lower(S)
    assert rc == 0
    np.testing.assert_allclose(L, np.linalg.cholesky(S), rtol=1e-12, atol=1e-13)
    rc, C = oracle.sym_pow(S, -0.5)
    lam, V = np.linalg.eigh(S)
    np.testing.assert_allclose(C, (V * lam ** -0.5) @ V.T, rtol=1e-10, atol=1e-12)
    rc, _ = oracle.cholesky_lower(np.array([[1.0, 2.0], [2.0, 1.0]]))
    assert rc == -2                                                                  # PosDefException analogue


def test_quantile_ci(oracle):
    x = np.arange(1.0, 65.0)
    lo, med, hi = oracle.quantile_ci(x)
    # n=64,q=.5: j=ceil(32-1.96*4)=ceil(24.16)=25, k=ceil(39.84)=40
    assert (lo, med, hi) == (25.0, 32.5, 40.0)


@pytest.mark.parametrize("kind", ["gmppi", "muaismppi", "musigmaaismppi", "cemppi", "pmcmppi", "cmamppi"])
def test_policy_call_vs_numpy(oracle, track, kind):
    """Whole pol(env) against the vectorised NumPy/LAPACK re-derivation, injected noise."""
    rng = np.random.default_rng(11)
    K, T, N = 48, 6, 4
    env = oracle.OracleEnv("car", 1, track=track)
    cs = 2 * T
    U0 = rng.uniform(-0.2, 0.2, cs)
    pol = oracle.OraclePolicy(kind, env, K, T, lam=10.0, alpha=1.0, U0=U0, cov=[0.0625, 0.1], N=N,
                              lam_ais=20.0, elite_threshold=0.8, cma_sigma=0.75)
    Z = rng.standard_normal((N, K, cs))
    di = rng.integers(0, K, (N, K)).astype(np.int32)
    du = rng.random((N, K))
    r = pol(env, Z, di, du)
    ref = NP.policy_call(kind, NP.car_params(), track, 1, env.state, U0, np.diag(np.tile([0.0625, 0.1], T)), Z, K, T,
                         10.0, N=N, lam_ais=20.0, thr=0.8, cma_sigma=0.75, res=(di, du))
    assert r["status"] == 0 and r["iters_run"] == ref["iters_run"]
    np.testing.assert_allclose(r["cost"], ref["cost"], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(r["weights"], ref["weights"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(r["E"], ref["E"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(r["control"], ref["control"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(pol.U, ref["U_next"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(r["Sigma_last"], ref["Sigma_last"], rtol=1e-6, atol=1e-10)
    if kind == "pmcmppi":
        assert np.array_equal(r["res_idx0"][:N - 1], ref["res_idx0"])


def test_policy_control_cost_alpha(oracle, track):
    """gamma = lambda(1-alpha) != 0 exercises Sigma_inv and the unclamped-V control cost (:272)."""
    rng = np.random.default_rng(13)
    K, T = 16, 5
    env = oracle.OracleEnv("car", 1, track=track)
    U0 = rng.uniform(-0.3, 0.3, 2 * T)
    pol = oracle.OraclePolicy("gmppi", env, K, T, lam=10.0, alpha=0.9, U0=U0, cov=[0.0625, 0.1])
    Z = rng.standard_normal((1, K, 2 * T))
    r = pol(env, Z)
    ref = NP.policy_call("gmppi", NP.car_params(), track, 1, env.state, U0, np.diag(np.tile([0.0625, 0.1], T)), Z, K, T, 10.0, alpha=0.9)
    np.testing.assert_allclose(r["cost"], ref["cost"], rtol=1e-9)
    np.testing.assert_allclose(r["control"], ref["control"], rtol=1e-9)


def test_mppi_mountaincar_config1(oracle):
    """BASELINE config 1: MountainCar :mppi K=20 H=15 lambda=0.1, Sigma=[1.5] -- plumbing."""
    env = oracle.OracleEnv("mountaincar")
    env.state = [-0.5, 0.0]
    K, T = 20, 15
    pol = oracle.OraclePolicy("mppi", env, K, T, lam=0.1, alpha=1.0, U0=[0.0], cov=[1.5])
    Z = np.random.default_rng(2).standard_normal((T, K, 1))
    r = pol(env, Z)
    E = math.sqrt(1.5) * Z
    np.testing.assert_allclose(r["E"], E, rtol=1e-15)
    # independent scalar re-evaluation of the costs
    cost = np.zeros(K)
    for k in range(K):
        x, v, t, done = -0.5, 0.0, 0, False
        for tt in range(T):
            a = min(max(E[tt, k, 0], -1.0), 1.0)
            t += 1
            v += a * 0.0015 + math.cos(3 * x) * (-0.0025); v = min(max(v, -0.07), 0.07)
            x += v; x = min(max(x, -1.2), 0.6)
            if x == -1.2 and v < 0: v = 0
            done = (x >= 0.45 and v >= 0.0) or t >= 200
            cost[k] -= (100000 if (x >= 0.45 and v >= 0) else 0) + abs(v) + (0.0 if done else -1.0)
    np.testing.assert_allclose(r["cost"], cost, rtol=1e-12)
    w = NP.compute_weights(0.1, cost)
    wn = np.einsum("k,tk->t", w, E[:, :, 0])
    assert r["control"][0] == pytest.approx(min(max(wn[0], -1), 1), rel=1e-10)
    np.testing.assert_allclose(pol.U[:-1], wn[1:], rtol=1e-10, atol=1e-14)
    assert pol.U[-1] == 0.0


def test_closed_loop_smoke(oracle, track):
    env = oracle.OracleEnv("car", 1, track=track)
    pol = oracle.OraclePolicy("cemppi", env, 64, 20, lam=10.0, U0=[0.0, 0.0], cov=[0.0625, 0.1], N=4, nthreads=4)
    r = pol.run_trial(env, 20240001, num_steps=25, laps=2)
    assert r["status"] == 0 and r["steps"] == 25 and r["trk_viol"] == 0 and r["mean_v"] > 9.0
    env2 = oracle.OracleEnv("mountaincar"); env2.state = [-0.5, 0.0]
    pol2 = oracle.OraclePolicy("mppi", env2, 20, 15, lam=0.1, U0=[0.0], cov=[1.5])
    r2 = pol2.run_trial(env2, 5, num_steps=200)
    assert r2["status"] == 0 and r2["steps"] >= 1
