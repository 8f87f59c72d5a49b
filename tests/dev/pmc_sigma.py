"""Where does the Σ′ deviation of :pmcmppi at cs = 300 come from?  N = 2: Σ′ = cov(E1[:, idx]) + 1e-8 I is recomputable in numpy."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from mpopis_amd import engine as eng_mod
track = O.load_track()
for (ncars, K) in ((3, 1024), (3, 4096), (1, 1024)):
    T, N, B = 50, 2, 1
    cs = 2 * ncars * T
    cov = np.tile([0.0625, 0.1], ncars)
    eng = eng_mod.Engine("car", ncars, "pmcmppi", K, T, batch=B, lam=10.0, ais_its=N, lam_ais=20.0, cov=cov, track=track)
    rng = np.random.default_rng(5)
    Z = rng.standard_normal((B, N, K, cs)); di = rng.integers(0, K, (B, 1, K)).astype(np.int32); du = rng.random((B, 1, K))
    got = eng.policy_step(Z, di, du, want_E=True)
    Sd = eng.get_Sigma()[0]
    env = O.OracleEnv("car", ncars, track=track)
    pol = O.OraclePolicy("pmcmppi", env, K, T, lam=10.0, U0=np.zeros(2 * ncars), cov=cov, N=N, lam_ais=20.0, nthreads=8)
    ref = pol(env, Z[0], di[0], du[0])
    idx = got["res_idx0"][0][0]
    assert np.array_equal(idx, ref["res_idx0"][0])
    E1 = Z[0, 0] * np.sqrt(np.tile(cov, T))            # (K, cs)
    X = E1[idx].astype(np.longdouble)
    mu = X.mean(0); Xc = X - mu
    Sn = np.array((Xc.T @ Xc) / (K - 1), dtype=np.float64) + 10e-9 * np.eye(cs)
    So = ref["Sigma_last"]
    print("ncars=%d K=%d: |dev-np|=%.2e |orc-np|=%.2e |dev-orc|=%.2e  diag: dev-np %.2e orc-np %.2e  offdiag dev-np %.2e" % (
        ncars, K, np.abs(Sd - Sn).max(), np.abs(So - Sn).max(), np.abs(Sd - So).max(),
        np.abs(np.diag(Sd - Sn)).max(), np.abs(np.diag(So - Sn)).max(), np.abs(Sd - Sn - np.diag(np.diag(Sd - Sn))).max()))
    eng.close()
