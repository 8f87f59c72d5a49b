"""dev: engine vs oracle on configurations at the edges of the parameter ranges (what the randomised sweep does not draw): degenerate elite sets,
a single sample, extreme lambda, alpha != 1 with adaptive policies.  Prints status / iteration counts / control deviation per case.
usage (GPU box): python tests/dev/edge_cases.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from mpopis_amd.engine import Engine
from mpopis_amd._lib import MPOPISError
track = O.load_track()
cases = [
    dict(kind="cemppi", K=150, T=10, N=4, elite_threshold=1.0),
    dict(kind="cemppi", K=150, T=10, N=4, elite_threshold=0.995),
    dict(kind="cemppi", K=150, T=10, N=4, elite_threshold=0.0),
    dict(kind="cemppi", K=150, T=10, N=4, elite_threshold=0.5, sigma_est="oas"),
    dict(kind="cmamppi", K=150, T=10, N=3, elite_threshold=0.99),
    dict(kind="cmamppi", K=150, T=10, N=3, elite_threshold=0.0),
    dict(kind="musigmaaismppi", K=2, T=10, N=3),
    dict(kind="musigmaaismppi", K=1, T=10, N=3),
    dict(kind="pmcmppi", K=2, T=10, N=3),
    dict(kind="pmcmppi", K=1, T=5, N=3),
    dict(kind="musigmaaismppi", K=256, T=10, N=3, lam=1e-6, lam_ais=1e-6),
    dict(kind="musigmaaismppi", K=256, T=10, N=3, lam=1e12, lam_ais=1e12),
    dict(kind="muaismppi", K=256, T=10, N=3, alpha=0.3),
    dict(kind="cemppi", K=256, T=10, N=3, alpha=0.0),
    dict(kind="gmppi", K=64, T=1, N=1),
    dict(kind="imppi", K=64, T=1, N=3),
]
rng = np.random.default_rng(1)
for c in cases:
    kind, K, T, N = c["kind"], c["K"], c["T"], c["N"]
    kw = {k: v for k, v in c.items() if k not in ("kind", "K", "T", "N")}
    lam = kw.pop("lam", 10.0); lam_ais = kw.pop("lam_ais", 20.0); alpha = kw.pop("alpha", 1.0)
    cs = 2 * T
    Z = rng.standard_normal((1, N if kind != "gmppi" else 1, K, cs))
    di = rng.integers(0, K, (1, max(N - 1, 1), K)).astype(np.int32); du = rng.random((1, max(N - 1, 1), K))
    env = O.OracleEnv("car", 1, track=track)
    try:
        pol = O.OraclePolicy(kind, env, K, T, lam=lam, alpha=alpha, U0=np.zeros(2), cov=[0.0625, 0.1], N=N, lam_ais=lam_ais, cma_sigma=0.75, **kw)
        r = pol(env, Z[0], di[0], du[0]); ost = r["status"]
    except Exception as e:
        r, ost = None, "ctor:" + str(e)[:40]
    try:
        eng = Engine("car", 1, kind, K, T, batch=1, lam=lam, alpha=alpha, ais_its=N, lam_ais=lam_ais, cma_sigma=0.75, cov=[0.0625, 0.1], track=track, **kw)
        try:
            g = eng.policy_step(Z, di, du); est = 0
        except MPOPISError as e:
            g, est = None, e.code
        eng.close()
    except MPOPISError as e:
        g, est = None, "ctor:%d" % e.code
    line = "%-16s %-50s oracle %-10s engine %-10s" % (kind, str({k: v for k, v in c.items() if k != "kind"}), ost, est)
    if r is not None and g is not None and ost == 0 and est == 0:
        line += " iters %d/%d control err %.2e finite %s/%s" % (r["iters_run"], g["iters_run"][0], float(np.max(np.abs(g["control"][0] - r["control"]))), np.isfinite(r["control"]).all(), np.isfinite(g["control"]).all())
    print(line, "" if str(ost) == str(est) else "   <-- MISMATCH")
