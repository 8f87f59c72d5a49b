"""dev: the oracle under AddressSanitizer / UBSan -- every policy x 1-4 cars x six shapes x degenerate elite sets x the five estimators, two policy calls + a short trial\neach (tests/dev/oracle_sanitizers.sh builds the instrumented library and runs this and the CPU oracle tests with it)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
track = O.load_track()
rng = np.random.default_rng(0)
n = 0
for kind in ["mppi", "gmppi", "imppi", "cemppi", "cmamppi", "muaismppi", "musigmaaismppi", "pmcmppi"]:
    for ncars in (1, 2, 3, 4):
        for K, T, N in ((1, 1, 2), (2, 3, 3), (17, 5, 3), (65, 10, 4), (150, 16, 3), (256, 25, 2)):
            for elite in (0.0, 0.5, 0.8, 0.99, 1.0):
                if kind not in ("cemppi", "cmamppi") and elite != 0.8:
                    continue
                for est in (("mle", "ss", "lw", "rblw", "oas") if kind == "cemppi" else ("mle",)):
                    env = O.OracleEnv("car", ncars, track=track)
                    for _ in range(int(rng.integers(0, 5))):
                        env.step(np.clip(rng.normal(0, 0.5, 2 * ncars), -1, 1))
                    Nn = 1 if kind in ("mppi", "gmppi") else N
                    pol = O.OraclePolicy(kind, env, K, T, lam=10.0, alpha=float(rng.choice([1.0, 0.5])), U0=np.zeros(2 * ncars), cov=np.tile([0.0625, 0.1], ncars), N=Nn,
                                         lam_ais=20.0, elite_threshold=elite, sigma_est=est, cma_sigma=0.75, nthreads=2)
                    cs = 2 * ncars * T
                    for step in range(2):
                        Z = rng.standard_normal((T, K, 2 * ncars)) if kind == "mppi" else rng.standard_normal((Nn, K, cs))
                        di = rng.integers(0, K, (max(Nn - 1, 1), K)).astype(np.int32); du = rng.random((max(Nn - 1, 1), K))
                        r = pol(env, Z, di, du)
                        n += 1
                        if r["status"] == 0:
                            env.step(r["control"])
                    r = pol.run_trial(env, 5, num_steps=3, laps=1)
print("oracle sweep under sanitizers:", n, "policy calls ok")
