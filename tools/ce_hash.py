import sys, os, hashlib; sys.path.insert(0, os.getcwd())
import numpy as np
from mpopis_amd.engine import Engine
for est in ("mle", "ss", "oas"):
    for K, B in ((150, 3), (256, 2), (40, 1)):
        eng = Engine("car", 1, "cemppi", K, 50, batch=B, lam=10.0, ais_its=10, elite_threshold=0.8, sigma_est=est, cov=[0.0625, 0.1], seed=7)
        h = hashlib.sha256()
        for _ in range(3):
            g = eng.policy_step(None)
            h.update(g["control"].tobytes()); h.update(g["cost"].tobytes()); h.update(g["weights"].tobytes()); h.update(g["iters_run"].tobytes()); h.update(eng.get_Sigma().tobytes())
        print(est, K, B, h.hexdigest()[:16], g["iters_run"])
        eng.close()
