import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
from mpopis_amd.engine import Engine
for est in ("mle", "ss", "oas"):
    for B in (1, 64):
        eng = Engine("car", 1, "cemppi", 150, 50, batch=B, lam=10.0, ais_its=10, elite_threshold=0.8, sigma_est=est, cov=[0.0625, 0.1], seed=20240000)
        eng.bench_policy_steps(3)
        eng.timing_enable(True); eng.timing_reset(); eng.bench_policy_steps(5); tm = eng.timing_read(); eng.timing_enable(False)
        ms, rl = eng.bench_policy_steps(20)
        print(est, B, "%.3f ms/step" % (ms / 20), "moments %.1f us" % (tm["moments"][0] / max(tm["moments"][1], 1) * 1e3))
        eng.close()
