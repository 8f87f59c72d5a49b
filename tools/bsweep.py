import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpopis_amd.engine import Engine
for B in (8, 16, 32, 48, 64, 80, 96, 128, 192):
    eng = Engine("car", 1, "gmppi", 4096, 50, batch=B, lam=10.0, cov=[0.0625, 0.1], seed=1)
    eng.bench_policy_steps(2); eng.timing_enable(True); eng.timing_reset(); eng.bench_policy_steps(5)
    tm = eng.timing_read(); us = tm["rollout"][0] / tm["rollout"][1] * 1e3
    print("B=%3d waves/SIMD=%.2f rollout %.1f us  -> %.3e rollouts/s (kernel only), us per wave-round %.1f" % (B, B * 64 / 1024, us, B * 4096 / (us * 1e-6), us / (B * 64 / 1024)))
    eng.close()
