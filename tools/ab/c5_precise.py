"""C5 headline workload, median of 15 x 10 steps (for A/B runs: tools/ab/run_py.sh tools/ab/c5_precise.py)"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from mpopis_amd.engine import Engine
eng = Engine("car", 1, "musigmaaismppi", 4096, 50, batch=64, lam=10.0, ais_its=10, lam_ais=20.0, cov=[0.0625, 0.1], seed=20240000)
eng.bench_policy_steps(40)
ms = sorted(eng.bench_policy_steps(10)[0] / 10 for _ in range(15))
eng.timing_enable(2); eng.timing_reset(); eng.bench_policy_steps(10); tm = eng.timing_read()
print("C5 64 trials: median %.4f ms/step (min %.4f max %.4f); rollout %.1f us per launch" % (ms[7], ms[0], ms[-1], tm["rollout"][0] / tm["rollout"][1] * 1e3))
eng.close()
