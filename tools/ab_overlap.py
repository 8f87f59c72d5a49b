"""Same-box comparison of the multi-stream schedule: bench workload (C5, 64 trials) split into 1 (single stream), 2, 3, 4 parts.
Prints ms per step and the per-class kernel times (under that schedule) of each mode."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpopis_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for ns in (1, 2, 3, 4):
    eng = Engine("car", 1, "musigmaaismppi", 4096, 50, batch=B, lam=10.0, ais_its=10, lam_ais=20.0, cov=[0.0625, 0.1], seed=20240000)
    eng.set_overlap(ns)
    eng.bench_policy_steps(3)
    ms, rl = eng.bench_policy_steps(20)
    eng.timing_enable(True); eng.timing_reset(); eng.bench_policy_steps(3); tm = eng.timing_read()
    print("parts=%d: %.3f ms/step  %.3e rollouts/s   %s" % (ns, ms / 20, rl / (ms * 1e-3),
          {k: (round(v[0] / max(v[1], 1) * 1e3, 1), v[1]) for k, v in tm.items() if v[1]}))
    eng.close()
