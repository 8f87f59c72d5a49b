"""C5 (1-car :μΣaismppi K=4096 H=50 N=10) at small batches (dev tool): step time, per-class kernel time (HIP events around every launch) and how much of
the step is spent INSIDE kernels -- what a persistent single-launch form could at most win is the rest.   usage: python tools/c5_small.py 8 [1 2 4 16 ...]"""
import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
from mpopis_amd.engine import Engine
for B in [int(a) for a in sys.argv[1:]] or [8]:
    eng = Engine("car", 1, "musigmaaismppi", 4096, 50, batch=B, lam=10.0, ais_its=10, lam_ais=20.0, cov=[0.0625, 0.1], seed=20240000)
    eng.bench_policy_steps(20)
    runs = sorted(eng.bench_policy_steps(20) for _ in range(5))
    ms = runs[2][0] / 20
    eng.timing_enable(True); eng.timing_reset()
    ms_t, _ = eng.bench_policy_steps(10)
    tm = eng.timing_read()
    eng.timing_enable(False)
    per = {k: v[0] / 10 for k, v in tm.items() if v[1]}
    n = {k: v[1] // 10 for k, v in tm.items() if v[1]}
    tot = sum(per.values())
    print("C5 B=%d: %.3f ms/step (with events %.3f); kernel classes per step [ms x launches]: %s; sum %.3f ms = %.1f %% of the timed step, %d launches-classes"
          % (B, ms, ms_t / 10, " ".join("%s=%.3fx%d" % (k, per[k], n[k]) for k in per), tot, 100 * tot / (ms_t / 10), sum(n.values())))
    eng.close()
