"""Closed-loop step time over a lap (mpopis_run_trials in windows; policy step + env step per MPC step, no host round trip).
usage (GPU box): python tools/closed_loop_bench.py [trials] [policy] [K] [N] [cars] [windows x steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpopis_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pol = sys.argv[2] if len(sys.argv) > 2 else "μΣaismppi"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
N = int(sys.argv[4]) if len(sys.argv) > 4 else 10
cars = int(sys.argv[5]) if len(sys.argv) > 5 else 1
nwin, wsteps = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (6, 50)
kw = dict(sigma_est="ss", elite_threshold=0.8) if pol == "cemppi" else (dict(elite_threshold=0.8, cma_sigma=0.75) if pol == "cmamppi" else {})
eng = Engine("car", cars, pol, K, 50, batch=B, lam=10.0, alpha=1.0, ais_its=N, lam_ais=20.0, cov=np.tile([0.0625, 0.1], cars), seed=20240000, **kw)
eng.bench_policy_steps(10)
tot = 0.0
for w in range(nwin):
    t0 = time.perf_counter(); eng.run_trials(num_steps=wsteps - 1, laps=4); dt = time.perf_counter() - t0
    tot += dt
    x = eng.get_state()[0]
    print("steps %3d..%3d: %.3f ms per MPC step | slot0 Vx=%.1f" % (w * wsteps, (w + 1) * wsteps, dt / wsteps * 1e3, x[0, 3]), flush=True)
print("whole: %.3f ms per MPC step of %d trials (COMPACT=%s)" % (tot / (nwin * wsteps) * 1e3, B, os.environ.get("MPOPIS_ROLLOUT_COMPACT", "default")))
eng.close()
