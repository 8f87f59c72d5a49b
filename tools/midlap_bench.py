"""Step time of the headline workload at the reset state and at mid-lap states (after a resident closed loop of n MPC steps): the rollout
kernel's nearest-point search has a straight-line fast path whose hit rate depends on where the cars are.
usage (GPU box): python tools/midlap_bench.py [trials] [policy] [K] [N] [cars]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mpopis_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pol = sys.argv[2] if len(sys.argv) > 2 else "μΣaismppi"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
N = int(sys.argv[4]) if len(sys.argv) > 4 else 10
cars = int(sys.argv[5]) if len(sys.argv) > 5 else 1
kw = dict(sigma_est="ss", elite_threshold=0.8) if pol == "cemppi" else (dict(elite_threshold=0.8, cma_sigma=0.75) if pol == "cmamppi" else {})
eng = Engine("car", cars, pol, K, 50, batch=B, lam=10.0, alpha=1.0, ais_its=N, lam_ais=20.0, cov=np.tile([0.0625, 0.1], cars), seed=20240000, **kw)
eng.bench_policy_steps(20)


def point(tag):
    eng.timing_enable(True); eng.timing_reset()
    ms, _ = eng.bench_policy_steps(10)
    tm = eng.timing_read(); eng.timing_enable(False)
    x = eng.get_state()[0]
    print("%-28s %.3f ms/step  rollout avg launch %.1f us   |  slot0 x=%.1f y=%.1f Vx=%.2f" % (tag, ms / 10, tm["rollout"][0] / max(1, tm["rollout"][1]) * 1e3, x[0, 0], x[0, 1], x[0, 3]), flush=True)


point("reset state")
done = 0
for n in (10, 30, 60, 100, 100):
    eng.set_U(np.zeros((B, 2 * cars * 50))) if False else None
    eng.run_trials(num_steps=n - 1, laps=4)
    done += n
    point("after %d closed-loop steps" % done)
eng.close()
