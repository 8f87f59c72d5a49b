"""Same-box sweep of the part-chain schedule: ms per MPC step (median of 5 x 10 steps) for parts = 1..4 over policies and resident-trial counts.
usage (GPU box): python tools/split_sweep.py [policy,policy,...] [B,B,...] [K]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpopis_amd.engine import Engine
pols = (sys.argv[1] if len(sys.argv) > 1 else "musigmaaismppi").split(",")
Bs = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "32,48,64,96,128").split(",")]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
for pol in pols:
    kw = dict(sigma_est="ss", elite_threshold=0.8) if pol == "cemppi" else {}
    for B in Bs:
        row = []
        for ns in (1, 2, 3, 4):
            eng = Engine("car", 1, pol, K, 50, batch=B, lam=10.0, ais_its=10, lam_ais=20.0, cov=[0.0625, 0.1], seed=20240000, **kw)
            eng.set_overlap(ns)
            eng.bench_policy_steps(10)
            ms = sorted(eng.bench_policy_steps(10)[0] / 10 for _ in range(5))[2]
            row.append(ms)
            eng.close()
        print("%-16s K=%d B=%3d  parts 1/2/3/4: %s ms/step   best %d (%.1f %% vs 1)" % (pol, K, B, " ".join("%.3f" % v for v in row), (1, 2, 3, 4)[int(np.argmin(row))],
              100 * (row[0] / min(row) - 1)), flush=True)
