"""Helper process for tests/test_gpu_wcov_rows.py: 1-car H = 50 handles (cs = 100: seven row tiles) whose AIS iterations go through the scatter kernel with
weights from costs (μΣ-AIS), gathered elite columns (CE, K = 4096: 819 columns) and resampled columns (PMC); a few policy steps each; prints one JSON
line with controls, costs, weights and Σ' as hex.  MPOPIS_WCOV_ROWS / MPOPIS_KSPLIT are read once per process by the library, hence the subprocess."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from mpopis_amd.engine import Engine

out = {}
for name, pol, K, B in (("musigma", "musigmaaismppi", 4096, 3), ("ce", "cemppi", 4096, 2), ("pmc", "pmcmppi", 1000, 2), ("mu", "muaismppi", 700, 2)):
    eng = Engine("car", 1, pol, K, 50, batch=B, lam=10.0, ais_its=4, cov=np.array([0.0625, 0.1]), seed=777)
    rec = []
    for _ in range(2):
        got = eng.policy_step(None)
        rec.append([got["control"].tobytes().hex(), got["cost"].tobytes().hex(), got["weights"].tobytes().hex(), eng.get_Sigma().tobytes().hex()])
    out[name] = rec
    eng.close()
print(json.dumps(out))
