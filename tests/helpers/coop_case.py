"""Helper process for tests/test_gpu_coop_fallback.py: a cs = 300 :cmamppi handle (3 cars), a few policy steps and a short closed loop;
prints one JSON line.  Environment variables (MPOPIS_COOP_*) are read once per process by the library, hence the subprocess."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from mpopis_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
loop = int(sys.argv[4]) if len(sys.argv) > 4 else 0
eng = Engine("car", 3, "cmamppi", K, 50, batch=B, lam=10.0, ais_its=4, elite_threshold=0.8, cma_sigma=0.75, cov=np.tile([0.0625, 0.1], 3), seed=777)
out = {"control": [], "status": 0}
for _ in range(steps):
    got = eng.policy_step(None)
    out["control"].append(got["control"].tolist())
    out["iters"] = got["iters_run"].tolist()
if loop:
    rec, act = eng.run_trials(loop, 2, log_actions=True)
    out["loop_status"] = rec[:, 15].tolist()
    out["loop_actions"] = act[:, :4].tolist()
eng.close()
print(json.dumps(out))
