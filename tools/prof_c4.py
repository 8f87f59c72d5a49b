import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpopis_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10           # MPC steps after the first (PMC passes serialise every dispatch: keep them short)
eng = Engine("car", 3, "cmamppi", 4096, 50, batch=B, lam=10.0, ais_its=10, cma_sigma=0.75, cov=np.tile([0.0625, 0.1], 3), seed=20240000)
t = time.time(); rec = eng.run_trials(num_steps=STEPS, laps=2); dt = time.time() - t
print("C4 cmamppi 3-car K=4096 H=50 B=%d: %.1f ms per MPC step, %.3e rollouts/s" % (B, dt / (STEPS + 1) * 1e3, rec[:, 14].sum() / dt))
eng.close()
