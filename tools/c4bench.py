"""C4 (3-car :cmamppi K=4096 H=50 N=10) closed loop at a list of batch sizes, warm (dev tool).  usage: python tools/c4bench.py 64 [32 ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpopis_amd.engine import Engine
for B in [int(a) for a in sys.argv[1:]] or [64]:
    eng = Engine("car", 3, "cmamppi", 4096, 50, batch=B, lam=10.0, ais_its=10, elite_threshold=0.8, cma_sigma=0.75, cov=np.tile([0.0625, 0.1], 3), seed=20240000)
    eng.run_trials(num_steps=1, laps=2)
    best = 1e9
    for rep in range(3):
        eng.reset(); eng.set_U(np.zeros((B, 300))); eng.seed(20240000)
        t0 = time.perf_counter(); rec = eng.run_trials(num_steps=9, laps=2); dt = time.perf_counter() - t0
        best = min(best, dt)
    print("C4 B=%d: %.2f ms per MPC step, %.3e rollouts/s (best of 3 x 10 steps)" % (B, best / 10 * 1e3, rec[:, 14].sum() / best))
    eng.close()
