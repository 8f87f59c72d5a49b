"""C4 (3-car :cmamppi K=4096 H=50 N=10) closed loop: per-class kernel time per MPC step for a batch size, in the schedule given by MPOPIS_NSPLIT
(1 = one stream: per-launch durations are kernel-in-isolation figures).  usage: MPOPIS_NSPLIT=1 python tools/c4prof.py 64"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpopis_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = Engine("car", 3, "cmamppi", 4096, 50, batch=B, lam=10.0, ais_its=10, elite_threshold=0.8, cma_sigma=0.75, cov=np.tile([0.0625, 0.1], 3), seed=20240000)
eng.run_trials(num_steps=1, laps=2)
eng.reset(); eng.set_U(np.zeros((B, 300))); eng.seed(20240000)
t0 = time.perf_counter(); eng.run_trials(num_steps=5, laps=2); dt = time.perf_counter() - t0
eng.reset(); eng.set_U(np.zeros((B, 300))); eng.seed(20240000)
eng.timing_enable(True); eng.timing_reset()
eng.run_trials(num_steps=3, laps=2)
tm = eng.timing_read()
print("NSPLIT=%s B=%d: %.2f ms per MPC step; per class ms/step (avg launch us, launches/step): %s" % (
    os.environ.get("MPOPIS_NSPLIT", "auto"), B, dt / 6 * 1e3, {k: (round(v[0] / 4, 3), round(v[0] / max(v[1], 1) * 1e3, 1), v[1] // 4) for k, v in tm.items() if v[1]}))
eng.close()
