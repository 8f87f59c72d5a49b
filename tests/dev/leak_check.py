"""dev: device memory / streams over many handle lifetimes (create, a few calls incl. the multi-stream schedule and the side streams, destroy).
usage (GPU box): python tests/dev/leak_check.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpopis_amd.engine import Engine
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2 ** 20
shapes = [("musigmaaismppi", 1, 4096, 50, 64), ("cmamppi", 3, 1024, 50, 4), ("pmcmppi", 1, 9000, 10, 2), ("cemppi", 2, 256, 16, 8), ("gmppi", 1, 1024, 50, 1), ("mppi", 1, 64, 10, 2)]
def cycle():
    for pol, cars, K, T, B in shapes:
        e = Engine("car", cars, pol, K, T, batch=B, lam=10.0, ais_its=3, lam_ais=20.0, elite_threshold=0.8, cma_sigma=0.75, cov=np.tile([0.0625, 0.1], cars), seed=1)
        e.policy_step(None); e.run_trials(num_steps=2, laps=1)
        e.close()
cycle()
m0 = free_mb()
for i in range(40):
    cycle()
    if i % 10 == 9:
        print("after %3d cycles of %d handles: free %.0f MiB (start %.0f)" % (i + 1, len(shapes), free_mb(), m0), flush=True)
m1 = free_mb()
print("drift over 240 handle lifetimes: %.1f MiB" % (m0 - m1))
assert m0 - m1 < 64, "device memory leaks"
