"""Soak run of the cooperative (multi-workgroup) Cholesky / Lanczos kernels through closed loops: many launches, error -4 (bounded wait expired) must never appear."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpopis_amd as M
t0 = time.time()
for pt, steps, trials in [(":μΣaismppi", 300, 8), (":cemppi", 300, 8), (":pmcmppi", 200, 4), (":cmamppi", 40, 8), (":cmamppi", 40, 1)]:
    t = time.time()
    try:
        rec, summ = M.simulate_car_racing(num_trials=trials, num_steps=steps, num_cars=3, policy_type=pt, num_samples=1024, horizon=50, ais_its=6, seed=7, quiet=True)
        print(pt, "trials", trials, "steps", rec[:, 2].astype(int), "min status", rec[:, 16].min(), "%.1f s" % (time.time() - t))
        assert rec[:, 16].min() != -4
    except Exception as e:
        print(pt, "raised:", repr(e)[:150]); assert "-4" not in repr(e)
print("soak done in %.0f s" % (time.time() - t0))

# several handles on the same device from concurrent host threads: cooperative clusters of different handles share the chip
import threading
res = {}
def worker(i, pt):
    try:
        rec, _ = M.simulate_car_racing(num_trials=6, num_steps=120, num_cars=3, policy_type=pt, num_samples=1024, horizon=50, ais_its=6, seed=100 + i, quiet=True)
        res[i] = float(rec[:, 16].min())
    except Exception as e:
        res[i] = repr(e)[:120]
ths = [threading.Thread(target=worker, args=(i, pt)) for i, pt in enumerate([":μΣaismppi", ":cemppi", ":cmamppi", ":μΣaismppi"])]
t = time.time()
for th in ths: th.start()
for th in ths: th.join()
print("4 concurrent handles:", res, "%.1f s" % (time.time() - t))
assert all(not (isinstance(v, float) and v == -4.0) and "-4" not in str(v) for v in res.values())
