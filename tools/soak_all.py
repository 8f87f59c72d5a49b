import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpopis_amd as M
t0 = time.time()
for rep in range(3):
    for pt in [":mppi", ":gmppi", ":imppi", ":cemppi", ":cmamppi", ":μaismppi", ":μΣaismppi", ":pmcmppi"]:
        for ncars in (1, 3):
            try:
                rec, _ = M.simulate_car_racing(num_trials=16, num_steps=150, num_cars=ncars, policy_type=pt, num_samples=2048, horizon=50, ais_its=8, seed=1000 * rep + 17, quiet=True)
                st = rec[:, 16].min()
                print(rep, pt, ncars, "steps med %d" % np.median(rec[:, 2]), "status", st, "viol", rec[:, 12:15].sum(0), flush=True)
                assert st != -4
            except Exception as e:
                print(rep, pt, ncars, "raised", repr(e)[:100], flush=True); assert "-4" not in repr(e)
print("done %.0f s" % (time.time() - t0))
