import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpopis_amd import build; build.build()
from mpopis_amd.engine import Engine
from mpopis_amd._lib import MPOPISError
for (nc, K, B) in [(3, 4096, 4), (1, 4096, 4), (1, 150, 4)]:
    eng = Engine("car", nc, "cmamppi", K, 50, batch=B, lam=10.0, ais_its=10, cma_sigma=0.75, cov=np.tile([0.0625, 0.1], nc), seed=20240000)
    t = time.time()
    try:
        rec = eng.run_trials(num_steps=30, laps=2)
        print("cma cars=%d K=%d: run_trials ok %.2fs steps=%s rollouts=%s" % (nc, K, time.time() - t, rec[:, 1], rec[:, 14]))
    except MPOPISError as e:
        print("cma cars=%d K=%d: %s" % (nc, K, e))
    eng.close()
