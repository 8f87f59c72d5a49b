import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
from mpopis_amd.engine import Engine
from mpopis_amd._lib import MPOPISError
for N in (2, 3, 4, 5, 6, 10):
    eng = Engine("car", 1, "cmamppi", 4096, 50, batch=1, lam=10.0, ais_its=N, cma_sigma=0.75, cov=[0.0625, 0.1], seed=20240000)
    try:
        got = eng.policy_step(None)
        print(N, "ok iters", got["iters_run"], "control", got["control"][0])
    except MPOPISError as e:
        print(N, "raised", e)
    eng.close()
