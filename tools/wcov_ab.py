import sys, os, hashlib; sys.path.insert(0, os.getcwd())
import numpy as np
from mpopis_amd.engine import Engine
for pol, K, B in (("musigmaaismppi", 4096, 3), ("cemppi", 1000, 2), ("pmcmppi", 777, 2), ("muaismppi", 2048, 2)):
    eng = Engine("car", 1, pol, K, 50, batch=B, lam=10.0, ais_its=4, cov=[0.0625, 0.1], seed=99)
    h = hashlib.sha256()
    for _ in range(2):
        g = eng.policy_step(None)
        h.update(g["control"].tobytes()); h.update(g["cost"].tobytes()); h.update(g["weights"].tobytes())
    print(pol, h.hexdigest()[:16], g["control"][0])
    eng.close()
