"""dev: hashes of everything a few small runs return -- policies x 1 / 3 cars, logged trajectories, a closed loop -- for bit-identity checks between two
builds (tools/ab/run_py.sh tools/hash_ab.py) or two settings of an environment knob"""
import sys, os, hashlib; sys.path.insert(0, os.getcwd())
import numpy as np
from mpopis_amd.engine import Engine
h_all = hashlib.sha256()
for pol, K, B, nc, log in (("musigmaaismppi", 4096, 3, 1, False), ("cemppi", 1000, 2, 1, False), ("pmcmppi", 777, 2, 1, False), ("muaismppi", 2048, 2, 1, False),
                           ("gmppi", 1024, 1, 1, False), ("gmppi", 300, 2, 1, True), ("cmamppi", 512, 2, 3, False), ("gmppi", 256, 2, 2, True), ("imppi", 4096, 40, 1, False)):
    kw = dict(batch=B, lam=10.0, ais_its=4, cov=np.tile([0.0625, 0.1], nc), seed=99, log_trajectories=log)
    if pol == "cmamppi": kw.update(elite_threshold=0.8, cma_sigma=0.75)
    eng = Engine("car", nc, pol, K, 50, **kw)
    h = hashlib.sha256()
    for _ in range(2):
        g = eng.policy_step(None)
        h.update(g["control"].tobytes()); h.update(g["cost"].tobytes()); h.update(g["weights"].tobytes())
        if log: h.update(eng.get_trajectories().tobytes())
    rec = eng.run_trials(6, 2)
    h.update(np.ascontiguousarray(rec).tobytes())
    print(pol, K, B, nc, log, h.hexdigest()[:16])
    h_all.update(h.digest())
    eng.close()
print("ALL", h_all.hexdigest()[:16])
