"""dev: stress of the wave-specialised rollout kernels (LDS mailboxes, spin waits): random shapes inside their regime, closed loops and single steps, for a
given number of seconds; any hang shows up as the caller's timeout.  usage: timeout 400 python tools/soak_duo.py 240"""
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np
from mpopis_amd.engine import Engine
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(5)
t0 = time.time(); n = 0
pols = ["gmppi", "imppi", "muaismppi", "musigmaaismppi", "cemppi", "pmcmppi", "cmamppi", "mppi"]
while time.time() - t0 < budget:
    nc = int(rng.choice([1, 1, 1, 2, 3, 4]))
    pol = pols[int(rng.integers(0, len(pols)))]
    K = int(rng.choice([1, 7, 64, 65, 150, 333, 1024, 2048, 4096])); T = int(rng.choice([1, 2, 5, 20, 50])); B = int(rng.integers(1, 9))
    if nc * K * B > 60000: K = max(1, 60000 // (nc * B))
    kw = dict(batch=B, lam=10.0, ais_its=int(rng.integers(1, 5)), cov=np.tile([0.0625, 0.1], nc), seed=int(rng.integers(1, 1 << 30)))
    if pol == "cmamppi": kw.update(elite_threshold=0.8, cma_sigma=0.75)
    if pol in ("cemppi", "cmamppi") and K < 10: K = 10
    try:
        eng = Engine("car", nc, pol, K, T, **kw)
    except Exception as e:                                      # constructor refusals the reference has too (shape rules)
        if "-4" in repr(e): raise
        continue
    try:
        for _ in range(3): eng.policy_step(None, minimal=True)
        eng.run_trials(int(rng.integers(1, 12)), 2)
    except Exception as e:
        if "-4" in repr(e): raise
    eng.close(); n += 1
print("soak_duo: %d handles in %.0f s, no hang, no HIP error" % (n, time.time() - t0))
