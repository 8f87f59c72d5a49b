import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpopis_amd import build; build.build()
from mpopis_amd.engine import Engine
from oracle import oracle as O
track = O.load_track()
nc, K, T = 3, 4096, 50
cs = 2 * nc * T
for N in (2, 3, 5):
    env = O.OracleEnv('car', nc, track=track)
    pol = O.OraclePolicy('cmamppi', env, K, T, lam=10.0, U0=np.zeros(2 * nc), cov=np.tile([0.0625, 0.1], nc), N=N, cma_sigma=0.75, nthreads=32)
    Z = np.stack([O.philox_normals(20240001, 0, n, cs * K).reshape(K, cs) for n in range(N)])
    r = pol(env, Z)
    eng = Engine("car", nc, "cmamppi", K, T, batch=1, lam=10.0, ais_its=N, cma_sigma=0.75, cov=np.tile([0.0625, 0.1], nc), seed=20240000)
    try:
        g = eng.policy_step(Z[None], want_E=True)
        print("N=%d oracle status %d iters %d | engine iters %d  cost relerr %.3e  E abs err %.3e control err %.3e" % (
            N, r['status'], r['iters_run'], g['iters_run'][0], np.max(np.abs(g['cost'][0] - r['cost']) / np.abs(r['cost'])),
            np.max(np.abs(g['E'][0].T - r['E'])), np.max(np.abs(g['control'][0] - r['control']))))
    except Exception as e:
        print("N=%d oracle status %d iters %d | engine error: %s" % (N, r['status'], r['iters_run'], e))
    eng.close()
