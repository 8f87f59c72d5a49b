"""Closed-loop :cmamppi (3 cars, K=4096): when the device reports PosDefException, does the oracle fail at the same MPC step?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from mpopis_amd.engine import Engine
from mpopis_amd._lib import MPOPISError
track = O.load_track()
seed, ncars, K, T, N = 11, 3, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 50, 10
cov = np.tile([0.0625, 0.1], ncars)
fail = None
for k in range(1, 9):
    eng = Engine("car", ncars, "cmamppi", K, T, batch=1, lam=10.0, ais_its=N, elite_threshold=0.8, cma_sigma=0.75, cov=cov, track=track, seed=seed)
    eng.seed_slots([seed + k])
    try:
        rec, act = eng.run_trials(200, 2, log_actions=True)
        print("trial", k, "device ok, steps", rec[0, 1])
    except MPOPISError as e:
        # find the failing step: re-run step by step
        eng.close()
        eng = Engine("car", ncars, "cmamppi", K, T, batch=1, lam=10.0, ais_its=N, elite_threshold=0.8, cma_sigma=0.75, cov=cov, track=track, seed=seed)
        eng.seed_slots([seed + k])
        acts = []
        for s in range(201):
            try:
                got = eng.policy_step(None, minimal=True)
            except MPOPISError as e2:
                print("trial", k, "device fails at MPC step", s, "code", e2.code)
                fail = (k, s, np.array(acts))
                break
            acts.append(got["control"][0].copy())
            eng.env_step(got["control"])
        eng.close()
        break
    eng.close()
if fail:
    k, s, acts = fail
    env = O.OracleEnv("car", ncars, track=track)
    pol = O.OraclePolicy("cmamppi", env, K, T, lam=10.0, U0=np.zeros(2 * ncars), cov=cov, N=N, elite_threshold=0.8, cma_sigma=0.75, nthreads=8)
    t = time.time()
    r = pol.run_trial(env, seed + k, num_steps=200, laps=2, log_actions=True)
    print("oracle: status", r["status"], "steps", r["steps"], "in %.0fs" % (time.time() - t))
    n = min(len(acts), int(r["steps"]) + 1)
    if n:
        print("max |action diff| over the first %d steps: %.3e" % (n, np.abs(acts[:n] - r["actions"][:n]).max()))
