"""Device side of the closed-loop comparison: log actions / iterations / states of trial `seed+1` step by step -> gpurun_out/c4_dev.npz"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from mpopis_amd.engine import Engine, default_track
from mpopis_amd._lib import MPOPISError
seed, ncars, K, T, N = 11, 3, 4096, 50, 10
cov = np.tile([0.0625, 0.1], ncars)
eng = Engine("car", ncars, "cmamppi", K, T, batch=1, lam=10.0, ais_its=N, elite_threshold=0.8, cma_sigma=0.75, cov=cov, seed=seed)
eng.seed_slots([seed + 1])
acts, its, states, mincost = [], [], [], []
for s in range(60):
    states.append(eng.get_state()[0][0].copy())
    try:
        got = eng.policy_step(None)
    except MPOPISError as e:
        print("device fails at step", s, e.code); break
    acts.append(got["control"][0].copy()); its.append(int(got["iters_run"][0])); mincost.append(float(got["cost"][0].min()))
    eng.env_step(got["control"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "c4_dev.npz"), acts=np.array(acts), its=np.array(its), states=np.array(states), mincost=np.array(mincost))
print("saved", len(acts), "steps; iters", its)
