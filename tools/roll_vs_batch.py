"""Average launch time of the rollout kernel against the number of resident trials (tail / wave-quantisation effects).
usage (GPU box): python tools/roll_vs_batch.py <cars> B1 B2 ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpopis_amd.engine import Engine
cars = int(sys.argv[1])
for B in [int(x) for x in sys.argv[2:]]:
    eng = Engine("car", cars, "musigmaaismppi", 4096, 50, batch=B, lam=10.0, ais_its=3, lam_ais=20.0, cov=np.tile([0.0625, 0.1], cars), seed=20240000)
    eng.bench_policy_steps(3)
    eng.timing_enable(2); eng.timing_reset()
    eng.bench_policy_steps(6)
    tm = eng.timing_read()
    us = tm["rollout"][0] / tm["rollout"][1] * 1e3
    S = 64 // cars
    waves = B * ((4096 + S - 1) // S)
    print("cars=%d B=%3d: rollout %.1f us per launch, %.2f us per trial, %d waves = %.2f x 4096" % (cars, B, us, us / B, waves, waves / 4096.0), flush=True)
    eng.close()
