"""dev: one line per build (MPOPIS_HIP_LIB selects it; tools/ab/run_py.sh alternates two builds on one box): the headline workload at the reset
state (step + rollout launch, median of 7 x 10 steps), over 100 closed-loop MPC steps, frozen at the state reached there, and a hash of
everything the closed loop returned (bit-identity between builds).   usage: python tools/ab_closed_loop.py [trials] [cars] [policy]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpopis_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cars = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pol = sys.argv[3] if len(sys.argv) > 3 else "musigmaaismppi"
kw = dict(elite_threshold=0.8, cma_sigma=0.75) if pol == "cmamppi" else {}
eng = Engine("car", cars, pol, 4096, 50, batch=B, lam=10.0, ais_its=10, lam_ais=20.0, cov=np.tile([0.0625, 0.1], cars), seed=20240000, **kw)
eng.bench_policy_steps(30)
def med(n=7):
    eng.timing_enable(2)
    out = []
    for _ in range(n):
        eng.timing_reset()
        ms, _ = eng.bench_policy_steps(10)
        tm = eng.timing_read()
        out.append((ms / 10, tm["rollout"][0] / max(1, tm["rollout"][1]) * 1e3))
    eng.timing_enable(False)
    out.sort()
    return out[len(out) // 2]
r_ms, r_us = med()
h = hashlib.sha256()
steps = 100 if pol != "cmamppi" else 12
t0 = time.perf_counter(); rec, act = eng.run_trials(num_steps=steps - 1, laps=4, log_actions=True); cl = (time.perf_counter() - t0) * 1e3 / steps
h.update(rec[:, :15].tobytes()); h.update(act.tobytes()); h.update(eng.get_state()[0].tobytes()); h.update(eng.get_U().tobytes())
eng.bench_policy_steps(2)
f_ms, f_us = med(5)
got = eng.policy_step(None)
h.update(got["cost"].tobytes()); h.update(got["control"].tobytes())
print("reset %.3f ms (rollout %.1f us) | closed loop %.3f ms/step | frozen@%d %.3f ms (rollout %.1f us) | hash %s" % (r_ms, r_us, cl, steps, f_ms, f_us, h.hexdigest()[:12]), flush=True)
eng.close()
