"""rollout-kernel timing of the bench workload on one stream (for same-box A/B of two builds, tools/ab/run_py.sh)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpopis_amd.engine import Engine
eng = Engine("car", 1, "musigmaaismppi", 4096, 50, batch=64, lam=10.0, ais_its=10, lam_ais=20.0, cov=[0.0625, 0.1], seed=20240000)
eng.set_overlap(1)
eng.bench_policy_steps(3)
eng.timing_enable(2); eng.timing_reset()
ms, rl = eng.bench_policy_steps(20)
tm = eng.timing_read()
one = "single stream: %.3f ms/step, rollout kernel %.1f us avg over %d launches" % (ms / 20, tm["rollout"][0] / tm["rollout"][1] * 1e3, tm["rollout"][1])
eng.set_overlap(-1); eng.timing_enable(False)
eng.bench_policy_steps(3)
ms, rl = eng.bench_policy_steps(20)
print(one + " | default schedule: %.3f ms/step  %.3e rollouts/s" % (ms / 20, rl / (ms * 1e-3)))
eng.close()
