"""dev: path counters of the rollout kernel (needs tools/ab/libstats.so, built by tools/path_stats.sh, copied over the library):
share of sub-steps in which some lane of the wave took the general sub-step, share of reward evaluations through the general search,
at the reset state and after n closed-loop MPC steps.   usage (GPU box): python tools/path_stats.py [trials] [policy] [K] [N] [cars]"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from mpopis_amd.engine import Engine
from mpopis_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pol = sys.argv[2] if len(sys.argv) > 2 else "μΣaismppi"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
N = int(sys.argv[4]) if len(sys.argv) > 4 else 10
cars = int(sys.argv[5]) if len(sys.argv) > 5 else 1
kw = dict(sigma_est="ss", elite_threshold=0.8) if pol == "cemppi" else (dict(elite_threshold=0.8, cma_sigma=0.75) if pol == "cmamppi" else {})
eng = Engine("car", cars, pol, K, 50, batch=B, lam=10.0, alpha=1.0, ais_its=N, lam_ais=20.0, cov=np.tile([0.0625, 0.1], cars), seed=20240000, **kw)
L = _lib.lib()
buf = (C.c_ulonglong * 8)()


def point(tag):
    L.mpopis_debug_path_stats(buf, 1)
    ms, _ = eng.bench_policy_steps(5)
    L.mpopis_debug_path_stats(buf, 1)
    s = [int(v) for v in buf]
    print("%-28s %.3f ms/step | sub-steps %d, with a general lane %.4f (lanes per such wave %.1f) | rewards %d, general search %.4f" % (
        tag, ms / 5, s[0], s[1] / max(1, s[0]), s[2] / max(1, s[1]), s[3], s[4] / max(1, s[3])), flush=True)


point("reset state")
done = 0
for n in (40, 60, 100):
    eng.run_trials(num_steps=n - 1, laps=4)
    done += n
    point("after %d closed-loop steps" % done)
eng.close()
