"""dev: path counters of the rollout kernel (needs tools/ab/libstats.so, built by tools/path_stats.sh, selected with MPOPIS_HIP_LIB=tools/ab/libstats.so):
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


nth = B * (K if cars == 1 else ((K + (64 // cars) - 1) // (64 // cars)) * 64)
sick = (C.c_ubyte * nth)()


def point(tag, n):
    """n closed-loop MPC steps (mpopis_run_trials: the state moves, pol.U follows it), counters over that window; then ONE more step for the
    per-thread flags (OR over that step's N rollouts of a thread's sample index: per-rollout share = 1 - (1 - f)^(1/N) if independent)"""
    import time
    L.mpopis_debug_path_stats(buf, 1)
    t0 = time.perf_counter(); eng.run_trials(num_steps=n - 1, laps=4); ms = (time.perf_counter() - t0) * 1e3
    L.mpopis_debug_path_stats(buf, 1)
    s = [int(v) for v in buf]
    L.mpopis_debug_sick(sick, nth, 1)
    eng.run_trials(num_steps=0, laps=4)
    L.mpopis_debug_sick(sick, nth, 1)
    f = np.frombuffer(sick, dtype=np.uint8)
    inv = lambda v: 1 - (1 - v) ** (1.0 / N)
    x = eng.get_state()[0]
    print("    rollouts that took the general sub-step %.4f, that end with Vx <= 0.5: %.4f" % (s[6] / max(1, s[0] // 500 * 64), s[7] / max(1, s[0] // 500 * 64)))
    print("%-26s %.3f ms/step | sub-steps with a general lane %.4f | rewards through the general search %.4f (five-point tier %.4f) | slot0 Vx=%.1f" % (
        tag, ms / n, s[1] / max(1, s[0]), s[4] / max(1, s[3]), s[5] / max(1, s[3]), x[0, 3]), flush=True)


if os.environ.get("FROZEN"):                                 # frozen-state policy steps after n closed-loop steps (what tools/midlap_bench.py times)
    eng.run_trials(num_steps=99, laps=4)
    for rep in range(2):
        L.mpopis_debug_path_stats(buf, 1)
        ms, _ = eng.bench_policy_steps(5)
        L.mpopis_debug_path_stats(buf, 1)
        s = [int(v) for v in buf]
        nro = s[0] // 500 * 64
        print("frozen at step 100: %.2f ms/step | sub-steps with a general lane %.4f | general search %.4f (tier 5: %.4f) | rollouts that took the general sub-step %.4f, that end with Vx <= 0.5: %.4f" % (ms / 5, s[1] / max(1, s[0]), s[4] / max(1, s[3]), s[5] / max(1, s[3]), s[6] / max(1, nro), s[7] / max(1, nro)))
    eng.close(); sys.exit(0)
done = 0
for n in ((4, 4) if pol == "cmamppi" else (10, 30, 30, 30, 50, 50)):
    point("closed-loop steps %d..%d" % (done, done + n), n)
    done += n + 1
eng.close()
