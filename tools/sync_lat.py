"""Latency of the synchronous pol(env) call through the C ABI (one trial): distribution of the wall time of mpopis_policy_call in a tight
loop and in the harness loop (env_step + get_state between calls), for the wait modes of MPOPIS_CALL_WAIT (0 runtime wait, 1 mailbox spin).
usage (GPU box): python tools/sync_lat.py [gmppi|cemppi|...] [K] [N]"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def run(policy, K, N, steps=200):
    from mpopis_amd.engine import Engine
    kw = dict(sigma_est="ss", elite_threshold=0.8) if policy == "cemppi" else {}
    eng = Engine("car", 1, policy, K, 50, batch=1, lam=10.0, alpha=1.0, ais_its=N, lam_ais=20.0, cov=[0.0625, 0.1], seed=20240000, **kw)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    x, t, done = eng.get_state()
    U = np.zeros((1, 100)); ctl = np.zeros((1, 2)); rew = np.zeros(1)
    a = (x.ctypes.data_as(dp), t.ctypes.data_as(ip), done.ctypes.data_as(ip), U.ctypes.data_as(dp), None, ctl.ctypes.data_as(dp), None, None, None)
    L, h = eng.L, eng._h
    out = {}
    for mode in ("tight", "harness"):
        ts = []
        for s in range(steps + 10):
            t0 = time.perf_counter()
            L.mpopis_policy_call(h, *a)
            dt = time.perf_counter() - t0
            if s >= 10:
                ts.append(dt * 1e3)
            if mode == "harness":
                L.mpopis_env_step(h, a[5], rew.ctypes.data_as(dp)); L.mpopis_get_state(h, a[0], a[1], a[2])
        ts.sort()
        out[mode] = "min %.4f p10 %.4f median %.4f p90 %.4f max %.4f" % (ts[0], ts[len(ts) // 10], ts[len(ts) // 2], ts[9 * len(ts) // 10], ts[-1])
    ms, _ = eng.bench_policy_steps(50)
    out["resident_ms_per_step"] = ms / 50
    eng.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        print(os.environ.get("MPOPIS_CALL_WAIT", "default"), sys.argv[2], run(sys.argv[2], int(sys.argv[3]), int(sys.argv[4])))
    else:
        pol, K, N = (sys.argv[1:4] + ["gmppi", "1024", "1"][len(sys.argv) - 1:])[:3]
        for w in ("0", "1"):
            subprocess.call([sys.executable, os.path.abspath(__file__), "--one", pol, str(K), str(N)], env=dict(os.environ, MPOPIS_CALL_WAIT=w))
