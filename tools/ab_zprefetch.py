"""Same-box A/B of the Z prefetch (MPOPIS_ZPREFETCH=2 (forced on) / 0 in subprocesses): step time of the bench workload and bit-identity of the results."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from mpopis_amd.engine import Engine
kind, ncars, K, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
eng = Engine("car", ncars, kind, K, 50, batch=B, lam=10.0, ais_its=10, lam_ais=20.0, elite_threshold=0.8, cma_sigma=0.75, cov=np.tile([0.0625, 0.1], ncars), seed=20240000)
got = eng.policy_step(None, want_E=True)
sig = float(np.sum(got["control"])) , float(np.sum(got["cost"])), float(np.sum(got["E"][0]))
eng.bench_policy_steps(3)
ms, rl = eng.bench_policy_steps(20)
print(json.dumps({"ms_per_step": ms / 20, "rollouts_per_s": rl / (ms * 1e-3), "sig": [repr(x) for x in sig]}))
''' % ROOT
cases = [("musigmaaismppi", 1, 4096, 64), ("musigmaaismppi", 1, 4096, 8), ("cemppi", 1, 150, 1), ("pmcmppi", 1, 4096, 8), ("musigmaaismppi", 3, 4096, 8)]
for c in cases:
    res = []
    for z in ("2", "0", "2", "0"):
        r = subprocess.run([sys.executable, "-c", WORKER] + [str(x) for x in c], capture_output=True, text=True, env=dict(os.environ, MPOPIS_ZPREFETCH=z))
        if r.returncode != 0:
            res.append({"err": r.stderr[-300:]}); continue
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    print(c, " | ".join("%s: %s" % (z, ("%.3f ms" % d["ms_per_step"]) if "ms_per_step" in d else d["err"]) for z, d in zip(("on", "off", "on", "off"), res)),
          "identical results:", all("sig" in d for d in res) and res[0]["sig"] == res[1]["sig"])
