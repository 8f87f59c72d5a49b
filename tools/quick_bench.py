"""Quick GPU look at per-config step times and kernel-class times (not the contract bench; see bench.py)."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpopis_amd import build; build.build()
from mpopis_amd.engine import Engine

def run(policy, K, T, B, ncars=1, N=10, steps=5):
    eng = Engine("car", ncars, policy, K, T, batch=B, lam=10.0, ais_its=N, lam_ais=20.0, cma_sigma=0.75,
                 cov=np.tile([0.0625, 0.1], ncars), seed=20240000)
    eng.bench_policy_steps(2)
    eng.timing_enable(True); eng.timing_reset()
    ms, rl = eng.bench_policy_steps(steps)
    tm = eng.timing_read()
    eng.timing_enable(False)
    ms2, rl2 = eng.bench_policy_steps(steps)
    t0 = time.perf_counter(); out = eng.policy_step(minimal=True); t1 = time.perf_counter()
    print(f"{policy} K={K} T={T} B={B} cars={ncars} N={eng.N}: {ms2/steps:.3f} ms/step  rollouts/s={rl2/(ms2*1e-3):.3e}  "
          f"mpc_steps/s={B*steps/(ms2*1e-3):.1f}  sync L2 call {1e3*(t1-t0):.3f} ms  iters={out['iters_run'][:4]}")
    print("   ", {k: (round(v[0]/max(v[1],1)*1e3,1), v[1]) for k, v in tm.items() if v[1]}, "(us avg, launches)")
    eng.close()

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "c2"): run("gmppi", 1024, 50, 1)
    if which in ("all", "c3"): run("cemppi", 150, 50, 1); run("cemppi", 150, 50, 64)
    if which in ("all", "c4"): run("cmamppi", 4096, 50, 1, ncars=3); run("cmamppi", 4096, 50, 8, ncars=3)
    if which in ("all", "c5"): run("musigmaaismppi", 4096, 50, 8); run("musigmaaismppi", 4096, 50, 64)
    if which in ("all", "pmc"): run("pmcmppi", 4096, 50, 8)
    if which in ("all", "mc3"): run("musigmaaismppi", 4096, 50, 32, ncars=3, N=4)          # multi-car rollout kernel at a chip-filling batch
    if which in ("all", "mu"): run("muaismppi", 4096, 50, 64)
