"""Quick GPU look at kernel times (not the contract bench; see bench.py)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
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
    print(f"{policy} K={K} T={T} B={B} cars={ncars} N={eng.N}: {ms2/steps:.3f} ms/step (timed {ms/steps:.3f}) "
          f"rollouts/s={rl2/(ms2*1e-3):.3e}  mpc_steps/s={B*steps/(ms2*1e-3):.1f}")
    print("   ", {k: (round(v[0]/max(v[1],1)*1e3,1), v[1]) for k, v in tm.items() if v[1]}, "(us avg, launches)")
    eng.close()

if __name__ == "__main__":
    for B in (1, 8, 64):
        run("gmppi", 4096, 50, B)
    run("gmppi", 1024, 50, 1)
    run("musigmaaismppi", 4096, 50, 8)
    run("musigmaaismppi", 4096, 50, 64)
    run("gmppi", 4096, 50, 8, ncars=3)
