"""dev: is the bench's first timed region slower than the ones behind it?  Same bracketing as bench.py (sync, perf_counter, bench_policy_steps(10), sync), 12 regions after the same pre-warm."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpopis_amd.engine import Engine
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 100
eng = Engine("car", 1, "μΣaismppi", 4096, 50, batch=64, lam=10.0, alpha=1.0, ais_its=10, lam_ais=20.0, cov=[0.0625, 0.1], seed=20240000)
if len(sys.argv) > 2: torch.cuda.synchronize(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()      # torch's lazy CUDA init out of the way first
eng.timing_enable(2); eng.timing_enable(False)
eng.bench_policy_steps(pre); eng.bench_policy_steps(2)
out = []
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.bench_policy_steps(10); torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 100)
print("prewarm %d:" % pre, " ".join("%.3f" % v for v in out))
eng.close()
