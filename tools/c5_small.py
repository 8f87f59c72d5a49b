import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
from mpopis_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = Engine("car", 1, "musigmaaismppi", 4096, 50, batch=B, lam=10.0, ais_its=10, lam_ais=20.0, cov=[0.0625, 0.1], seed=20240000)
eng.bench_policy_steps(3)
ms, rl = eng.bench_policy_steps(10)
print("C5 B=%d %.3f ms/step" % (B, ms / 10))
eng.close()
