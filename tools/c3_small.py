import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
from mpopis_amd.engine import Engine
pol = sys.argv[1] if len(sys.argv) > 1 else "cemppi"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 150
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
eng = Engine("car", 1, pol, K, 50, batch=B, lam=10.0, ais_its=10, lam_ais=20.0, elite_threshold=0.8, sigma_est="ss", cov=[0.0625, 0.1], seed=20240000)
eng.bench_policy_steps(3)
ms, rl = eng.bench_policy_steps(10)
print("%s K=%d B=%d %.3f ms/step" % (pol, K, B, ms / 10))
eng.close()
