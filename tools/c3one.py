"""C3 (:cemppi K=150 H=50 N=10 :ss) at one trial, frozen state: ms per step (dev tool).  usage: python tools/c3one.py [policy] [K] [trials]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpopis_amd.engine import Engine
pol = sys.argv[1] if len(sys.argv) > 1 else "cemppi"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 150
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kw = dict(elite_threshold=0.8, sigma_est="ss") if pol == "cemppi" else {}
eng = Engine("car", 1, pol, K, 50, batch=B, lam=10.0, ais_its=10, lam_ais=20.0, cov=[0.0625, 0.1], seed=20240000, **kw)
eng.bench_policy_steps(10)
best = min(eng.bench_policy_steps(20)[0] / 20 for _ in range(5))
print("%s K=%d B=%d: %.4f ms per step" % (pol, K, B, best))
eng.close()
