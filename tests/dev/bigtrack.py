import sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/mpopis_amd") else os.getcwd())
import numpy as np
from mpopis_amd.engine import Engine
for P in (48, 200, 300, 600, 1200, 2048):
    a = np.linspace(0, 2*np.pi, P, endpoint=False)
    tx, ty, tw = 200*np.cos(a), 200*np.sin(a) - 200, np.full(P, 15.0)
    try:
        eng = Engine("car", 1, "gmppi", 256, 20, batch=2, lam=10.0, cov=[0.0625, 0.1], track=(tx, ty, tw), seed=1)
        x = eng.get_state()[0]; x[:, 0] = 0.0; x[:, 1] = 0.0; eng.set_state(x)
        got = eng.policy_step(None)
        print(P, "ok", got["control"][0], float(got["cost"].min()))
        eng.close()
    except Exception as e:
        print(P, "FAILED", str(e)[:150])
