"""Helper process for tests/test_gpu_rollout_duo.py: one-car and multi-car handles in the few-waves regime, a few policy steps each; prints one JSON line with the
cost vectors and controls as hex (bit patterns).  MPOPIS_ROLLOUT_DUO is read once per process by the library, hence the subprocess."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from mpopis_amd.engine import Engine

out = {}
for name, pol, K, B, track, ncars in (("gmppi", "gmppi", 1024, 1, None, 1), ("ragged", "cemppi", 150, 3, None, 1), ("mu", "musigmaaismppi", 1000, 2, None, 1),
                                     ("bigtrack", "imppi", 200, 2, 960, 1), ("cars3", "cmamppi", 700, 2, None, 3), ("cars2", "gmppi", 333, 1, None, 2),
                                     ("cars4", "musigmaaismppi", 256, 2, None, 4)):
    kw = dict(batch=B, lam=10.0, ais_its=3, cov=np.tile([0.0625, 0.1], ncars), seed=4242)
    if pol == "cmamppi":
        kw.update(elite_threshold=0.8, cma_sigma=0.75)
    eng = Engine("car", ncars, pol, K, 50, **kw)
    if track:
        th = np.linspace(0, 2 * np.pi, track, endpoint=False)
        r = 40.0 + 6.0 * np.sin(3 * th)
        mid = np.stack([r * np.cos(th), r * np.sin(th)], 1)
        nrm = mid / np.linalg.norm(mid, axis=1, keepdims=True)
        eng.set_track(mid[:, 0], mid[:, 1], np.full(track, 4.0))
        x0 = np.zeros((B, 8)); x0[:, 0] = mid[0, 0]; x0[:, 1] = mid[0, 1]; x0[:, 2] = np.pi / 2; x0[:, 3] = 5.0
        eng.set_state(x0)
    rec = []
    for _ in range(3):
        got = eng.policy_step(None)
        rec.append(got["control"].tobytes().hex() + got["cost"].tobytes().hex() + got["weights"].tobytes().hex())
    out[name] = rec
    eng.close()
print(json.dumps(out))
