"""Exploration for the 'nominal trajectory brakes to a stop' parity case: x0 with Vx = 0.5 m/s, U = full brake.
usage (GPU box): python tests/dev/brake_to_stop.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from mpopis_amd.engine import Engine

track = O.load_track()
for kind, K, N in (("gmppi", 1024, 1), ("musigmaaismppi", 1024, 4)):
    for vx0, pedal in ((0.5, -1.0), (0.5, -0.3), (2.0, -1.0)):
        T = 50; cs = 2 * T
        env = O.OracleEnv("car", 1, track=track)
        x0 = env.state.copy(); x0[3] = vx0
        env.state = x0
        U0 = np.tile([0.0, pedal], T)
        pol = O.OraclePolicy(kind, env, K, T, lam=10.0, U0=np.zeros(2), cov=[0.0625, 0.1], N=N, lam_ais=20.0, nthreads=8)
        pol.U = U0
        eng = Engine("car", 1, kind, K, T, batch=1, lam=10.0, ais_its=N, lam_ais=20.0, cov=[0.0625, 0.1], track=track, seed=5)
        eng.set_state(x0[None]); eng.set_U(U0[None])
        rng = np.random.default_rng(11)
        Z = rng.standard_normal((1, N, K, cs))
        got = eng.policy_step(Z, want_E=True)
        ref = pol(env, Z[0])
        _, traj = pol.simulate_model(U0 if N == 1 else pol.U, ref["E"], log=True) if False else pol.simulate_model(U0, ref["E"], log=True)
        vx = np.abs(traj.reshape(K, T, 1, 8)[:, :, :, 3]).min(axis=(1, 2))
        rel = np.abs(got["cost"][0] - ref["cost"]) / (np.abs(ref["cost"]) + 1e-9)
        print("%-15s Vx0=%.1f pedal=%.1f: stalled (min|Vx|<1e-3) %4d of %d | cost rel err: max %.2e, >1e-7: %d, >1e-5: %d | control err %.2e | U err %.2e | w err %.2e" % (
            kind, vx0, pedal, int((vx < 1e-3).sum()), K, rel.max(), int((rel > 1e-7).sum()), int((rel > 1e-5).sum()),
            float(np.max(np.abs(got["control"][0] - ref["control"]))), float(np.max(np.abs(eng.get_U()[0] - pol.U))),
            float(np.max(np.abs(got["weights"][0] - ref["weights"])))))
        eng.close()
