"""Which rollouts of the C4 parity case deviate from the oracle, and do they touch the Vx -> 0 chatter corner?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from mpopis_amd import engine as eng_mod
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_baseline_shapes import start_states
track = O.load_track()
ncars, K, T, N, B = 3, 4096, 50, 3, 2
cs = 2 * ncars * T
cov = np.tile([0.0625, 0.1], ncars)
x0 = start_states(O, track, ncars, B)
print("x0", x0.reshape(B, ncars, 8)[:, :, :5])
eng = eng_mod.Engine("car", ncars, "cmamppi", K, T, batch=B, lam=10.0, ais_its=N, lam_ais=20.0, elite_threshold=0.8, cma_sigma=0.75, cov=cov, track=track, log_trajectories=True)
eng.set_state(x0)
rng = np.random.default_rng(1234 + K + cs)
Z = rng.standard_normal((B, N, K, cs))
got = eng.policy_step(Z, want_E=True)
traj = eng.get_trajectories()
U = eng.get_U()
for b in range(B):
    e = O.OracleEnv("car", ncars, track=track); e.state = x0[b]
    p = O.OraclePolicy("cmamppi", e, K, T, lam=10.0, U0=np.zeros(2 * ncars), cov=cov, N=N, lam_ais=20.0, elite_threshold=0.8, cma_sigma=0.75, nthreads=8)
    ref = p(e, Z[b])
    rel = np.abs(got["cost"][b] - ref["cost"]) / np.abs(ref["cost"])
    bad = np.where(rel > 1e-7)[0]
    print("slot", b, "bad rollouts", len(bad), "max rel", rel.max(), "median rel", np.median(rel))
    # last-iteration trajectories from the device logger: min |Vx| per rollout over cars
    tr = traj[b].reshape(K, T, ncars, 8)
    minvx = np.min(np.abs(tr[:, :, :, 3]), axis=(1, 2))
    for k in bad[:20]:
        print("   k=%d rel=%.2e cost=%.6g ref=%.6g min|Vx|=%.3g" % (k, rel[k], got["cost"][b][k], ref["cost"][k], minvx[k]))
    good = np.setdiff1d(np.arange(K), bad)
    print("   min|Vx| over good rollouts: min %.3g ; over bad: max %.3g" % (minvx[good].min(), minvx[bad].max() if len(bad) else -1))
    print("   control diff", np.abs(got["control"][b] - ref["control"]).max(), "weights diff", np.abs(got["weights"][b] - ref["weights"]).max())
