"""Regenerate tests/golden/golden_v1.npz: seeded inputs and the outputs of the CPU oracle (oracle/mpopis_oracle.c) for the
hot path, at sizes the oracle finishes in milliseconds.

These vectors are ORACLE-generated (the reference is Julia and cannot run in the build image, see the header of
oracle/mpopis_oracle.h: parity unpinned).  They serve two purposes: (1) `-m "not gpu"` tests detect any drift of the oracle
itself, (2) `-m gpu` tests check the HIP engine against committed numbers without calling the oracle.  When a Julia
toolchain is available, tools/gen_golden.jl dumps the corresponding vectors of the real reference (tests/golden/julia_*.json)
against which the oracle can then be pinned.

    python tests/golden/make_golden.py        # rewrites golden_v1.npz next to this script
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle  # noqa: E402

POLICIES = ["mppi", "gmppi", "imppi", "cemppi", "cmamppi", "muaismppi", "musigmaaismppi", "pmcmppi"]
K, T, N = 32, 5, 3


def main():
    oracle.build()
    track = oracle.load_track()
    out = {}
    rng = np.random.default_rng(20240001)
    # --- RNG definition shared by oracle and engine (Philox4x32-10 + Box-Muller)
    out["philox_normals"] = oracle.philox_normals(20240001, 3, 1, 16)
    i0, u = oracle.philox_resample_draws(20240001, 3, 1 | 0x80000000, 16)
    out["philox_res_i0"], out["philox_res_u"] = i0, u
    # --- car_racing.jl _step! and reward / within_track
    p = oracle.car_default_params()
    S0 = np.array([[0, 0, np.pi / 2, 10, 0, 0, 0, 0], [30.0, 50.0, 1.2, 18.0, -0.7, 0.3, 0.1, 0.5], [100.0, -20.0, -2.0, 6.0, 1.5, -0.8, -0.2, -1.0],
                   [5.0, 5.0, 0.3, 0.4, 0.1, 0.0, 0.05, -1.0], [200.0, 100.0, 3.0, 25.0, 2.0, 1.0, 0.3, 1.0]], dtype=np.float64)
    A = np.array([[0.0, 0.0], [1.0, 1.0], [-1.0, -0.5], [0.3, -1.0], [-0.7, 0.2]])
    out["car_s0"], out["car_a"] = S0, A
    out["car_s1"] = np.stack([oracle.car_step(p, s, a) for s, a in zip(S0, A)])
    pos = np.stack([np.array([track[0][i % 48], track[1][i % 48]]) + rng.uniform(-25, 25, 2) for i in range(24)])
    wt = [oracle.within_track(track, q) for q in pos]
    out["wt_pos"], out["wt_within"], out["wt_dist"] = pos, np.array([w for w, _ in wt], dtype=np.int32), np.array([d for _, d in wt])
    # --- one policy call per policy symbol, injected noise, two consecutive MPC steps
    for kind in POLICIES:
        env = oracle.OracleEnv("car", 1, track=track)
        st = env.state; st[3] = 14.0; st[1] = 2.0; env.state = st
        pol = oracle.OraclePolicy(kind, env, K, T, lam=10.0, alpha=1.0, U0=[0.0, 0.0], cov=[0.0625, 0.1], N=N, lam_ais=20.0,
                                  elite_threshold=0.8, cma_sigma=0.75)
        Neff = 1 if kind in ("mppi", "gmppi") else N
        for step in range(2):
            Z = rng.standard_normal((T, K, 2)) if kind == "mppi" else rng.standard_normal((Neff, K, 2 * T))
            di = rng.integers(0, K, (max(Neff - 1, 1), K)).astype(np.int32)
            du = rng.random((max(Neff - 1, 1), K))
            r = pol(env, Z, di, du)
            assert r["status"] == 0, (kind, r["status"])
            pre = "pol_%s_%d_" % (kind, step)
            out[pre + "Z"], out[pre + "di"], out[pre + "du"] = Z, di, du
            out[pre + "control"], out[pre + "cost"], out[pre + "weights"] = r["control"], r["cost"], r["weights"]
            out[pre + "iters"], out[pre + "U"] = np.array([r["iters_run"]]), pol.U.copy()
            env.step(r["control"])
            out[pre + "x"], out[pre + "reward"] = env.state, np.array([env.reward()])
    # --- closed loop with the device-reproducible Philox streams
    env = oracle.OracleEnv("car", 1, track=track)
    pol = oracle.OraclePolicy("cemppi", env, K, 8, lam=10.0, U0=[0.0, 0.0], cov=[0.0625, 0.1], N=2, elite_threshold=0.8)
    r = pol.run_trial(env, 12, num_steps=10, laps=2, log_actions=True)
    out["trial_rec"] = np.array([r["rew"], r["steps"], r["mean_v"], r["max_v"], r["mean_beta"], r["max_beta"], r["rollouts"]])
    out["trial_actions"], out["trial_x"] = r["actions"], env.state
    # --- the two scalar-action envs (RL.jl dynamics restated from memory: unpinned)
    for name, x0 in (("mountaincar", [-0.5, 0.0]), ("cartpole", [0.02, 0.0, -0.03, 0.05])):
        env = oracle.OracleEnv(name); env.state = x0
        pol = oracle.OraclePolicy("cemppi", env, 20, 15, lam=0.1, U0=[0.0], cov=[1.5], N=5, elite_threshold=0.8)
        Z = rng.standard_normal((5, 20, 15))
        r = pol(env, Z)
        out["%s_Z" % name], out["%s_control" % name], out["%s_cost" % name] = Z, r["control"], r["cost"]
        out["%s_iters" % name] = np.array([r["iters_run"]])
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    print("wrote", os.path.join(HERE, "golden_v1.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
