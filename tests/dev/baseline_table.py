"""Fills the result table of BASELINE.md section 5: for C2..C5 the CPU restatement (oracle, OpenMP over k like Threads.@threads) and the HIP
engine on the SAME box and the same device-RNG noise stream: rollouts/s, MPC steps/s, roofline fractions by BASELINE section 3's accounting,
worst per-call deviation (control / cost) of the engine from the CPU result (engine state and U re-synchronised to the CPU loop every step).  Test infrastructure (uses the oracle): lives under tests/.
usage (on the GPU box): python tests/dev/baseline_table.py [threads]   -> markdown rows on stdout"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from mpopis_amd.engine import Engine

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 32
track = O.load_track()
SEED = 20240000
CFGS = [("C2 `:gmppi` K=1024 H=50 1-car", "gmppi", 1, 1024, 1, 1, 40),
        ("C3 `:cemppi` K=150 H=50 N=10", "cemppi", 1, 150, 10, 1, 10),
        ("C4 `:cmamppi` K=4096 H=50 3-car (N=10)", "cmamppi", 3, 4096, 10, 1, 2),
        ("C5 `:μΣaismppi` K=4096 H=50 N=10 ×64 trials", "musigmaaismppi", 1, 4096, 10, 64, 1)]
print("| config | backend | GPUs | rollouts/s | MPC steps/s | HBM frac | FP64 frac | max rel. err vs CPU (control / cost) | host cores |")
print("|---|---|---|---|---|---|---|---|---|")
for name, kind, ncars, K, N, B, cpu_steps in CFGS:
    T = 50
    cs = 2 * ncars * T
    cov = np.tile([0.0625, 0.1], ncars)
    Neff = 1 if kind == "gmppi" else N
    B_alg = 8 * ((4 if kind in ("musigmaaismppi", "cemppi", "cmamppi") else 3) * cs + 4)
    F_alg = 3.5e5 * ncars
    # ---- CPU: one trial, device-RNG noise stream fed in (the oracle has no RNG of its own) --------------------------------------
    env = O.OracleEnv("car", ncars, track=track)
    pol = O.OraclePolicy(kind, env, K, T, lam=10.0, U0=np.zeros(2 * ncars), cov=cov, N=N, lam_ais=20.0, elite_threshold=0.8, cma_sigma=0.75, nthreads=threads)
    eng1 = Engine("car", ncars, kind, K, T, batch=1, lam=10.0, ais_its=N, lam_ais=20.0, elite_threshold=0.8, cma_sigma=0.75, cov=cov, track=track, seed=SEED)
    cerr = kerr = 0.0
    t_cpu = 0.0
    n_roll_cpu = 0
    for step in range(cpu_steps):
        Z = np.stack([O.philox_normals(SEED + 1, step, n, cs * K).reshape(K, cs) for n in range(Neff)])
        t0 = time.perf_counter()
        ref = pol(env, Z)
        t_cpu += time.perf_counter() - t0
        got = eng1.policy_step(None)
        n_roll_cpu += ref["iters_run"] * K
        cerr = max(cerr, float(np.max(np.abs(got["control"][0] - ref["control"]) / np.maximum(1e-3, np.abs(ref["control"])))))
        rel = np.abs(got["cost"][0] - ref["cost"]) / (np.abs(ref["cost"]) + 1e-9)
        kerr = max(kerr, float(np.sort(rel)[-max(1, K // 500) - 1]))            # beyond the few standstill-chatter rollouts
        env.step(ref["control"])
        eng1.set_state(env.state[None]); eng1.set_U(pol.U[None])                 # per-call parity: re-synchronise (closed loops are sensitive maps)
    eng1.close()
    r_cpu = n_roll_cpu / t_cpu
    print("| %s | cpu (C restatement, OpenMP) | – | %.3g | %.3g | %.2g %% | %.2g %% | – | %d threads |" % (
        name, r_cpu, cpu_steps / t_cpu, 100 * r_cpu * B_alg / 8.0e12, 100 * r_cpu * F_alg / 78.6e12, threads))
    # ---- HIP: B resident trials, device RNG, closed loop stays on the device -------------------------------------------------------
    eng = Engine("car", ncars, kind, K, T, batch=B, lam=10.0, ais_its=N, lam_ais=20.0, elite_threshold=0.8, cma_sigma=0.75, cov=cov, track=track, seed=SEED)
    if kind == "cmamppi":                                                         # synthetic repeated steps drive CMA's Σ indefinite: time the closed loop
        t0 = time.perf_counter(); rec = eng.run_trials(num_steps=10, laps=2); dt = time.perf_counter() - t0
        steps, rolls = 11, float(rec[:, 14].sum())
    else:
        eng.bench_policy_steps(3)
        ms, rolls = eng.bench_policy_steps(20)
        dt, steps = ms * 1e-3, 20
    eng.close()
    r_gpu = rolls / dt
    print("| | hip | 1 | %.3g | %.3g | %.2g %% | %.2g %% | %.1e / %.1e | |" % (r_gpu, B * steps / dt, 100 * r_gpu * B_alg / 8.0e12, 100 * r_gpu * F_alg / 78.6e12, cerr, kerr))
