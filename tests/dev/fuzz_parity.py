"""Randomised engine-vs-oracle parity sweep (test infrastructure: lives under tests/ because it uses the oracle; the committed tests hold fixed cases).  Random policy / cars / K / T / N / B,
random start states, injected or device noise, random multi-stream split.  usage: python tests/dev/fuzz_parity.py <n_cases> <seed>
A few costs per case may differ beyond 1e-7 without being a bug: rollouts that brake to a standstill chatter; the sweep
allows ncars*K/200 of them per slot as long as control and U agree to 1e-6 (the committed tests identify those rollouts from the oracle's
own trajectory instead)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from mpopis_amd.engine import Engine
from mpopis_amd._lib import MPOPISError

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
track = O.load_track()
kinds = ["gmppi", "imppi", "muaismppi", "musigmaaismppi", "cemppi", "pmcmppi", "cmamppi", "mppi"]
if os.environ.get("FUZZ_KINDS"):                                   # e.g. FUZZ_KINDS=pmcmppi,cmamppi
    kinds = os.environ["FUZZ_KINDS"].split(",")
bad = 0
t0 = time.time()
for case in range(ncases):
    kind = kinds[int(rng.integers(0, len(kinds)))]
    ncars = int(rng.choice([1, 1, 2, 3, 4]))
    T = int(rng.choice([1, 3, 7, 10, 16, 25, 50]))
    K = int(rng.choice([1, 2, 17, 64, 65, 150, 256, 700, 1024]))
    N = 1 if kind in ("gmppi", "mppi") else int(rng.integers(2, 5))
    B = int(rng.integers(1, 7))
    split = int(rng.choice([0, 2, 3, 4]))
    est = str(rng.choice(["mle", "ss", "lw", "rblw", "oas"])) if kind == "cemppi" else "mle"
    device_rng = bool(rng.integers(0, 2)) and kind != "mppi"
    cs = 2 * ncars * T
    if kind == "cmamppi" and cs * round(0.2 * K) < K:
        continue
    seed = int(rng.integers(1, 2 ** 31))
    cov = np.tile([0.0625, 0.1], ncars)
    tag = "%s cars=%d K=%d T=%d N=%d B=%d split=%d est=%s rng=%s" % (kind, ncars, K, T, N, B, split, est, "dev" if device_rng else "inj")
    try:
        eng = Engine("car", ncars, kind, K, T, batch=B, lam=10.0, ais_its=N, lam_ais=20.0, elite_threshold=0.8, sigma_est=est, cma_sigma=0.75,
                     cov=cov, track=track, seed=seed)
    except MPOPISError as e:
        print("create refused:", tag, e); continue
    eng.set_overlap(split)
    envs, pols = [], []
    for b in range(B):
        e = O.OracleEnv("car", ncars, track=track)
        for _ in range(int(rng.integers(0, 30))):
            e.step(np.clip(np.tile([0.05, 0.5], ncars) + 0.2 * rng.standard_normal(2 * ncars), -1, 1))
        envs.append(e)
        pols.append(O.OraclePolicy(kind, e, K, T, lam=10.0, U0=np.zeros(2 * ncars), cov=cov, N=N, lam_ais=20.0, elite_threshold=0.8,
                                   sigma_est=est, cma_sigma=0.75, nthreads=8))
    eng.set_state(np.stack([e.state for e in envs]))
    ok = True
    for step in range(2):
        if kind == "mppi":
            Z = rng.standard_normal((B, T, K, 2 * ncars))
        elif device_rng:
            Z = np.stack([np.stack([O.philox_normals(seed + b + 1, step, n, cs * K).reshape(K, cs) for n in range(N)]) for b in range(B)])
        else:
            Z = rng.standard_normal((B, N, K, cs))
        if device_rng:
            dd = [[O.philox_resample_draws(seed + b + 1, step, n | 0x80000000, K) for n in range(max(N - 1, 1))] for b in range(B)]
            di = np.array([[d[0] for d in row] for row in dd], dtype=np.int32); du = np.array([[d[1] for d in row] for row in dd])
        else:
            di = rng.integers(0, K, (B, max(N - 1, 1), K)).astype(np.int32); du = rng.random((B, max(N - 1, 1), K))
        refs = [pols[b](envs[b], Z[b], di[b], du[b]) for b in range(B)]
        worst = min(r["status"] for r in refs)
        try:
            got = eng.policy_step(None if device_rng else Z, None if device_rng else di, None if device_rng else du)
        except MPOPISError as e:
            if e.code != worst:
                print("FAIL status", tag, "step", step, "engine", e.code, "oracle", worst); ok = False
            break
        if worst:
            print("FAIL status", tag, "engine ok, oracle", worst); ok = False; break
        U = eng.get_U()
        for b in range(B):
            r = refs[b]
            rel = np.abs(got["cost"][b] - r["cost"]) / (np.abs(r["cost"]) + 1e-9)
            nbad = int((rel > 1e-7).sum())
            ea = float(np.abs(got["control"][b] - r["control"]).max()); eu = float(np.abs(U[b] - pols[b].U).max())
            if got["iters_run"][b] != r["iters_run"] or nbad > max(2, ncars * K // 200) or ea > 1e-6 or eu > 1e-6:
                print("FAIL", tag, "step", step, "slot", b, "iters", got["iters_run"][b], r["iters_run"], "cost-bad", nbad, "max rel %.2e" % rel.max(), "ctrl %.2e U %.2e" % (ea, eu))
                ok = False
        if not ok:
            break
    bad += (not ok)
    eng.close()
print("%d cases, %d failed, %.0fs" % (ncases, bad, time.time() - t0))
