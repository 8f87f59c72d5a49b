import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpopis_amd import engine as eng_mod
from oracle import oracle
track = oracle.load_track()
for ncars, T in [(1, 10), (3, 50)]:
    cs = 2 * ncars * T
    K, N = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 3
    cov = np.tile([0.0625, 0.1], ncars)
    d = np.tile([0.0625, 0.1], cs // 2) * 10.0 ** (-17.0 * np.arange(cs) / (cs - 1.0))
    S = np.diag(d)
    eng = eng_mod.Engine("car", ncars, "cmamppi", K, T, batch=2, lam=10.0, ais_its=N, cma_sigma=0.75, cov=cov, track=track, seed=5)
    eng.set_Sigma(S)
    Z = np.random.default_rng(3).standard_normal((2, N, K, cs))
    try:
        got = eng.policy_step(Z); code = 0
    except Exception as e:
        code = getattr(e, "code", -99); print("engine error", e)
    U = eng.get_U(); Sg = eng.get_Sigma()
    for b in range(2):
        env = oracle.OracleEnv("car", ncars, track=track)
        pol = oracle.OraclePolicy("cmamppi", env, K, T, lam=10.0, U0=np.zeros(2 * ncars), cov=cov, N=N, cma_sigma=0.75, nthreads=8)
        pol.Sigma = S
        r = pol(env, Z[b])
        print(ncars, T, "slot", b, "oracle status", r["status"], "iters", r["iters_run"], "engine code", code)
        if code == 0 and r["status"] == 0:
            print("  iters", got["iters_run"][b], "control err", np.max(np.abs(got["control"][b] - r["control"])), "U err", np.max(np.abs(U[b] - pol.U)))
            D = np.abs(Sg[b] - r["Sigma_last"]); i = np.unravel_index(np.argmax(D), D.shape)
            print("  Sigma max abs err", D.max(), "at", i, "dev", Sg[b][i], "ref", r["Sigma_last"][i], "max|ref|", np.abs(r["Sigma_last"]).max(), "offdiag dev/ref", Sg[b][0, 1], r["Sigma_last"][0, 1])
    eng.close()
