"""Randomised engine-vs-oracle parity sweep (test infrastructure: lives under tests/ because it uses the oracle; a fixed-seed slice of it runs
in the GPU suite as tests/test_gpu_fuzz_slice.py).  Random policy / cars / K / T / N / B, random start states, injected or device noise, random
multi-stream split.  usage: python tests/dev/fuzz_parity.py <n_cases> <seed>      (FUZZ_KINDS=pmcmppi,cmamppi restricts the policies;
FUZZ_BIGK=1: K beyond the one-workgroup sort / alias kernels -- 7169 ... 20000 -- at short horizons, the policies that sort or resample;
FUZZ_WILD=1: lambda, alpha, lambda_ais, the elite threshold and the CMA step size drawn off their defaults)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from mpopis_amd.engine import Engine
from mpopis_amd._lib import MPOPISError
from tests.helpers.fuzz_case import run_case

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
track = O.load_track()
kinds = ["gmppi", "imppi", "muaismppi", "musigmaaismppi", "cemppi", "pmcmppi", "cmamppi", "mppi"]
if os.environ.get("FUZZ_KINDS"):
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
    if os.environ.get("FUZZ_BIGK"):
        kind = str(rng.choice(["cemppi", "pmcmppi", "cmamppi", "pmcmppi", "cemppi", "musigmaaismppi"]))
        K = int(rng.choice([7169, 8192, 8193, 9001, 12289, 16384, 20000]))
        T, B, ncars = int(rng.choice([1, 3, 5, 8])), int(rng.integers(1, 4)), int(rng.choice([1, 1, 2, 3]))
        N = int(rng.integers(2, 4))
    split = int(rng.choice([0, 2, 3, 4]))
    est = str(rng.choice(["mle", "ss", "lw", "rblw", "oas"])) if kind == "cemppi" else "mle"
    device_rng = bool(rng.integers(0, 2)) and kind != "mppi"
    if kind == "cmamppi" and 2 * ncars * T * round(0.2 * K) < K:
        continue
    c = dict(kind=kind, ncars=ncars, K=K, T=T, N=N, B=B, split=split, est=est, device_rng=device_rng, seed=int(rng.integers(1, 2 ** 31)))
    if os.environ.get("FUZZ_WILD"):                             # the policies' own parameters off their defaults
        c.update(lam=float(10.0 ** rng.uniform(-1, 3)), alpha=float(rng.choice([1.0, 1.0, 0.9, 0.5, 0.0])), lam_ais=float(10.0 ** rng.uniform(0, 3)),
                 elite=float(rng.choice([0.5, 0.8, 0.9, 0.95])), cma_sigma=float(rng.choice([0.5, 0.75, 1.0, 1.5])))
        if kind == "cmamppi" and 2 * ncars * T * round((1 - c["elite"]) * K) < K:
            continue
    st, msgs = run_case(O, Engine, MPOPISError, track, c, rng)
    for m in msgs:
        print(m)
    bad += st == "fail"
    if st == "fail" and os.environ.get("FUZZ_DUMP"):
        print("stopped at case index %d (dump: %s)" % (case, os.environ["FUZZ_DUMP"]))
        break
print("%d cases, %d failed, %.0fs" % (ncases, bad, time.time() - t0))
