"""Worker of test_pmcmppi_alias_*: one :pmcmppi policy step with tie-heavy injected noise; writes the resampled indices, weights and control
to an .npz.  Run once with the default alias-table construction (parallel + certified fall-back) and once with MPOPIS_ALIAS_PAR=0
(sequential kernel only): the test compares the two files."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpopis_amd.engine import Engine

K, groups, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
T, N, B = 6, 3, 2
cs = 2 * T
rng = np.random.default_rng(K + groups)
if groups > 0:                                                   # a few distinct noise columns -> a few distinct weights in equal-sized groups
    proto = 0.6 * rng.standard_normal((B, N, groups, cs))
    Z = np.repeat(proto, (K + groups - 1) // groups, axis=2)[:, :, :K, :].copy()
    if groups == 3:
        Z = Z[:, :, rng.permutation(K), :]
else:
    Z = rng.standard_normal((B, N, K, cs))                       # generic weights: every decision certified, no fall-back
di = rng.integers(0, K, (B, N - 1, K)).astype(np.int32)
du = rng.random((B, N - 1, K))
eng = Engine("car", 1, "pmcmppi", K, T, batch=B, lam=10.0, ais_its=N, lam_ais=20.0, cov=[0.0625, 0.1])
got = eng.policy_step(Z, di, du)
np.savez(out, res=got["res_idx0"], control=got["control"], cost=got["cost"], weights=got["weights"], iters=got["iters_run"])
eng.close()
