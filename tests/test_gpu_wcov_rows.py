"""The row form of the scatter (moments) kernel -- kernels_mfma.hip, ROWS: whole tile rows per wave, k-permuted chunks, two LDS images -- against the
pair-list form on the same inputs.  The default rule only picks the row form from ~12 resident K = 4096 trials up (it needs >= 192 workgroups), more than an
oracle comparison can afford, so this test forces it (MPOPIS_WCOV_ROWS=2) at small batches and compares with the pair-list form (MPOPIS_WCOV_ROWS=0),
which the oracle parity tests cover.  The two cut the K range differently into partial sums, so Σ' agrees to rounding, not bit for bit; the noise of the
following iterations is drawn from a factor of Σ', so controls / costs / weights agree to the same order amplified by the rollout."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = os.path.join(ROOT, "tests", "helpers", "rows_case.py")


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, CASE], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def _arr(h):
    return np.frombuffer(bytes.fromhex(h), dtype=np.float64)


@pytest.mark.parametrize("ksplit", ["", "2", "6"])
def test_row_form_agrees_with_pair_list_form(ksplit):
    extra = {"MPOPIS_KSPLIT": ksplit} if ksplit else {}
    ref = _run(dict(extra, MPOPIS_WCOV_ROWS="0"))
    got = _run(dict(extra, MPOPIS_WCOV_ROWS="2"))
    assert set(ref) == set(got)
    for name in ref:
        for step, (r, g) in enumerate(zip(ref[name], got[name])):
            # (the first step's Σ' is a few roundings away; from the second step on every quantity carries the first step's difference through U)
            for what, rh, gh, tol in zip(("control", "cost", "weights", "Sigma"), r, g, (1e-9, 1e-9, 1e-8, 1e-11 if step == 0 else 1e-8)):
                a, b = _arr(rh), _arr(gh)
                assert a.shape == b.shape
                scale = max(1.0, float(np.max(np.abs(a)))) if what != "Sigma" else float(np.max(np.abs(a)))
                err = float(np.max(np.abs(a - b))) / scale
                assert err < tol, (name, step, what, err)


def test_row_form_is_what_runs_when_forced():
    """guards the guard: with an odd number of partials the row form is not eligible and the forced run must equal the pair-list run bit for bit"""
    ref = _run({"MPOPIS_WCOV_ROWS": "0", "MPOPIS_KSPLIT": "5"})
    got = _run({"MPOPIS_WCOV_ROWS": "2", "MPOPIS_KSPLIT": "5"})
    assert ref == got
    ref2 = _run({"MPOPIS_WCOV_ROWS": "0", "MPOPIS_KSPLIT": "4"})
    got2 = _run({"MPOPIS_WCOV_ROWS": "2", "MPOPIS_KSPLIT": "4"})
    assert ref2 != got2                                                # ... and with an even number the forced run really takes the other kernel
