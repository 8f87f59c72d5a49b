"""The two-wave rollout kernels of the few-waves regime (k_rollout_car_duo, k_rollout_cars_duo for 2..4 cars: one wave integrates the dynamics, its partner wave evaluates the reward
through an LDS mailbox) does the same arithmetic in the same order as the one-wave kernel: costs and controls agree bit for bit, on the default track
(tables in LDS), on a 960-point track (ring only in LDS), with ragged K and with several trials."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = os.path.join(ROOT, "tests", "helpers", "duo_case.py")


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, CASE], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_two_wave_rollout_is_bit_identical_to_the_one_wave_kernel():
    one = _run({"MPOPIS_ROLLOUT_DUO": "0"})
    duo = _run({"MPOPIS_ROLLOUT_DUO": "1000000"})
    auto = _run({})
    assert set(one) == set(duo) == set(auto)
    for name in one:
        assert duo[name] == one[name], name
        assert auto[name] == one[name], name
