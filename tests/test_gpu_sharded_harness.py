"""SURVEY 8(e) end to end on hardware (one GPU, two ranks over gloo): trial k -> rank (k-1) mod 2, each rank keeps its trials
{1,3,5,7} / {2,4,6} in ONE handle (mpopis_seed_slots), one gather of the summary records; the result must equal the unsharded
7-trial run record for record.  (On a multi-GPU node with the nccl backend the same call goes through mpopis_gather_summary.)"""
import os
import socket
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_sharded_simulate_car_racing_equals_unsharded(tmp_path):
    from mpopis_amd import build
    build.build()
    from mpopis_amd.examples import simulate_car_racing
    full, _ = simulate_car_racing(num_trials=7, num_steps=12, policy_type=":μΣaismppi", num_samples=256, horizon=20, ais_its=3,
                                  seed=4321, quiet=True)
    out = str(tmp_path / "sharded.npy")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "helpers", "sharded_harness_worker.py"), out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    sh = np.load(out)
    assert sh.shape == full.shape and list(sh[:, 0]) == [1, 2, 3, 4, 5, 6, 7]
    assert np.array_equal(sh[:, :16], full[:, :16])            # trial id + the 15 record fields (the last column is wall time)
