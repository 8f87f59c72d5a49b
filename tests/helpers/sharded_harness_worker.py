"""Worker for tests/test_gpu_sharded_harness.py: one rank of a 2-rank (gloo) run of simulate_car_racing, both ranks on cuda:0.
Rank 0 writes the gathered records to the path given as argv[1]."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch.distributed as dist  # noqa: E402
from mpopis_amd.examples import simulate_car_racing  # noqa: E402

dist.init_process_group("gloo")
rec, summ = simulate_car_racing(num_trials=7, num_steps=12, policy_type=":μΣaismppi", num_samples=256, horizon=20, ais_its=3,
                                seed=4321, quiet=True, dist=dist, device=0)
if dist.get_rank() == 0:
    np.save(sys.argv[1], rec)
else:
    assert rec is None and summ is None
dist.barrier()
dist.destroy_process_group()
