"""Worker of tests/test_gpu_fullsize.py::test_part_chain_schedule_survives_another_library_initialising_the_gpu_first: torch touches the device BEFORE the
engine creates its streams (what a host process with other GPU libraries looks like), then the headline shape is timed on the default schedule and on one
stream.  Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
from mpopis_amd.engine import Engine
eng = Engine("car", 1, "μΣaismppi", 4096, 50, batch=64, lam=10.0, alpha=1.0, ais_its=10, lam_ais=20.0, cov=[0.0625, 0.1], seed=20240000)
res = {}
for ov in (0, 1):
    eng.set_overlap(ov); eng.bench_policy_steps(20)
    res[ov] = sorted(eng.bench_policy_steps(10)[0] / 10 for _ in range(5))[2]
eng.close()
print(json.dumps({"default_ms": res[0], "one_stream_ms": res[1]}))
