"""dev: does the part-chain schedule survive another library initialising the GPU first?  usage: python tools/stream_order.py [torch_first|engine_first] [nccl]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "engine_first"
import torch
def torch_init():
    torch.zeros(1, device="cuda"); torch.cuda.synchronize()
    if "nccl" in sys.argv:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        t = torch.ones(1, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
if mode == "torch_first": torch_init()
from mpopis_amd.engine import Engine
eng = Engine("car", 1, "μΣaismppi", 4096, 50, batch=64, lam=10.0, alpha=1.0, ais_its=10, lam_ais=20.0, cov=[0.0625, 0.1], seed=20240000)
if mode == "engine_first": torch_init()
res = {}
for ov in (0, 1, 2, 3, 4):
    eng.set_overlap(ov); eng.bench_policy_steps(20)
    res[ov] = sorted(eng.bench_policy_steps(10)[0] / 10 for _ in range(5))[2]
print(mode, " ".join(a for a in sys.argv[2:]), "ms/step by set_overlap(0=auto,1,2,3,4):", " ".join("%.3f" % res[o] for o in (0, 1, 2, 3, 4)))
eng.close()
