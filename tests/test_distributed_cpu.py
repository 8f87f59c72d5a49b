"""N>1 path on CPU: trial sharding + the single summary-stats gather, world_size 2 over gloo."""
import os
import socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mpopis_amd.examples import shard_trials, _gather_records
    mine = shard_trials(7, rank, world)
    rec = np.array([[k] + [float(k * 10 + i) for i in range(5)] for k in mine])
    out = _gather_records(rec, dist)
    if rank == 0:
        q.put(out[np.argsort(out[:, 0])])
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    from mpopis_amd.examples import shard_trials
    assert shard_trials(7, 0, 2) == [1, 3, 5, 7] and shard_trials(7, 1, 2) == [2, 4, 6]
    assert sorted(sum((shard_trials(64, r, 8) for r in range(8)), [])) == list(range(1, 65))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # rank 0 holds 4 trials, rank 1 holds 3: uneven shards are padded inside _gather_records
    assert list(out[:, 0]) == [1, 2, 3, 4, 5, 6, 7]
    assert np.array_equal(out[:, 1], np.arange(1, 8) * 10.0)


def test_assemble_gathered_matches_the_torch_path_layout():
    """The nccl branch of simulate_car_racing (records through mpopis_gather_summary) and the gloo branch (_gather_records) must
    build the same table: trial ids from (rank, slot), record fields in place, Ex Time last."""
    from mpopis_amd.examples import assemble_gathered, shard_trials
    from mpopis_amd._lib import RECORD_LEN
    world, num_trials = 3, 8
    parts = []
    for g in range(world):
        mine = shard_trials(num_trials, g, world)
        rec = np.array([[100.0 * k + f for f in range(RECORD_LEN)] for k in mine]).reshape(-1, RECORD_LEN)
        rec[:, 15] = 7.0 + g                                   # the rank's wall time travels in the status slot
        parts.append(rec)
    out = assemble_gathered(parts, world)
    out = out[np.argsort(out[:, 0])]
    assert out.shape == (num_trials, RECORD_LEN + 2)
    assert list(out[:, 0]) == list(range(1, num_trials + 1))
    for row in out:
        k = int(row[0])
        assert np.array_equal(row[1:16], [100.0 * k + f for f in range(15)])
        assert row[16] == 0.0 and row[17] == 7.0 + (k - 1) % world
    assert assemble_gathered([np.zeros((0, RECORD_LEN))] * 2, 2).shape == (0, RECORD_LEN + 2)


def test_bench_spawns_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` as the driver starts it (no torchrun, no WORLD_SIZE): bench.py re-executes itself under torch.distributed.run
    with two ranks; --launch-check stops after the rendezvous (gloo, no GPU), rank 0 prints the only JSON line."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-check"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"launch_check": 2, "n_gpus": 2}, out.stdout[-500:]
    # under a launcher that disagrees with --gpus the bench refuses with a message, not an assert
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=120, env=env2, cwd=root)
    assert bad.returncode != 0 and "WORLD_SIZE (1) != --gpus (2)" in bad.stderr
