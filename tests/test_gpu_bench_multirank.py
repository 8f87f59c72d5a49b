"""bench.py's N > 1 control flow, exercised on ONE GPU: two ranks over gloo, both on cuda:0 (development flags of bench.py).
RCCL refuses two ranks on one device, so this run also covers the guarded fall-back from the ABI gather
(mpopis_gather_summary) to torch.distributed's gather; on a real multi-GPU node the ABI path is taken.  Checks the contract
fields of the JSON line, the weak/strong pair and that both ranks' trials are accounted for."""
import json
import os
import socket
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _records(out):
    """(compact, full): the ONE stdout line the driver parses, and the full record bench.py writes to stderr (BENCH_DETAIL ...) and gpurun_out/."""
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len([l for l in lines if l.startswith("{")]) == 1, out.stdout[-2000:]      # rank 0 prints exactly one JSON line
    assert lines[-1].startswith("{") and len(lines[-1].encode()) < 6000               # ... it is the LAST line and fits what the driver keeps (round 5: 21.9 KB, parsed null)
    det = [l for l in out.stderr.splitlines() if l.startswith("BENCH_DETAIL ")]
    assert len(det) == 1
    return json.loads(lines[-1]), json.loads(det[0][len("BENCH_DETAIL "):])


def test_bench_two_ranks_on_one_gpu():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--same-gpu", "--no-cpu-baseline", "--repeats", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    c, d = _records(out)
    assert c["n_gpus"] == 2 and c["scaling"] == "weak" and c["value"] == pytest.approx(d["value"], rel=1e-6) and c["strong_scaling"]["trials_per_gpu"] == 32
    assert c["summary_gather_path"] == d["summary_gather"] and c["rccl_ranks_seen"] is None and c["roofline"]["frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-4)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "rollouts/s"
    assert d["config"]["trials_per_gpu"] == 64 and d["dtype"] == "f64" and d["vs_baseline"] is None
    # whole-job value: both ranks' rollouts over the max time
    per_step = 2 * 64 * 10 * 4096
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    s = d["strong_scaling"]
    assert s["total_trials"] == 64 and s["trials_per_gpu"] == 32 and s["value"] > 0
    assert d["summary_gather"].startswith("torch.distributed gather") and d["rccl_required"] is False      # gloo development backend: not strict by default
    r = d["roofline"]
    # roofline.frac: the one-stream isolation pass (three regions of 2 steps: one launch per AIS iteration, all 64 trials in it)
    assert r["rollouts_per_launch"] == 64 * 4096 and r["launches"] == 3 * 2 * 10 and 0 < r["frac"] < 1 and "one-stream pass" in r["kernel_isolation"]
    # the timed region's schedule: four part-chains for this shape, a launch covers a quarter of the trials
    ds = r["default_schedule"]
    assert ds["parts"] == 4 and ds["rollout_launches"] == 4 * 2 * 10 and ds["rollouts_per_launch"] == 16 * 4096
    assert 0 < r["frac_default_schedule"] < r["frac"] and r["one_stream"]["ms_per_step"] > 0
    assert "128 independent trials IN TOTAL" in d["config"]["workload"] and d["config"]["total_trials"] == 128
    assert s["efficiency_vs_one_gpu"] > 0 and s["one_gpu_ms_per_step"] > 0
    assert "N = 1 only" in d["n1_only"] and "cpu_baseline" not in d and "configs" not in d


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (how the driver starts it): bench.py spawns its two ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "2",
           "--backend", "gloo", "--same-gpu", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    c, d = _records(out)
    assert c["repeats"]["n"] == 3
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["strong_scaling"]["trials_per_gpu"] == 32
    assert d["repeats"]["n"] == 3 and d["repeats"]["ms_per_step"]["min"] <= d["repeats"]["ms_per_step"]["median"] <= d["repeats"]["ms_per_step"]["max"]
    assert d["summary_gather"].startswith("torch.distributed gather") or d["summary_gather"].startswith("mpopis_gather_summary")


def test_bench_single_gpu_line_has_every_baseline_config():
    """N = 1 line: contract fields, repeats statistics, roofline extras and the C2/C3/C4 block (quick variant: one trial each)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--repeats", "2", "--no-cpu-baseline", "--quick-configs"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    c, d = _records(out)
    assert c["roofline"]["bound"] == "hbm" and 0 < c["roofline"]["frac"] < 1 and c["roofline"]["avg_launch_us"] > 0 and c["config"]["trials_per_gpu"] == 64
    assert c["midlap"]["closed_loop_ms"] > 0 and c["midlap"]["frozen_ms"] > 0 and [x["c"] for x in c["configs"]] == ["C2", "C3", "C4"]
    assert [x["trials"] for x in c["c4"]] == [1] and 0 < c["c4"][0]["frac"] < 1 and c["detail"] == "gpurun_out/bench_detail_latest.json"
    assert json.load(open(os.path.join(ROOT, c["detail"])))["value"] == d["value"]
    assert d["n_gpus"] == 1 and d["unit"] == "rollouts/s" and d["dtype"] == "f64"
    r = d["roofline"]
    assert 0 < r["step_frac"] < r["frac"] < 1
    # PMC-derived fields: numbers only while profiles/pmc_rollout.json carries the sha of this tree's rollout sources, else null + the reason
    if "matches this tree" in r["pmc_source"]:
        assert 0 < r["kernel_traffic_frac"] < 1 and 0 < r["fp64_executed_frac"] < 1 and r["issue_rate"]["frac_of_attainable"] > 0.5
    else:
        assert "STALE" in r["pmc_source"] or "missing" in r["pmc_source"]
        assert r["traffic"] is None and r["kernel_traffic_frac"] is None and r["fp64_executed_frac"] is None and r["issue_rate"] is None and r["valu_busy_frac"] is None
    names = [c["config"][:2] for c in d["configs"]]
    assert names == ["C2", "C3", "C4"]
    for c in d["configs"]:
        assert c["ms_per_step"] > 0 and c["rollouts_per_s"] > 0 and c["dominant"]["avg_launch_us"] > 0
        assert abs(sum(c["kernel_ms_per_step"].values()) - c["ms_per_step"]) < 0.5 * c["ms_per_step"]      # kernel classes account for the step
        # the reference's call pattern: one synchronous pol(env) per MPC step through the C ABI, one wait instead of four
        assert 0 < c["abi_sync_ms_per_step"] < 3 * c["ms_per_step"] + 0.2 and c["abi_sync"]["closed_loop"]["steps"] >= 5
        assert c["abi_sync"]["closed_loop"]["median"] <= 1.05 * c["abi_four_call_ms_per_step"]      # same loop, one wait instead of four
        assert c["resident_closed_loop_ms_per_step"] > 0
        assert c["rollout_roofline"]["schedule"] == "one stream" and 0 < c["rollout_roofline"]["frac"] < 1


def test_bench_strict_rccl_fails_instead_of_degrading():
    """N > 1 with RCCL required (the default on the nccl backend; forced here on the gloo development backend): two ranks on ONE device cannot form an
    RCCL communicator, so the run must FAIL (exit 3, reason on stderr, no JSON line) rather than report a number whose gather never went
    through csrc/engine_comm.hip.  With --allow-gather-fallback the same command degrades and labels the line."""
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo", "--same-gpu", "--no-cpu-baseline", "--repeats", "0", "--require-rccl"]
    out = subprocess.run(base + ["--master-port", str(_free_port())] + tail, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode != 0
    assert "RCCL behind the C ABI is required" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    out = subprocess.run(base + ["--master-port", str(_free_port())] + tail + ["--allow-gather-fallback"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    c, d = _records(out)
    assert c["summary_gather_path"].startswith("torch.distributed gather") and c["rccl_ranks_seen"] is None and d["rccl_required"] is False
