"""The stdout line of bench.py must stay short: the round driver keeps the last 8 KB of stdout and parses the LAST line; round 5's 21.9 KB line
was cut and recorded as `parsed: null`.  bench.compact_line() is a pure function of the full result dict, so it is checked here without a GPU:
built from a canned dict with EVERY optional block present (N = 1 blocks and the N > 1 strong-scaling block together, long prose everywhere)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PROSE = "lorem ipsum " * 80          # ~1 KB of text wherever the full record carries an explanation


def _agree():
    return {"control": 4.1e-14, "cost": 6.6e-15, "cost_own_samples_excl_largest": 1.2e-14, "iters_equal": True, "steps": 2, "costs_off_by_more_than_1e-5": 0, "what": PROSE}


def _row(name, trials):
    return {"config": name, "trials": trials, "steps": 10, "ms_per_step": 22.861234567, "rollouts_per_s": 114712345.678, "mpc_steps_per_s": 2799.1234, "loop": PROSE,
            "kernel_ms_per_step": {k: 1.2345678 for k in ("rollout", "sample", "potrf", "reweight", "moments", "select")}, "dominant": {"class": "rollout", "avg_launch_us": 1400.123, "share_of_kernel_time": 0.6},
            "kernel_time_over_step_time": 1.3, "schedule": PROSE, "alg_bytes_per_rollout": 9632, "hbm_frac": 0.138123, "fp64_reference_algorithm_frac": 1.5,
            "rollout_roofline": {"kernel": "k_rollout_cars<3,...>", "avg_launch_us": 1401.5, "launches": 40, "rollouts_per_launch": 262144.0, "alg_bytes_per_rollout": 9632, "frac": 0.2251234, "schedule": "one stream"},
            "abi_sync_ms_per_step": 4.134, "abi_sync": {"closed_loop": {"median": 4.1, "steps": 12}}, "abi_four_call_ms_per_step": 4.2, "what": PROSE,
            "max_rel_err_vs_cpu": _agree(), "cpu_one_trial": {"rollouts_per_s": 8889.09, "mpc_steps_per_s": 0.217, "steps": 1, "seconds": 4.6, "threads": 32}}


def canned():
    return {
        "metric": "trajectory rollouts/sec (+ MPC steps/sec), Car-Racing K=4096 H=50", "value": 499612345.678, "unit": "rollouts/s", "n_gpus": 8, "steps": 20, "warmup": 5,
        "ms_per_step": 5.2471234, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "mpc_steps_per_s": 12197.123,
        "config": {"workload": "Car-Racing 1-car :μΣaismppi K=4096 H=50 N=10 λ=10 λ_ais=20, 512 independent trials IN TOTAL = 64 per GPU x 8 GPUs (weak scaling of BASELINE configs[4]; " + PROSE, "total_trials": 512, "trials_per_gpu": 64, "rollouts_per_step": 2621440, "prewarm_steps": 100, "parallelism": "trials sharded x8, RCCL gather of summary stats"},
        "repeats": {"n": 101, "what": PROSE, "ms_per_step": {"median": 5.25, "min": 5.21, "max": 5.4}, "value": {"median": 5e8, "max": 5.1e8, "min": 4.9e8}},
        "roofline": {"bound": "hbm", "limiter": "fp64_valu_issue", "kernel": "k_rollout_car<1, 4, false, true>", "achieved": 2406.71234, "peak": 8000.0, "unit": "GB/s", "frac": 0.30083, "traffic": 223912345.0,
                     "pmc_source": "profiles/pmc_rollout.json matches this tree (" + PROSE + ")", "kernel_isolation": PROSE, "frac_default_schedule": 0.2012,
                     "default_schedule": {"parts": 4, "rollout_avg_launch_us": 131.2, "rollout_launches": 800, "rollouts_per_launch": 65536.0, "what": PROSE},
                     "one_stream": {"ms_per_step": 5.7012, "value": 4.6e8}, "frac_definition": PROSE, "what_binds": PROSE, "step_frac": 0.2019, "kernel_traffic_frac": 0.0795, "fp64_executed_frac": 0.5071,
                     "avg_launch_us": 352.04, "launches": 600, "rollouts_per_launch": 262144.0, "alg_bytes_per_rollout": 3232, "schedule": PROSE, "bound_note": PROSE, "valu_busy_frac": 0.8312,
                     "issue_rate": {"valu_wave_insts_per_rollout": 41600.0, "frac_of_attainable": 0.9961, "frac_of_datasheet": 0.797, "source": PROSE},
                     "fp64_reference_algorithm_tflops": 260.1, "fp64_peak_tflops": 78.6, "fp64_reference_algorithm_frac": 3.31, "note": PROSE},
        "kernel_ms_per_step": {k: 1.23456789 for k in ("rollout", "sample", "potrf", "reweight", "moments", "select", "finalize")},
        "kernel_ms_per_step_schedule": PROSE, "summary_gather": "mpopis_gather_summary (RCCL behind the C ABI)", "rccl_ranks_seen": 8, "rccl_required": True,
        "midlap_states": {"max_rel_err_vs_cpu": {"slots": [0, 21, 42, 63], "control": 1.92e-9, "oracle_vs_itself_control": 3.86e-7, "cost_all_rollouts": 1.8e-13, "costs_off_by_more_than_1e-5": 0,
                                                 "iters_equal": True, "chatter_share": 0.0486, "what": PROSE},
                          "closed_loop": {"what": PROSE, "ms_per_step": 5.6494, "value": 4.64e8},
                          "frozen_at_step_100": {"what": PROSE, "ms_per_step": 6.5778, "value": 3.985e8, "rollout_avg_launch_us": 341.47}},
        "strong_scaling": {"scaling": "strong", "total_trials": 64, "trials_per_gpu": 8, "value": 1.2e9, "unit": "rollouts/s", "ms_per_step": 2.1, "mpc_steps_per_s": 30476.0,
                           "one_gpu_ms_per_step": 5.25, "speedup_vs_one_gpu": 2.5, "efficiency_vs_one_gpu": 0.3125, "note": PROSE},
        "configs": [_row("C2 Car-Racing 1-car :gmppi K=1024 H=50", t) for t in (1, 64)] + [_row("C3 Car-Racing 1-car :cemppi K=150 H=50 N=10 elite=0.8 Σ_est=:ss", t) for t in (1, 64)]
                   + [_row("C4 Car-Racing 3-car :cmamppi K=4096 H=50 N=10", t) for t in (1, 8, 32, 64)],
        "cpu_baseline": {"value": 327123.4, "unit": "rollouts/s", "cores": 32, "kind": "port", "value_1thread": 18100.5, "sample_1thread": PROSE, "sample": PROSE, "mpc_steps_per_s": 7.98,
                         "thread_calibration_rollouts_per_s": {"8": 150000, "16": 270000, "32": 327000, "64": 250000}},
        "max_rel_err_vs_cpu": _agree(),
        "n1_only": PROSE,
    }


def test_compact_line_is_short_and_complete():
    full = canned()
    assert len(json.dumps(full).encode()) > 30000                     # the full record is of the size that broke round 5
    line = bench.compact_line(full, "gpurun_out/bench_detail_latest.json")
    assert "\n" not in line and len(line.encode()) <= bench.COMPACT_LIMIT < 6000
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == full["value"] or abs(d["value"] - full["value"]) < 1e-8 * full["value"]
    assert d["config"]["trials_per_gpu"] == 64 and d["config"]["rollouts_per_step"] == 2621440 and len(d["config"]["workload"]) <= 160
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    for k in ("kernel", "traffic", "avg_launch_us", "launches", "rollouts_per_launch", "alg_bytes_per_rollout", "frac_default_schedule", "step_frac", "valu_busy_frac"):
        assert r[k] is not None, k
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 32 and cb["value"] > 0 and cb["value_1thread"] > 0 and cb["unit"] == "rollouts/s" and len(cb["sample"]) <= 120
    assert d["max_rel_err_vs_cpu"]["iters_equal"] is True and d["max_rel_err_vs_cpu"]["control"] < 1e-5
    assert d["midlap"]["closed_loop_ms"] > 0 and d["midlap"]["frozen_ms"] > 0 and d["midlap"]["control_err"] > 0
    assert [c["trials"] for c in d["c4"]] == [1, 64] and all(0 < c["frac"] < 1 and c["avg_launch_us"] > 0 for c in d["c4"])
    assert d["summary_gather_path"].startswith("mpopis_gather_summary") and d["rccl_ranks_seen"] == 8
    assert d["strong_scaling"]["trials_per_gpu"] == 8
    assert line.count("lorem") <= 12                                  # only the two bounded strings (workload <= 160, sample <= 120 characters) carry text


def test_compact_line_of_a_minimal_record():
    """flags off (--no-cpu-baseline --no-configs --no-midlap --repeats 0): the optional blocks are simply absent"""
    full = {k: v for k, v in canned().items() if k not in ("configs", "cpu_baseline", "max_rel_err_vs_cpu", "midlap_states", "strong_scaling", "repeats")}
    full["midlap_states"] = None
    d = json.loads(bench.compact_line(full))
    assert "cpu_baseline" not in d and "c4" not in d and "midlap" not in d and d["roofline"]["frac"] > 0


def test_compact_line_sheds_blocks_rather_than_growing():
    full = canned()
    full["configs"] = full["configs"] * 12                            # a future block that would overflow
    line = bench.compact_line(full)
    assert len(line.encode()) <= bench.COMPACT_LIMIT
    d = json.loads(line)
    assert "roofline" in d and "cpu_baseline" in d and "configs" not in d
