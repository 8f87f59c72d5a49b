#!/usr/bin/env python
"""bench.py -- headline benchmark of the MPOPIS rollout-and-reweight path on MI355X.

Workload = BASELINE.json configs[4], the configuration the metric is quoted on: Car-Racing 1-car
:μΣaismppi, K=4096, H=50, N=10 AIS iterations, 64 independent trials.  It fits one MI355X, so at
N=1 all 64 trials are resident on the GPU.  `value` is WEAK scaling (64 trials per GPU at every N:
trials are independent, so more GPUs = more trials/seeds, sharded with no data-path collective); for
N > 1 the same JSON line also carries `strong_scaling` = configs[4] exactly as written (64 trials in
total, 64/N per GPU), measured right after the weak pass -- at 8 trials per GPU the step is bound by
the latency of the 500-sub-step dependency chain, not by throughput.
One "step" = one MPC step of every resident trial = 64*10*4096 model rollouts + 64*10 reweightings
+ 64*9 (mu, Sigma) updates.  Metric: trajectory rollouts/s (whole job) and
MPC steps/s.  Noise comes from the device Philox streams; inputs are resident in HBM.
`value` is timed on the engine's DEFAULT schedule (for this shape: four skewed part-chains on their own streams).  `roofline.frac` needs the
dominant kernel's duration with the chip to itself, so it comes from a labelled one-stream pass (mpopis_set_overlap(h, 1)) of the same
workload right after the timed region; `roofline.frac_default_schedule` is the same formula on the timed region's own (time-shared) launches.

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K, H, N_AIS, CARS = 4096, 50, 10, 1
TRIALS_PER_GPU = 64
PREWARM_STEPS = 100         # untimed, before the W warm-up steps (see main): ~0.6 s of the same work
LAM, LAM_AIS = 10.0, 20.0
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # FP64 vector (public spec)
BYTES_PER_ROLLOUT = 8 * (4 * 2 * CARS * H + 4)      # 3232 B (Σ-adapting iterations, cs=100)
FLOPS_PER_ROLLOUT = 3.5e5 * CARS                    # SURVEY 8(d) reference-algorithm flop-equivalents


COMPACT_LIMIT = 4096           # bytes: the round driver keeps the last 8 KB of stdout; round 5's 21.9 KB line was cut and parsed as null


def _r(x, nd=6):
    """numbers rounded to nd significant digits (the full-precision values are in the detail file)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float("%.*g" % (nd, float(x)))
    except (TypeError, ValueError):
        return None


def _pick(d, keys, nd=6):
    return {k: _r(d.get(k), nd) for k in keys if d is not None and k in d} if d else None


def compact_line(out, detail_path=None):
    """The ONE stdout line of the contract, built from the full result dict `out` (which goes to the detail file and stderr): contract fields,
    `roofline`, `cpu_baseline`, the agreement column, the mid-lap figures and C4 at 1 / 64 trials -- numbers only, no prose.  Kept under
    COMPACT_LIMIT bytes whatever flags are on (tests/test_bench_line.py builds it from a canned dict; the GPU test asserts it on the real line)."""
    rf = out.get("roofline") or {}
    c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    c["value"], c["ms_per_step"] = _r(c["value"], 9), _r(c["ms_per_step"], 7)
    c["mpc_steps_per_s"] = _r(out.get("mpc_steps_per_s"), 7)
    cfg = out.get("config") or {}
    c["config"] = {"workload": str(cfg.get("workload", ""))[:160], **{k: cfg.get(k) for k in ("total_trials", "trials_per_gpu", "rollouts_per_step", "parallelism") if k in cfg}}
    rep = out.get("repeats")
    if rep:
        c["repeats"] = {"n": rep.get("n"), "ms_median": _r(rep["ms_per_step"]["median"]), "ms_min": _r(rep["ms_per_step"]["min"]), "ms_max": _r(rep["ms_per_step"]["max"])}
    c["roofline"] = {"bound": rf.get("bound"), "limiter": rf.get("limiter"), "kernel": rf.get("kernel"), "achieved": _r(rf.get("achieved")), "peak": rf.get("peak"), "unit": rf.get("unit"),
                     "frac": _r(rf.get("frac")), "traffic": _r(rf.get("traffic")),
                     **{k: _r(rf.get(k)) for k in ("avg_launch_us", "launches", "rollouts_per_launch", "alg_bytes_per_rollout", "frac_default_schedule", "step_frac",
                                                   "valu_busy_frac", "kernel_traffic_frac", "fp64_executed_frac")},
                     "one_stream_ms_per_step": _r((rf.get("one_stream") or {}).get("ms_per_step")),
                     "issue_frac_of_attainable": _r((rf.get("issue_rate") or {}).get("frac_of_attainable")),
                     "pmc": "fresh" if "matches" in str(rf.get("pmc_source")) else ("stale" if "STALE" in str(rf.get("pmc_source")) else "none")}
    kms = out.get("kernel_ms_per_step")
    if kms:
        c["kernel_ms_per_step_one_stream"] = {k: _r(v, 4) for k, v in kms.items()}
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {**_pick(cb, ("value", "unit", "cores", "kind", "value_1thread", "mpc_steps_per_s")), "sample": str(cb.get("sample_short") or cb.get("sample", ""))[:120]}
    ag = out.get("max_rel_err_vs_cpu")
    if ag:
        c["max_rel_err_vs_cpu"] = _pick(ag, ("control", "cost", "iters_equal", "steps", "costs_off_by_more_than_1e-5"), 3)
    ml = out.get("midlap_states")
    if ml:
        am = ml.get("max_rel_err_vs_cpu") or {}
        c["midlap"] = {"closed_loop_ms": _r(ml["closed_loop"]["ms_per_step"]), "closed_loop_value": _r(ml["closed_loop"]["value"]),
                       "frozen_ms": _r(ml["frozen_at_step_100"]["ms_per_step"]), "frozen_rollout_launch_us": _r(ml["frozen_at_step_100"].get("rollout_avg_launch_us")),
                       "control_err": _r(am.get("control"), 3), "oracle_self_sensitivity": _r(am.get("oracle_vs_itself_control"), 3),
                       "cost_err": _r(am.get("cost_all_rollouts"), 3), "iters_equal": am.get("iters_equal")}
    rows = out.get("configs") or []
    if rows:
        c["configs"] = [{"c": r["config"][:2], "trials": r["trials"], "ms": _r(r["ms_per_step"], 4), "rps": _r(r["rollouts_per_s"], 4),
                         **({"sync_ms": _r(r["abi_sync_ms_per_step"], 4)} if r.get("abi_sync_ms_per_step") else {}),
                         **({"err": _r(max(r["max_rel_err_vs_cpu"]["control"], r["max_rel_err_vs_cpu"]["cost"]), 2)} if r.get("max_rel_err_vs_cpu") else {}),
                         **({"cpu_rps": _r(r["cpu_one_trial"]["rollouts_per_s"], 3)} if r.get("cpu_one_trial") else {})} for r in rows]
        c["configs_keys"] = "ms=ms per MPC step, rps=rollouts/s, sync_ms=one-trial pol(env) via the C ABI, err=max rel err vs cpu (control, cost), cpu_rps=oracle one trial"
        c4 = [next(r for r in rows if r["config"].startswith("C4") and r["trials"] == t) for t in (1, 64) if any(r["config"].startswith("C4") and r["trials"] == t for r in rows)]
        c["c4"] = [{"trials": r["trials"], "ms_per_step": _r(r["ms_per_step"], 5), "rollouts_per_s": _r(r["rollouts_per_s"], 5),
                    **_pick(r.get("rollout_roofline"), ("kernel", "avg_launch_us", "rollouts_per_launch", "alg_bytes_per_rollout", "frac"), 7)} for r in c4]
    c["summary_gather_path"] = out.get("summary_gather")
    c["rccl_ranks_seen"] = out.get("rccl_ranks_seen")
    st = out.get("strong_scaling")
    if st:
        c["strong_scaling"] = _pick(st, ("total_trials", "trials_per_gpu", "value", "ms_per_step", "one_gpu_ms_per_step", "speedup_vs_one_gpu", "efficiency_vs_one_gpu"))
    if detail_path:
        c["detail"] = detail_path
    line = json.dumps(c, ensure_ascii=False, separators=(",", ":"))
    # belt and braces: whatever a future field adds, the line the driver parses stays short -- drop the optional blocks, widest first
    for k in ("configs", "configs_keys", "kernel_ms_per_step_one_stream", "repeats", "strong_scaling", "midlap", "c4"):
        if len(line.encode()) <= COMPACT_LIMIT:
            break
        c.pop(k, None)
        line = json.dumps(c, ensure_ascii=False, separators=(",", ":"))
    return line


def emit(out):
    """Full record -> gpurun_out/bench_detail_latest.json (merged back by gpurun; copied to profiles/ per round) and stderr; compact line -> stdout, last."""
    rel = os.path.join("gpurun_out", "bench_detail_latest.json")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, rel), "w") as f:
            json.dump(out, f, ensure_ascii=False, indent=1)
    except OSError:
        rel = None
    sys.stderr.write("BENCH_DETAIL " + json.dumps(out, ensure_ascii=False) + "\n")
    sys.stderr.flush()
    print(compact_line(out, rel))
    sys.stdout.flush()


def _oracle_policy(policy, cars, Kc, Nc, nthreads, **kw):
    import numpy as np
    from oracle import oracle as O
    env = O.OracleEnv("car", cars, track=O.load_track())
    kind = {"μΣaismppi": "musigmaaismppi"}.get(policy, policy)
    pol = O.OraclePolicy(kind, env, Kc, H, lam=LAM, U0=np.zeros(2 * cars), cov=np.tile([0.0625, 0.1], cars), N=Nc, lam_ais=LAM_AIS,
                         nthreads=nthreads, **kw)
    return env, pol


def _oracle_noise(cars, Kc, n_iter, step):
    import numpy as np
    from oracle import oracle as O
    cs = 2 * cars * H
    return np.stack([O.philox_normals(20240001, step, n, cs * Kc).reshape(Kc, cs) for n in range(n_iter)])


def cpu_port(policy, cars, Kc, Nc, nthreads, budget_s, max_steps=256, check_device=None, check_steps=2, **kw):
    """The oracle (C restatement, OpenMP over k like Threads.@threads :269) on this box's host cores: whole pol(env) steps of ONE trial
    of the given config (the reference's own usage), bounded by budget_s seconds / max_steps steps.
    check_device: also run the ENGINE (one resident trial, device RNG of the same Philox stream the oracle is fed) in lock step for the
    first check_steps steps and report the worst deviation of its control / per-rollout cost from the oracle's -- BASELINE.md section 5's
    "max rel. err vs CPU" column.  The oracle is the checker here; the comparison sits outside every timed bracket."""
    import numpy as np
    n_iter = 1 if policy == "gmppi" else Nc
    env, pol = _oracle_policy(policy, cars, Kc, Nc, nthreads, **kw)
    eng, agree = None, None
    if check_device is not None:
        from mpopis_amd.engine import Engine
        eng = Engine("car", cars, policy, Kc, H, batch=1, lam=LAM, alpha=1.0, ais_its=Nc, lam_ais=LAM_AIS, cov=np.tile([0.0625, 0.1], cars),
                     seed=20240000, device=check_device, **kw)
        agree = {"control": 0.0, "cost": 0.0, "cost_own_samples_excl_largest": 0.0, "iters_equal": True, "steps": 0, "costs_off_by_more_than_1e-5": 0}
    steps, t_total, rollouts = 0, 0.0, 0
    try:
        while True:
            Z = _oracle_noise(cars, Kc, n_iter, steps)
            if eng is not None and steps < check_steps:
                eng.set_U(pol.U[None])                    # per-call statement: same state, same pol.U, same draws (closed loops are sensitive maps)
                U_before = pol.U.copy()
            t0 = time.perf_counter()
            r = pol(env, Z)
            t_total += time.perf_counter() - t0
            if r["status"] != 0:
                break
            if eng is not None and steps < check_steps:
                got = eng.policy_step(None, want_E=True)
                agree["iters_equal"] = bool(agree["iters_equal"] and int(got["iters_run"][0]) == int(r["iters_run"]))
                agree["control"] = max(agree["control"], float(np.max(np.abs(got["control"][0] - r["control"]) / np.maximum(1e-3, np.abs(r["control"])))))
                # per-rollout cost on IDENTICAL samples: the engine's final noise matrix through the oracle's model, ALL K rollouts, nothing set aside
                cost_same = pol.simulate_model(U_before, np.ascontiguousarray(got["E"][0].T))
                rel_same = np.abs(got["cost"][0] - cost_same) / (np.abs(cost_same) + 1e-9)
                agree["cost"] = max(agree["cost"], float(rel_same.max()))
                agree["costs_off_by_more_than_1e-5"] += int(np.sum(rel_same > 1e-5))
                # (until round 4 this column compared cost[k] of the engine with cost[k] of the oracle, each on its OWN last-iteration samples --
                # ~1e-8 apart for the adaptive policies -- and dropped the K/500 largest deviations; kept as a side field)
                rel = np.sort(np.abs(got["cost"][0] - r["cost"]) / (np.abs(r["cost"]) + 1e-9))
                agree["cost_own_samples_excl_largest"] = max(agree["cost_own_samples_excl_largest"], float(rel[-max(1, Kc // 500) - 1]))
                agree["steps"] += 1
            steps += 1
            rollouts += int(r["iters_run"]) * Kc
            if t_total >= budget_s or steps >= max_steps:
                break
    finally:
        if eng is not None:
            eng.close()
    out = {"rollouts_per_s": rollouts / max(t_total, 1e-9), "mpc_steps_per_s": steps / max(t_total, 1e-9), "steps": steps, "seconds": t_total, "threads": nthreads}
    if agree is not None:
        agree["what"] = ("engine (1 resident trial, device Philox stream) vs the CPU oracle fed the same stream, %d pol(env) call(s), same state and pol.U per call: "
                         "control = max |dev - cpu| / max(1e-3, |cpu|); cost = max relative per-rollout deviation over ALL K rollouts of the last iteration on identical "
                         "samples (the engine's final noise matrix through the oracle's model; nothing set aside); tolerance of the north star: 1e-5" % agree["steps"])
        out["max_rel_err_vs_cpu"] = agree
    return out


def midlap_agreement(x, U, device, slots=(0, 21, 42, 63), nthreads=16):
    """Engine vs the CPU oracle in the states the mid-lap block times (closed-loop step 100, pol.U rolled on): ONE pol(env) from (x, U) of a
    few of the resident slots -- engine on a fresh handle (device Philox stream, MPC step 0), oracle fed the same stream -- like
    tests/test_gpu_midlap_parity.py.  control: max |dev - cpu|; cost: per-rollout, the engine's final samples through the oracle's model
    (identical controls on both sides), max relative deviation over ALL K rollouts -- nothing set aside -- and the count beyond 1e-5;
    chatter_share: rollouts whose oracle trajectory comes within 0.12 m/s of Vx = 0 (one sub-step of full brake: the reference's sign(Vx)
    regime, src/envs/car_racing.jl:311).  The oracle is the checker here; outside every timed bracket."""
    import numpy as np
    from oracle import oracle as O
    from mpopis_amd.engine import Engine
    slots = [b for b in slots if b < x.shape[0]]
    seed = 20250000
    cs = 2 * CARS * H
    eng = Engine("car", CARS, "μΣaismppi", K, H, batch=len(slots), lam=LAM, alpha=1.0, ais_its=N_AIS, lam_ais=LAM_AIS, cov=np.tile([0.0625, 0.1], CARS),
                 seed=seed, device=device)
    try:
        eng.set_state(x[slots]); eng.set_U(U[slots])
        got = eng.policy_step(None, want_E=True)
    finally:
        eng.close()
    out = {"slots": slots, "control": 0.0, "oracle_vs_itself_control": 0.0, "cost_all_rollouts": 0.0, "costs_off_by_more_than_1e-5": 0, "iters_equal": True, "chatter_share": 0.0}
    for i, b in enumerate(slots):
        env = O.OracleEnv("car", CARS, track=O.load_track())
        env.state = x[b]
        pol = O.OraclePolicy("musigmaaismppi", env, K, H, lam=LAM, U0=np.zeros(2 * CARS), cov=np.tile([0.0625, 0.1], CARS), N=N_AIS, lam_ais=LAM_AIS, nthreads=nthreads)
        pol.U = U[b]
        U_orig = U[b].copy()
        Z = np.stack([O.philox_normals(seed + i + 1, 0, n, cs * K).reshape(K, cs) for n in range(N_AIS)])
        ref = pol(env, Z)
        cost_same, traj = pol.simulate_model(U_orig, np.ascontiguousarray(got["E"][i].T), log=True)
        rel = np.abs(got["cost"][i] - cost_same) / (np.abs(cost_same) + 1e-9)
        out["control"] = max(out["control"], float(np.max(np.abs(got["control"][i] - ref["control"]))))
        # the calibration: the same oracle with pol.U nudged by 1e-13 relative -- how well conditioned the policy itself is at this state
        env2 = O.OracleEnv("car", CARS, track=O.load_track())
        env2.state = x[b]
        pol2 = O.OraclePolicy("musigmaaismppi", env2, K, H, lam=LAM, U0=np.zeros(2 * CARS), cov=np.tile([0.0625, 0.1], CARS), N=N_AIS, lam_ais=LAM_AIS, nthreads=nthreads)
        pol2.U = U[b] * (1.0 + 1e-13)
        out["oracle_vs_itself_control"] = max(out["oracle_vs_itself_control"], float(np.max(np.abs(pol2(env2, Z)["control"] - ref["control"]))))
        out["cost_all_rollouts"] = max(out["cost_all_rollouts"], float(rel.max()))
        out["costs_off_by_more_than_1e-5"] += int((rel > 1e-5).sum())
        out["iters_equal"] = bool(out["iters_equal"] and int(got["iters_run"][i]) == int(ref["iters_run"]))
        out["chatter_share"] = max(out["chatter_share"], float((np.abs(traj.reshape(K, H, CARS, 8)[:, :, :, 3]).min(axis=(1, 2)) < 0.12).mean()))
    out["what"] = ("engine vs CPU oracle, one pol(env) from the state / pol.U of %d resident slots after the mid-lap block (device Philox stream on both sides): control = max |dev - cpu|; "
                   "oracle_vs_itself_control = the oracle against the oracle with pol.U nudged by 1e-13 relative (the policy's own conditioning at these states); "
                   "cost = max relative per-rollout deviation over ALL K rollouts on identical samples (none set aside); tolerance of the north star: 1e-5" % len(slots))
    return out


def cpu_baseline(seconds_target=12.0, check_device=None):
    """The oracle on this box's host cores, headline workload: whole pol(env) steps of ONE trial, bounded to ~10-30 s of CPU work.  The thread
    count is calibrated first (one step each at 8..all cores): on the GPU boxes more OpenMP threads than physical cores available to the
    container make the oracle slower, and the fastest setting is the fair baseline."""
    ncpu = os.cpu_count() or 1
    Z0 = _oracle_noise(CARS, K, N_AIS, 0)
    best_t, best_n, t_cal, table = None, 1, 0.0, {}
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128, ncpu)}):
        env, pol = _oracle_policy("μΣaismppi", CARS, K, N_AIS, nt)
        t0 = time.perf_counter()
        pol(env, Z0)
        dt = time.perf_counter() - t0
        t_cal += dt
        table[str(nt)] = round(N_AIS * K / dt)          # rollouts/s of one MPC step at this thread count: the calibration, printed so the choice is reproducible
        if best_t is None or dt < best_t:
            best_t, best_n = dt, nt
        if dt > 4 * best_t:
            break
    r = cpu_port("μΣaismppi", CARS, K, N_AIS, best_n, max(2.0, seconds_target - t_cal), check_device=check_device)
    # BASELINE.md section 4 also asks for the 1-thread figure: one MPC step of the same trial on a single core
    env1, pol1 = _oracle_policy("μΣaismppi", CARS, K, N_AIS, 1)
    t0 = time.perf_counter()
    pol1(env1, Z0)
    t_one = time.perf_counter() - t0
    return {"value": r["rollouts_per_s"], "unit": "rollouts/s", "cores": best_n, "kind": "port",
            "value_1thread": N_AIS * K / t_one, "sample_1thread": "1 MPC step of 1 trial (%d rollouts), 1 thread, %.1f s" % (N_AIS * K, t_one),
            "sample": "%d MPC step(s) of 1 trial, same config (%d rollouts), C oracle + OpenMP over k, %.1f s (+%.1f s calibrating the thread count; host reports %d CPUs)"
                      % (r["steps"], r["steps"] * N_AIS * K, r["seconds"], t_cal, ncpu),
            "sample_short": "%d MPC steps of 1 trial (%d rollouts), C oracle + OpenMP over k, %.1f s; host reports %d CPUs" % (r["steps"], r["steps"] * N_AIS * K, r["seconds"], ncpu),
            "mpc_steps_per_s": r["mpc_steps_per_s"], "thread_calibration_rollouts_per_s": table, "max_rel_err_vs_cpu": r.get("max_rel_err_vs_cpu")}


def cpu_configs(nthreads, check_device=None):
    """CPU rows of BASELINE.md section 5's table for C2, C3, C4: one trial each (the reference's own usage), bounded samples; with the
    engine run in lock step for the agreement column."""
    out = {}
    out["C2"] = cpu_port("gmppi", 1, 1024, 1, nthreads, 1.0, check_device=check_device)
    out["C3"] = cpu_port("cemppi", 1, 150, 10, nthreads, 1.5, check_device=check_device, sigma_est="ss", elite_threshold=0.8)
    out["C4"] = cpu_port("cmamppi", 3, 4096, 10, nthreads, 4.0, max_steps=2, check_device=check_device, elite_threshold=0.8, cma_sigma=0.75)
    return out


def measure_sync_calls(policy, cars, Kc, Nc, steps, device, frozen=True, **kw):
    """The reference's own call pattern (src/examples/car_example.jl:203-207): ONE trial, the host owns the env, `act = pol(env)` is a
    synchronous call per MPC step -- state and pol.U in, control and the rolled pol.U out (mpopis_policy_call: one host wait).  Wall time of
    that call per MPC step, measured from this Python process (ctypes, pre-bound pointers), in three loops on handles with the same seed:
      frozen       the same start state every call (what mpopis_bench_policy_steps does: directly comparable with the row's resident ms_per_step;
                   not for :cmamppi, whose Σ update needs the state to move)
      closed_loop  env(act) between the calls (mpopis_env_step + mpopis_get_state play the host's env, not timed).  The kernels themselves
                   take longer at mid-lap states than at the reset state (rollouts that brake to a standstill / leave the anchor's
                   neighbourhood take the general paths), so this is compared with `resident_closed_loop_ms_per_step`: the same MPC steps
                   of the same seed run by mpopis_run_trials without any host round trip
      four_calls   closed loop through the pre-ABI-3 composition set_state + set_U + policy_step + get_U (four waits)"""
    import ctypes as C
    import numpy as np
    from mpopis_amd.engine import Engine
    cs = 2 * cars * H
    mk = lambda: Engine("car", cars, policy, Kc, H, batch=1, lam=LAM, alpha=1.0, ais_its=Nc, lam_ais=LAM_AIS, cov=np.tile([0.0625, 0.1], cars),
                        seed=20240000, device=device, **kw)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    res = {}
    for mode in (("frozen",) if frozen else ()) + ("closed_loop", "four_calls"):
        eng = mk()
        try:
            x, t, done = eng.get_state()
            U = np.zeros((1, cs)); ctl = np.zeros((1, 2 * cars)); rew = np.zeros(1)
            xp, tp, dnp, Up, cp, rp = (x.ctypes.data_as(dp), t.ctypes.data_as(ip), done.ctypes.data_as(ip), U.ctypes.data_as(dp), ctl.ctypes.data_as(dp), rew.ctypes.data_as(dp))
            L, h = eng.L, eng._h
            ts, failed = [], None
            for s_ in range(steps + 3):
                t0 = time.perf_counter()
                if mode != "four_calls":
                    rc = L.mpopis_policy_call(h, xp, tp, dnp, Up, None, cp, None, None, None)
                else:
                    rc = L.mpopis_set_state(h, xp, tp, dnp) or L.mpopis_set_U(h, Up) or L.mpopis_policy_step(h, None, cp, None, None, None, None, None) or L.mpopis_get_U(h, Up)
                dt = time.perf_counter() - t0
                if rc != 0:                               # :cmamppi closed loops can end in the reference's own PosDefException
                    failed = int(rc)
                    break
                if s_ >= 3:
                    ts.append(dt * 1e3)
                if mode != "frozen" and (L.mpopis_env_step(h, cp, rp) != 0 or L.mpopis_get_state(h, xp, tp, dnp) != 0):
                    break
            ts.sort()
            res[mode] = {"median": ts[len(ts) // 2] if ts else None, "mean": sum(ts) / len(ts) if ts else None, "min": ts[0] if ts else None,
                         "p90": ts[min(len(ts) - 1, int(0.9 * len(ts)))] if ts else None, "steps": len(ts), "stopped_with": failed}
        finally:
            eng.close()
    eng = mk()
    try:
        eng.run_trials(num_steps=2, laps=4)               # the same 3 warm-up steps
        t0 = time.perf_counter()
        eng.run_trials(num_steps=steps - 1, laps=4)
        resident = (time.perf_counter() - t0) * 1e3 / steps
    finally:
        eng.close()
    out = {"abi_sync_ms_per_step": (res["frozen"] if frozen else res["closed_loop"])["median"],
           "abi_sync_loop": "frozen start state (comparable with ms_per_step of this row)" if frozen else "closed loop (comparable with this row's closed-loop ms_per_step)",
           "abi_sync": {k: res[k] for k in res if k != "four_calls"}, "abi_four_call_ms_per_step": res["four_calls"]["median"], "abi_four_calls": res["four_calls"],
           "resident_closed_loop_ms_per_step": resident,
           "what": "one trial through the C ABI from this Python process (ctypes, pre-bound pointers): wall ms of the synchronous pol(env) call per MPC step; "
                   "resident_closed_loop = mpopis_run_trials over steps 3..%d of the same seed (policy step + env step on the device, no host round trip)" % (steps + 2)}
    return out


def alg_bytes(policy, cs):
    """SURVEY 8(d) / BASELINE.md section 3: 8*(3*cs+4) B per rollout, one more cs-row pass for the policies that adapt Σ′."""
    return 8 * ((4 if policy in ("μΣaismppi", "cemppi", "cmamppi", "pmcmppi") else 3) * cs + 4)


def measure_config(name, policy, cars, Kc, Nc, trials, steps, device, closed_loop=False, **kw):
    """One BASELINE config on this GPU, device RNG, `trials` resident trials: ms per MPC step (HIP events / wall around `steps` steps after a
    warm-up), rollouts/s, MPC steps/s, per-class kernel time of a step, the dominant class and its average launch.  closed_loop: the resident
    closed loop (mpopis_run_trials: policy step + env step per MPC step) -- used for :cmamppi, whose Σ update needs the state to move
    (DESIGN.md section 5); otherwise mpopis_bench_policy_steps like the headline workload."""
    import numpy as np
    from mpopis_amd.engine import Engine
    cs = 2 * cars * H
    eng = Engine("car", cars, policy, Kc, H, batch=trials, lam=LAM, alpha=1.0, ais_its=Nc, lam_ais=LAM_AIS, cov=np.tile([0.0625, 0.1], cars),
                 seed=20240000, device=device, **kw)
    try:
        if closed_loop:
            eng.run_trials(num_steps=1, laps=2)                                   # warm-up: 2 MPC steps
            eng.reset(); eng.set_U(np.zeros((trials, cs))); eng.seed(20240000)
            t0 = time.perf_counter()
            rec = eng.run_trials(num_steps=steps - 1, laps=2)
            ms = (time.perf_counter() - t0) * 1e3
            rollouts = float(rec[:, 14].sum())
            eng.reset(); eng.set_U(np.zeros((trials, cs))); eng.seed(20240000)
            tsteps = min(steps, 4)
            eng.timing_enable(True); eng.timing_reset()
            eng.run_trials(num_steps=tsteps - 1, laps=2)
        else:
            # warm-up by time, not by step count: a C2 step is 0.13-0.24 ms, and a handful of them after the ~0.1 s of allocations behind Engine()
            # leaves the clock where the idle GPU had it (C2 at 64 trials read 0.214 or 0.243 ms depending on what ran before)
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.3:
                eng.bench_policy_steps(steps)
            runs = sorted(eng.bench_policy_steps(steps) for _ in range(5))
            ms, rollouts = runs[2]                                                # median of five timed regions of `steps` steps
            tsteps = min(steps, 5)
            eng.timing_enable(True); eng.timing_reset()
            eng.bench_policy_steps(tsteps)
        tm = eng.timing_read()
        eng.timing_enable(False)
        # the rollout kernel with the chip to itself (ONE stream, HIP events around the rollout launches only): the contract's roofline formula for
        # this config -- alg_bytes x rollouts per launch / average launch time / 8 TB/s -- next to profiles/r06_c4_b64_kernel_stats.csv
        eng.set_overlap(1)
        if closed_loop:
            eng.reset(); eng.set_U(np.zeros((trials, cs))); eng.seed(20240000)
            eng.timing_enable(2); eng.timing_reset()
            rl_iso = float(eng.run_trials(num_steps=min(steps, 4) - 1, laps=2)[:, 14].sum())      # executed rollouts of exactly this call
        else:
            eng.bench_policy_steps(2)
            eng.timing_enable(2); eng.timing_reset()
            _, rl_iso = eng.bench_policy_steps(min(steps, 5))
        tm_iso = eng.timing_read()
        eng.timing_enable(False)
    finally:
        eng.close()
    iso = None
    if tm_iso.get("rollout", (0, 0))[1]:
        i_ms, i_n = tm_iso["rollout"]
        per_l = rl_iso / i_n
        iso = {"kernel": ("k_rollout_cars<%d,...>" % cars) if cars > 1 else "k_rollout_car<1,...>", "avg_launch_us": i_ms / i_n * 1e3, "launches": i_n, "rollouts_per_launch": per_l,
               "alg_bytes_per_rollout": alg_bytes(policy, cs), "frac": per_l * alg_bytes(policy, cs) / (i_ms / i_n * 1e-3) / (HBM_PEAK_GBS * 1e9), "schedule": "one stream"}
    per_step = {k: v[0] / tsteps for k, v in tm.items() if v[1]}
    dom = max(per_step, key=per_step.get)
    overlap = sum(per_step.values()) / (ms / steps)
    rps = rollouts / (ms * 1e-3)
    ba = alg_bytes(policy, cs)
    # the reference's own call pattern (one trial, synchronous pol(env) per MPC step through the C ABI), next to the resident figure
    sync = measure_sync_calls(policy, cars, Kc, Nc, 60 if Kc * Nc * cars < 20000 else 12, device, frozen=not closed_loop, **kw) if trials == 1 else {}
    return {**sync, "rollout_roofline": iso, "config": name, "trials": trials, "steps": steps, "ms_per_step": ms / steps, "rollouts_per_s": rps, "mpc_steps_per_s": trials * steps / (ms * 1e-3),
            "loop": "closed loop (mpopis_run_trials)" if closed_loop else "policy steps (mpopis_bench_policy_steps)",
            "kernel_ms_per_step": per_step, "dominant": {"class": dom, "avg_launch_us": tm[dom][0] / tm[dom][1] * 1e3, "share_of_kernel_time": per_step[dom] / sum(per_step.values())},
            "kernel_time_over_step_time": overlap,
            "schedule": ("multi-stream (the engine's automatic four-part schedule for this shape): kernel classes of different part-chains overlap, "
                         "their times are summed over the streams and per-launch durations include time-sharing") if overlap > 1.2 else "one stream",
            "alg_bytes_per_rollout": ba, "hbm_frac": rps * ba / (HBM_PEAK_GBS * 1e9), "fp64_reference_algorithm_frac": rps * 3.5e5 * cars / (FP64_PEAK_TFLOPS * 1e12)}


def baseline_configs(device, quick=False):
    """BASELINE.json configs[1..3] (C2, C3, C4) at one resident trial (the reference's own usage) and at chip-filling batches."""
    out = []
    S = 3 if quick else 1
    for trials in ((1, 64) if not quick else (1,)):
        out.append(measure_config("C2 Car-Racing 1-car :gmppi K=1024 H=50", "gmppi", 1, 1024, 1, trials, 40 // S, device))
    for trials in ((1, 64) if not quick else (1,)):
        out.append(measure_config("C3 Car-Racing 1-car :cemppi K=150 H=50 N=10 elite=0.8 Σ_est=:ss", "cemppi", 1, 150, 10, trials, 20 // S, device, sigma_est="ss", elite_threshold=0.8))
    for trials in ((1, 8, 32, 64) if not quick else (1,)):
        out.append(measure_config("C4 Car-Racing 3-car :cmamppi K=4096 H=50 N=10", "cmamppi", 3, 4096, 10, trials, 10 // S, device, closed_loop=True, elite_threshold=0.8, cma_sigma=0.75))
    return out


def source_sha(files):
    """sha256 over the named source files (what a committed measurement file is tied to)."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


PMC_SOURCES = ("mpopis_amd/csrc/kernels_rollout.hip", "mpopis_amd/csrc/car_dynamics.h")     # what profiles/pmc_rollout.json measured


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks ourselves (one process per GPU, rank r -> cuda:r, rendezvous
    on 127.0.0.1) by re-executing this file under torch.distributed.run with the same arguments; rank 0's JSON line is the only thing
    on stdout.  The trial loop this shards is src/examples/car_example.jl:170-188."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)       # (the round driver runs `bench.py --gpus 1 --steps 20 --warmup 5`: same as the defaults)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--trials-per-gpu", type=int, default=TRIALS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=100, help="extra repetitions of the timed region (after it), for median / min / max")
    ap.add_argument("--no-configs", action="store_true", help="skip the C2/C3/C4 block (BASELINE configs[1..3])")
    ap.add_argument("--quick-configs", action="store_true", help="C2/C3/C4 at one trial and fewer steps only (tests)")
    ap.add_argument("--no-midlap", action="store_true", help="skip the mid-lap-state variant of the workload (N = 1)")
    ap.add_argument("--multi-stream", action="store_true", help="(accepted for old command lines; the part-chain schedule is the engine's default now and the one-stream pass always runs)")
    # development aids for exercising the N > 1 control flow on a 1-GPU box (never used by the driver): all ranks on cuda:0 over gloo.
    # RCCL refuses two ranks on one device, so this also exercises the fall-back from the ABI gather to torch.distributed's.
    ap.add_argument("--allow-gather-fallback", action="store_true",
                    help="N > 1: if RCCL cannot be bound behind the C ABI (mpopis_comm_init), degrade to torch.distributed's gather and say so in the line. "
                         "Default for the nccl backend is STRICT: the run fails instead, so that a scaling number always exercised csrc/engine_comm.hip")
    ap.add_argument("--require-rccl", action="store_true", help="force the strict behaviour also with the gloo development backend (tests)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--same-gpu", action="store_true")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous only (gloo, no GPU): rank 0 prints {\"launch_check\": world}; used by the CPU test of the self-launch")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE (%d) != --gpus (%d): launch N ranks or none (bench.py spawns them itself)" % (world, args.gpus))
    if args.launch_check:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.ones(1)
            dist.all_reduce(t)
            assert int(t.item()) == world
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": world, "n_gpus": args.gpus}))
        return
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        else:
            dist.init_process_group("gloo")

    cdev = "cuda" if args.backend == "nccl" else "cpu"        # where the small torch.distributed tensors live
    from mpopis_amd import build
    if rank == 0:
        build.build()
    if dist is not None:
        dist.barrier()
    from mpopis_amd.engine import Engine
    B = args.trials_per_gpu
    # trial slot b on rank g is global trial g*B+b: seeds 20240000 + trial + 1 (seed+k, car_example.jl:187-188)
    eng = Engine("car", CARS, "μΣaismppi", K, H, batch=B, lam=LAM, alpha=1.0, ais_its=N_AIS, lam_ais=LAM_AIS,
                 cov=np.tile([0.0625, 0.1], CARS), seed=20240000 + rank * B, device=local_rank)

    # torch's lazy CUDA initialisation (context, allocator, its own streams) out of the way NOW: left to the first torch.cuda.synchronize() -- the
    # opening bracket of the timed region -- it overlaps the region's first steps and costs it 1.5 % (tools/first_sample.py: 5.35 vs 5.27 ms,
    # the eleven regions behind it unaffected).  torch is plumbing here (barrier, all_reduce of the times); its start-up is not the workload.
    # (Round 5 found the order to matter: HIP deals hardware queues to streams in creation order, and with torch initialised BEFORE the engine two of the
    # four part-chain streams shared a queue: 6.85 instead of 5.3 ms per step.  The engine now probes and repairs that itself, verify_part_streams.)
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()

    def sync(barrier=True):
        # opening bracket: barrier + device sync.  Closing bracket: device sync only -- the all_reduce(MAX) of the per-rank times that follows
        # orders the ranks, so no collective's latency is charged to the timed region
        if dist is not None and barrier:
            dist.barrier()
        torch.cuda.synchronize()

    # the one collective (per-trial summary records -> rank 0) goes through the C ABI's RCCL gather; communicator set-up is
    # outside the timed region.  If RCCL cannot be bound behind the ABI on this node, fall back to torch.distributed's gather
    # (same bytes) and say so in the output line.
    gather_path = "none (1 GPU)"

    def abi_comm_init(e, timeout_s=120.0):
        """RCCL communicator behind the ABI, guarded: a rank that cannot even load librccl makes every rank skip the collective
        init (it would block the others), and the init itself runs under a timeout so that a stuck rendezvous degrades to the
        torch.distributed gather instead of hanging the bench."""
        import threading
        try:
            Engine.comm_unique_id()                  # forces the dlopen of librccl
            can = 1
        except Exception:                            # noqa: BLE001
            can = 0
        flag = torch.tensor([can], device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            return "librccl not loadable behind the ABI on some rank"
        res = {}

        def run():
            try:
                if cdev == "cuda":
                    torch.cuda.set_device(local_rank)       # the current device is thread-local: a new thread starts on cuda:0, and the object broadcast below lands there
                e.comm_init_from_dist(dist)
                res["ok"] = True
            except Exception as ex:                  # noqa: BLE001
                res["err"] = str(ex)[:80]
        th = threading.Thread(target=run, daemon=True)
        th.start()
        th.join(timeout_s)
        good = 1 if res.get("ok") else 0
        flag = torch.tensor([good], device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return None
        return res.get("err", "timeout" if th.is_alive() else "failed on another rank")

    rccl_ranks_seen = None
    strict_rccl = world > 1 and (args.backend == "nccl" or args.require_rccl) and not args.allow_gather_fallback

    def rccl_failed(why):
        # every rank reaches this together (the verdicts above are all_reduce'd), so the group can be torn down cleanly
        if rank == 0:
            sys.stderr.write("bench.py: RCCL behind the C ABI is required for --gpus %d and could not be used: %s\n"
                             "          (pass --allow-gather-fallback to degrade to torch.distributed's gather and label the line)\n" % (world, why))
        dist.destroy_process_group()
        sys.exit(3)

    if dist is not None:
        err = abi_comm_init(eng)
        if err is None:
            rccl_ranks_seen = eng.comm_count()              # ncclCommCount of the communicator the gather runs on
            seen = torch.tensor([rccl_ranks_seen], device=cdev)
            dist.all_reduce(seen, op=dist.ReduceOp.MIN)
            if int(seen.item()) != world:
                err = "ncclCommCount reports %d ranks, expected %d" % (int(seen.item()), world)
        if err is not None and strict_rccl:
            rccl_failed(err)
        gather_path = "mpopis_gather_summary (RCCL behind the C ABI)" if err is None else "torch.distributed gather (ABI path: %s)" % err

    def summary_gather(e):
        """per-trial summary (first planned control pair of the rolled U + its norm): the only data that ever leaves a GPU"""
        Uh = e.get_U()
        nb = Uh.shape[0]
        if gather_path.startswith("mpopis"):
            rec = np.zeros((nb, 16))
            rec[:, :2] = Uh[:, :2]; rec[:, 2] = np.linalg.norm(Uh, axis=1); rec[:, 3] = float(rank)
            return e.gather_summary(rec, nb)
        rec = torch.tensor(np.concatenate([Uh[:, :2], np.linalg.norm(Uh, axis=1, keepdims=True), np.full((nb, 1), float(rank))], 1),
                           device=cdev, dtype=torch.float64)
        gl = [torch.zeros_like(rec) for _ in range(world)] if rank == 0 else None
        dist.gather(rec, gl, dst=0)
        return gl

    # Before the W warm-up steps: ~0.3 s of the same work, untimed, so that a fresh box has left its idle power state (clock ramp) and every
    # lazily loaded code object / LDS attribute is in place when the contract's warm-up starts.  Reported as config.prewarm_steps.
    eng.timing_enable(2)                     # HIP events (rollout kernel only) are used by the passes AFTER the timed region; created here, BEFORE the
    eng.timing_enable(False)                 # warm-up (16 k hipEventCreate calls take ~0.1 s of host time: with the GPU idle meanwhile the timed
    eng.bench_policy_steps(PREWARM_STEPS)    # region used to start at a lower clock).  The timed region itself records no events: in the default
    eng.bench_policy_steps(args.warmup)      # schedule (four streams) two records per rollout launch cost ~2 % of the step.
    sync()
    t0 = time.perf_counter()
    ms_dev, rollouts = eng.bench_policy_steps(args.steps)
    # summary stats only: per-trial record (control + mean cost) gathered to rank 0 over RCCL
    if dist is not None:
        summary_gather(eng)
    sync(barrier=False)
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=cdev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # R more repetitions of the same timed region (same bracketing), for the spread: median / min / max over R+1 samples.
    samples = [dt]
    for _ in range(max(0, args.repeats)):
        sync()
        t0r = time.perf_counter()
        eng.bench_policy_steps(args.steps)
        if dist is not None:
            summary_gather(eng)
        sync(barrier=False)
        tr = torch.tensor([time.perf_counter() - t0r], device=cdev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        samples.append(float(tr.item()))
    # Outside the timed region.  (1) The default schedule once more with HIP events around the rollout launches: how long a launch takes when the
    # part-chains time-share the chip (roofline.default_schedule / frac_default_schedule).
    eng.timing_enable(2); eng.timing_reset()
    _, rl_def = eng.bench_policy_steps(args.steps)
    tm = eng.timing_read()
    eng.timing_enable(False)
    # (2) The same workload on ONE stream (mpopis_set_overlap(h, 1)) -- every launch then has the chip to itself, which is what a per-kernel
    # roofline figure and per-class kernel times need.
    eng.set_overlap(1)
    eng.bench_policy_steps(3)
    eng.timing_enable(2); eng.timing_reset()
    one = sorted(eng.bench_policy_steps(args.steps) for _ in range(3))
    ms_one, rl_one = one[1]                  # median of three regions
    rl_one_all = sum(r for _, r in one)      # rollouts behind the launches the timing events of this pass cover
    tm_one = eng.timing_read()
    eng.timing_enable(True); eng.timing_reset()
    eng.bench_policy_steps(min(args.steps, 5))
    tm_all = eng.timing_read()
    eng.timing_enable(False)
    eng.set_overlap(0)

    # ---- the same workload at mid-lap states (N = 1 only; outside the timed region) -------------------------------------------------
    # The timed region keeps every trial at the reset state (the synthetic workload of the contract).  In a closed loop the cars are at speed
    # on curved track sections: more rollouts brake to a standstill or leave their anchor's neighbourhood, and the rollout kernel's general
    # paths run more often.  Reported next to the contract figure: 100 closed-loop MPC steps of all trials (mpopis_run_trials), and the timed
    # region's policy steps from the states reached there.
    midlap = None
    if world == 1 and not args.no_midlap:
        t0m = time.perf_counter()
        recm = eng.run_trials(num_steps=99, laps=4)
        dtm = time.perf_counter() - t0m
        rl_cl = float(recm[:, 14].sum())
        eng.bench_policy_steps(2)
        eng.timing_enable(2); eng.timing_reset()
        ms_m, rl_m = eng.bench_policy_steps(args.steps)
        tm_m = eng.timing_read()
        eng.timing_enable(False)
        agree_m = None
        if not args.no_cpu_baseline:
            try:
                agree_m = midlap_agreement(eng.get_state()[0], eng.get_U(), local_rank)
            except Exception as ex:                      # noqa: BLE001
                agree_m = {"error": str(ex)[:120]}
        midlap = {"max_rel_err_vs_cpu": agree_m,"closed_loop": {"what": "100 closed-loop MPC steps of all trials from the reset state (mpopis_run_trials: policy step + env step + bookkeeping on the device, no host round trip): "
                                          "the state moves, so a few to 20 % of the rollouts brake to a standstill inside the horizon and take the general sub-step",
                                  "ms_per_step": dtm / 100 * 1e3, "value": rl_cl / dtm},
                  "frozen_at_step_100": {"what": "the timed region's policy steps repeated from the states reached after those 100 steps (state frozen, pol.U keeps rolling: about half of the rollouts then stop -- the harshest mix)",
                                         "ms_per_step": ms_m / args.steps, "value": rl_m / (ms_m * 1e-3), "rollout_avg_launch_us": tm_m["rollout"][0] / max(1, tm_m["rollout"][1]) * 1e3}}

    # ---- strong scaling: BASELINE configs[4] as written = 64 trials in total, 64/N per GPU (N > 1 only) ------------------
    strong = None
    if world > 1 and TRIALS_PER_GPU % world == 0:
        Bs = TRIALS_PER_GPU // world
        eng_s = Engine("car", CARS, "μΣaismppi", K, H, batch=Bs, lam=LAM, alpha=1.0, ais_its=N_AIS, lam_ais=LAM_AIS,
                       cov=np.tile([0.0625, 0.1], CARS), seed=20240000, device=local_rank)
        eng_s.seed_slots([20240000 + 1 + rank + i * world for i in range(Bs)])       # trial k -> rank (k-1) mod N
        if gather_path.startswith("mpopis"):
            err_s = abi_comm_init(eng_s)
            if err_s is not None:
                if strict_rccl:
                    rccl_failed("second communicator (strong-scaling handle): " + err_s)
                gather_path = "torch.distributed gather (ABI path failed for the second communicator)"
        eng_s.bench_policy_steps(args.warmup)
        sync()
        t0s = time.perf_counter()
        _, rl_s = eng_s.bench_policy_steps(args.steps)
        summary_gather(eng_s)
        sync(barrier=False)
        dts = torch.tensor([time.perf_counter() - t0s], device=cdev, dtype=torch.float64)
        dist.all_reduce(dts, op=dist.ReduceOp.MAX)
        dts = float(dts.item())
        strong = {"scaling": "strong", "total_trials": TRIALS_PER_GPU, "trials_per_gpu": Bs, "value": rl_s * world / dts, "unit": "rollouts/s",
                  "ms_per_step": dts / args.steps * 1e3, "mpc_steps_per_s": TRIALS_PER_GPU * args.steps / dts,
                  # efficiency against ONE GPU running the same 64 trials: that is exactly what every rank did in the weak pass above
                  "one_gpu_ms_per_step": dt / args.steps * 1e3, "speedup_vs_one_gpu": dt / dts, "efficiency_vs_one_gpu": dt / dts / world,
                  "note": "64/N trials per GPU: below ~32 trials a GPU is latency-bound (one K=4096 trial = 64 waves on 1024 SIMDs)"}
        eng_s.close()

    if rank == 0:
        total_rollouts = rollouts * world
        value = total_rollouts / dt
        # roofline of the dominant kernel: from the ONE-STREAM pass (a launch = all trials, nothing else on the GPU while it runs) ...
        r_ms, r_n = tm_one["rollout"]
        r_avg_s = (r_ms / max(r_n, 1)) * 1e-3
        per_launch = rl_one_all / max(r_n, 1)        # rollouts one launch of the kernel processes
        ach_gbs = per_launch * BYTES_PER_ROLLOUT / r_avg_s / 1e9
        ach_tf = per_launch * FLOPS_PER_ROLLOUT / r_avg_s / 1e12
        # ... and the same formula on the timed region's own launches (default schedule: a launch covers one part-chain's trials and shares the
        # chip with the other chains' kernels, so its duration is not a kernel-in-isolation figure)
        d_ms, d_n = tm["rollout"]
        d_avg_s = (d_ms / max(d_n, 1)) * 1e-3
        d_per_launch = rl_def / max(d_n, 1)
        parts = max(1, round(d_n / max(1, args.steps * N_AIS)))
        # PMC-derived side fields: counters cannot be collected inside this run (rocprofv3 --pmc passes are separate runs of this same command,
        # tools/profile_round.sh); they come from profiles/pmc_rollout.json, which records the sha of the rollout kernel's sources it measured.
        # They are emitted only when that sha matches the files of THIS tree -- a stale file yields null + the reason, never old numbers.
        traffic, valu_busy, flops_exec, valu_per_rollout, clock_ghz = None, None, None, None, None
        pj = os.path.join(ROOT, "profiles", "pmc_rollout.json")      # written by tools/pmc_summary.py from separate --pmc passes
        sha_now = source_sha(PMC_SOURCES)
        pmc_state = "profiles/pmc_rollout.json missing"
        if os.path.exists(pj):
            try:
                pm = json.load(open(pj))
                if pm.get("source_sha") == sha_now:
                    traffic, valu_busy = pm.get("hbm_bytes_per_rollout") * per_launch, pm.get("valu_busy_frac")
                    flops_exec = pm.get("fp64_flops_per_rollout")
                    valu_per_rollout, clock_ghz = pm.get("valu_insts_per_rollout"), pm.get("effective_clock_ghz")
                    pmc_state = "profiles/pmc_rollout.json matches this tree (source_sha %s over %s)" % (sha_now, " + ".join(PMC_SOURCES))
                else:
                    pmc_state = "profiles/pmc_rollout.json is STALE (measured source_sha %s, this tree %s): PMC-derived fields are null" % (pm.get("source_sha"), sha_now)
            except Exception as ex:                      # noqa: BLE001
                pmc_state = "profiles/pmc_rollout.json unreadable: %s" % str(ex)[:60]
        srt = sorted(samples)
        med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
        step_bytes = total_rollouts / world * BYTES_PER_ROLLOUT            # per GPU: algorithmic bytes of the whole path in the timed region
        out = {
            "metric": "trajectory rollouts/sec (+ MPC steps/sec), Car-Racing K=4096 H=50",
            "value": value, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "mpc_steps_per_s": B * world * args.steps / dt,
            "config": {"workload": ("Car-Racing 1-car :μΣaismppi K=4096 H=50 N=10 λ=10 λ_ais=20, %d independent trials per GPU (BASELINE configs[4])" % B) if world == 1 else
                                   ("Car-Racing 1-car :μΣaismppi K=4096 H=50 N=10 λ=10 λ_ais=20, %d independent trials IN TOTAL = %d per GPU x %d GPUs (weak scaling of BASELINE configs[4]; "
                                    "configs[4] as written, 64 trials over all GPUs, is the `strong_scaling` block)" % (B * world, B, world)),
                       "total_trials": B * world,
                       "trials_per_gpu": B, "rollouts_per_step": int(B * N_AIS * K), "prewarm_steps": PREWARM_STEPS, "parallelism": ("trials sharded x%d, RCCL gather of summary stats" % world) if world > 1 else "one GPU: all trials resident, no collective"},
            "repeats": {"n": len(samples), "what": "the timed region repeated back to back (first sample = the contract's timed region = `value`)",
                        "ms_per_step": {"median": med / args.steps * 1e3, "min": srt[0] / args.steps * 1e3, "max": srt[-1] / args.steps * 1e3},
                        "value": {"median": total_rollouts / med, "max": total_rollouts / srt[0], "min": total_rollouts / srt[-1]}},
            "roofline": {"bound": "hbm", "limiter": "fp64_valu_issue", "kernel": "k_rollout_car<1, 4, false, true>", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "pmc_source": pmc_state,
                         "kernel_isolation": "one-stream pass of the same workload (mpopis_set_overlap(h, 1); = MPOPIS_NSPLIT=1), right after the timed region, same process: "
                                             "%.3f ms per step there; profiles/r06_bench_kernel_stats.csv is taken the same way" % (ms_one / args.steps),
                         "frac_default_schedule": d_per_launch * BYTES_PER_ROLLOUT / d_avg_s / 1e9 / HBM_PEAK_GBS,
                         "default_schedule": {"parts": parts, "rollout_avg_launch_us": d_avg_s * 1e6, "rollout_launches": d_n, "rollouts_per_launch": d_per_launch,
                                              "what": "the schedule of the timed region (`value`), measured again with HIP events right after it: the engine's default for this shape = %d skewed part-chains on their own HIP streams; "
                                                      "a rollout launch covers 1/%d of the trials and time-shares the chip with the other chains' kernels" % (parts, parts)},
                         "one_stream": {"ms_per_step": ms_one / args.steps, "value": rl_one / (ms_one * 1e-3)},
                         "frac_definition": "contract formula: whole-path algorithmic bytes per launch (SURVEY 8d: %d B x rollouts per launch) / the dominant kernel's average launch time / 8 TB/s" % BYTES_PER_ROLLOUT,
                         "what_binds": "FP64 VALU issue of the rollout kernel (HBM is at kernel_traffic_frac of peak: nothing is re-read; no MFMA in this kernel)",
                         "step_frac": step_bytes / dt / (HBM_PEAK_GBS * 1e9),
                         "kernel_traffic_frac": (traffic / r_avg_s / (HBM_PEAK_GBS * 1e9)) if traffic else None,
                         "fp64_executed_frac": (flops_exec * per_launch / r_avg_s / (FP64_PEAK_TFLOPS * 1e12)) if flops_exec else None,
                         "avg_launch_us": r_avg_s * 1e6, "launches": r_n, "rollouts_per_launch": per_launch, "alg_bytes_per_rollout": BYTES_PER_ROLLOUT,
                         "schedule": "achieved / frac / avg_launch_us: one-stream pass, %d launch per AIS iteration, all %d trials in one launch, nothing else on the GPU while it runs" % (max(1, round(r_n / (3 * args.steps * N_AIS))), B),
                         "bound_note": "achieved / peak / frac are the HBM figure the bench contract prescribes (`bound`); `limiter` names the resource that actually limits the kernel; valu_busy_frac from the PMC pass in profiles/",
                         "valu_busy_frac": valu_busy,
                         # what the FP64 VALU sustains: tools/mfma_rate.hip -- bare independent v_fma_f64 streams issue one wave-instruction per
                         # 5.0 cycles per SIMD at 4 waves per SIMD (the kernel's occupancy: 128 VGPRs), 4.6 at 8, 8.8 from a lone wave; the
                         # data-sheet rate is 4.0.  attained = (VALU wave-instructions per launch x 5.0 cycles / 1024 SIMDs / clock) / launch time
                         "issue_rate": ({"valu_wave_insts_per_rollout": valu_per_rollout, "cycles_per_inst_attainable_4_waves": 5.0, "cycles_per_inst_datasheet": 4.0,
                                         "clock_ghz": clock_ghz,
                                         "frac_of_attainable": valu_per_rollout * per_launch / 64.0 * 5.0 / 1024.0 / (clock_ghz * 1e9) / r_avg_s,
                                         "frac_of_datasheet": valu_per_rollout * per_launch / 64.0 * 4.0 / 1024.0 / (clock_ghz * 1e9) / r_avg_s,
                                         "source": "profiles/pmc_rollout.json (SQ_INSTS_VALU, clock from GRBM_GUI_ACTIVE) + tools/mfma_rate.hip"}
                                        if valu_per_rollout and clock_ghz else None),
                         "fp64_reference_algorithm_tflops": ach_tf, "fp64_peak_tflops": FP64_PEAK_TFLOPS,
                         "fp64_reference_algorithm_frac": ach_tf / FP64_PEAK_TFLOPS,
                         "note": "fp64_reference_* prices SURVEY 8(d)'s 3.5e5 flop-equivalents of the REFERENCE formulation per rollout; the kernel executes ~4x fewer (transcendental-free sub-step), so this can exceed 1"},
            "kernel_ms_per_step": {k: v[0] / max(1, min(args.steps, 5)) for k, v in tm_all.items() if v[1]},
            "kernel_ms_per_step_schedule": "one-stream pass (per-class times add up to its step time; in the default schedule the classes of different part-chains overlap)",
            "summary_gather": gather_path, "rccl_ranks_seen": rccl_ranks_seen, "rccl_required": bool(strict_rccl),
            "midlap_states": midlap,
        }
        if strong is not None:
            out["strong_scaling"] = strong
        if not args.no_configs and world == 1:
            out["configs"] = baseline_configs(local_rank, quick=args.quick_configs)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(check_device=local_rank)
            out["max_rel_err_vs_cpu"] = out["cpu_baseline"].pop("max_rel_err_vs_cpu")      # headline config: BASELINE.md section 5's agreement column
            if "configs" in out and not args.quick_configs:
                cpu = cpu_configs(out["cpu_baseline"]["cores"], check_device=local_rank)      # CPU rows of the same table, one trial each
                for c in out["configs"]:
                    if c["trials"] != 1:
                        continue                      # the agreement check and the CPU row are ONE-trial measurements: they go on the one-trial row only
                    row = dict(cpu[c["config"][:2]])
                    c["max_rel_err_vs_cpu"] = row.pop("max_rel_err_vs_cpu", None)
                    c["cpu_one_trial"] = row
        if world > 1:
            out["n1_only"] = "cpu_baseline, max_rel_err_vs_cpu and the configs block (C2/C3/C4, abi_sync_ms_per_step) are measured at N = 1 only (one GPU, rank 0's host cores)"
        emit(out)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
