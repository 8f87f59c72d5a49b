"""simulate_car_racing / simulate_mountaincar (src/examples/car_example.jl:51-416,
mountaincar_example.jl:49-207): the trial loop runs as ONE device-resident batch per rank (all trials are
independent, car_example.jl:170), optionally sharded over ranks (trial k -> rank (k-1) mod G) with one RCCL
gather of the per-trial summary records (mpopis_gather_summary behind the C ABI when the process group is
nccl == RCCL; torch.distributed's own gather for the gloo CPU tests).  Prints the reference's tables."""
import math
import time
import numpy as np

from .engine import Engine
from ._lib import MPOPISError, ERR_ARG, RECORD_LEN


def quantile_ci(x, p=0.05, q=0.5):
    """src/examples/example_utils.jl:2-10 (order-statistic CI of the median, normal approximation)."""
    x = np.sort(np.asarray(x, dtype=np.float64))
    n = len(x)
    zm, zp = -1.959963984540054, 1.959963984540054          # quantile(Normal(), p/2), p = 0.05
    if p != 0.05:
        from statistics import NormalDist
        zm, zp = NormalDist().inv_cdf(p / 2), NormalDist().inv_cdf(1 - p / 2)
    j = max(int(math.ceil(n * q + zm * math.sqrt(n * q * (1 - q)))), 1)
    k = min(int(math.ceil(n * q + zp * math.sqrt(n * q * (1 - q)))), n)
    return x[j - 1], float(np.quantile(x, q)), x[k - 1]


def _summary(rows):
    """AVE/STD/MED/L95/U95/MIN/MAX per column (car_example.jl:328-410)."""
    rows = np.asarray(rows, dtype=np.float64)
    out = {}
    out["AVE"] = rows.mean(0)
    out["STD"] = rows.std(0, ddof=1) if rows.shape[0] > 1 else np.full(rows.shape[1], np.nan)
    q = np.array([quantile_ci(rows[:, c]) for c in range(rows.shape[1])])
    out["MED"], out["L95"], out["U95"] = q[:, 1], q[:, 0], q[:, 2]
    out["MIN"], out["MAX"] = rows.min(0), rows.max(0)
    return out


def _gather_records(rec, dist):
    """One collective for summary stats only: RCCL gather to rank 0."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return rec
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    # ranks may hold different numbers of trials: pad to the maximum (row id 0 marks padding)
    n = torch.tensor([rec.shape[0]], device=dev)
    dist.all_reduce(n, op=dist.ReduceOp.MAX)
    nmax, width = int(n.item()), rec.shape[1]
    pad = np.zeros((nmax, width))
    pad[:rec.shape[0]] = rec
    t = torch.from_numpy(pad).to(dev)
    gl = [torch.zeros_like(t) for _ in range(dist.get_world_size())] if dist.get_rank() == 0 else None
    dist.gather(t, gl, dst=0)
    if dist.get_rank() != 0:
        return None
    out = np.concatenate([g.cpu().numpy() for g in gl], axis=0)
    return out[out[:, 0] > 0]


def assemble_gathered(parts, world):
    """Rows of the summary table from what mpopis_gather_summary hands rank 0: parts[g] = (n_g, RECORD_LEN) records of rank g's
    trials in slot order, i.e. of trials g+1, g+1+world, ... (shard_trials).  Column layout of the result: [trial id, record[0:15],
    0, Ex Time] -- the sending rank stored its wall time in record slot 15 (the status slot, unused after a successful run)."""
    rows = []
    for g, part in enumerate(parts):
        for i, row in enumerate(np.asarray(part, dtype=np.float64).reshape(-1, RECORD_LEN)):
            rows.append(np.concatenate([[1 + g + i * world], row[:15], [0.0], [row[15]]]))
    return np.array(rows).reshape(-1, RECORD_LEN + 2)


def shard_trials(num_trials, rank, world):
    """trial k (1-based) -> rank (k-1) mod world; returns this rank's 1-based trial ids."""
    return [k for k in range(1, num_trials + 1) if (k - 1) % world == rank]


def simulate_car_racing(num_trials=1, num_steps=200, num_cars=1, policy_type="cemppi", laps=2, num_samples=150, horizon=50,
                        λ=10.0, α=1.0, U0=None, cov_mat=None, ais_its=10, λ_ais=20.0, ce_elite_threshold=0.8, ce_Σ_est="ss",
                        cma_σ=0.75, cma_elite_threshold=0.8, state_x_sigma=0.0, state_y_sigma=0.0, state_ψ_sigma=0.0,
                        seed=None, log_runs=True, device=0, dist=None, quiet=False):
    """Returns (records, summary) on rank 0 (None elsewhere).  Differences from the reference harness, all
    forced by the platform: plotting/GIF options are not offered (out of scope); the state noise (single car only,
    car_example.jl:224-236) is drawn from the trial's device stream instead of the env's MersenneTwister.
    ce_Σ_est defaults to :ss like the reference (car_example.jl:66); :mle is the other supported estimator."""
    pt = str(policy_type).lstrip(":")
    rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    if seed is None:
        seed = int(np.random.default_rng().integers(1, 10 ** 10))
    U0 = np.zeros(num_cars * 2) if U0 is None else np.asarray(U0, dtype=np.float64)
    cov_mat = np.tile([0.0625, 0.1], num_cars) if cov_mat is None else cov_mat
    if world > 1:
        sd = [seed]                                    # the seed must be common to all ranks
        dist.broadcast_object_list(sd, src=0)
        seed = sd[0]
    mine = shard_trials(num_trials, rank, world)
    t0 = time.time()
    # every rank keeps a handle (even with no trials: it takes part in the collective); all of a rank's trials
    # k0, k0+G, ... live in ONE resident batch, slot i seeded like the reference's trial k_i: seed!(pol, seed + k) (:187-188)
    eng = Engine("car", num_cars, pt, num_samples, horizon, batch=max(len(mine), 1), lam=λ, alpha=α, ais_its=ais_its, lam_ais=λ_ais,
                 elite_threshold=(cma_elite_threshold if pt == "cmamppi" else ce_elite_threshold), sigma_est=str(ce_Σ_est).lstrip(":"),
                 cma_sigma=cma_σ, seed=seed, device=device, cov=cov_mat, U0=U0)
    r = np.zeros((0, RECORD_LEN))
    if mine:
        eng.seed_slots([seed + k for k in mine])
        eng.set_state_noise(state_x_sigma, state_y_sigma, state_ψ_sigma)
        r = eng.run_trials(num_steps, laps)
    ex_time = time.time() - t0                 # Ex Time of this rank's batch (its trials run concurrently)
    n_max = (num_trials + world - 1) // world
    if world > 1 and dist.get_backend() == "nccl":
        # RCCL gather behind the C ABI (mpopis_gather_summary); slot 15 of a record (status) carries the rank's Ex Time
        r = r.copy()
        r[:, 15] = ex_time
        eng.comm_init_from_dist(dist)
        parts = eng.gather_summary(r, n_max)
        eng.close()
        if parts is None:
            return None, None
        allrec = assemble_gathered(parts, world)
    else:
        eng.close()
        rec = np.array([np.concatenate([[k], r[i], [ex_time]]) for i, k in enumerate(mine)]).reshape(-1, RECORD_LEN + 2)
        allrec = _gather_records(rec, dist)
    if allrec is None:
        return None, None
    allrec = allrec[np.argsort(allrec[:, 0])]
    cols = [1, 2, 3] + [4 + i for i in range(laps)] + [8, 9, 10, 11, 12, 13] + ([14] if num_cars > 1 else []) + [allrec.shape[1] - 1]
    table = allrec[:, cols]
    summ = _summary(table)
    if log_runs and not quiet:
        hdr = "Trial    #: %12s : %7s: %12s" % ("Reward", "Steps", "Reward/Step")
        hdr += "".join(" : %6s%d" % ("lap ", i + 1) for i in range(laps))
        hdr += " : %7s : %7s : %7s : %7s : %7s : %7s" % ("Mean V", "Max V", "Mean β", "Max β", "β Viol", "T Viol")
        hdr += (" : %7s" % "C Viol" if num_cars > 1 else "") + " : %7s" % "Ex Time"
        print(hdr)
        for row, full in zip(table, allrec):
            print("Trial %4d: " % int(full[0]) + " : ".join("%12.2f" % v if i in (0, 2) else "%7.2f" % v for i, v in enumerate(row)))
        print("-----------------------------------")
        for name in ("AVE", "STD", "MED", "L95", "U95", "MIN", "MAX"):
            print("Trials %3s: " % name + " : ".join("%12.2f" % v if i in (0, 2) else "%7.2f" % v for i, v in enumerate(summ[name])))
    return allrec, summ


def _simulate_simple(env_kind, ss, num_trials=1, num_steps=200, policy_type="cemppi", num_samples=20, horizon=15, λ=0.1, α=1.0, U0=(0.0,),
                         cov_mat=(1.5,), ais_its=5, λ_ais=0.1, ce_elite_threshold=0.8, ce_Σ_est="mle", cma_σ=0.75,
                         cma_elite_threshold=0.8, seed=None, x0=None, log_runs=True, device=0, quiet=False):
    pt = str(policy_type).lstrip(":")
    if seed is None:
        seed = int(np.random.default_rng().integers(1, 10 ** 10))
    eng = Engine(env_kind, 0, pt, num_samples, horizon, batch=num_trials, lam=λ, alpha=α, ais_its=ais_its, lam_ais=λ_ais,
                 elite_threshold=(cma_elite_threshold if pt == "cmamppi" else ce_elite_threshold), sigma_est=str(ce_Σ_est).lstrip(":"),
                 cma_sigma=cma_σ, seed=seed, device=device, cov=np.asarray(cov_mat, dtype=np.float64), U0=np.asarray(U0, dtype=np.float64))
    if x0 is not None:
        eng.set_state(np.asarray(x0, dtype=np.float64).reshape(num_trials, ss))
    t0 = time.time()
    rec = eng.run_trials(num_steps, 0)
    ex = time.time() - t0
    eng.close()
    table = np.concatenate([rec[:, :3], np.full((num_trials, 1), ex)], 1)
    summ = _summary(table)
    if log_runs and not quiet:
        print("Trial    #: %12s : %7s: %12s : %7s" % ("Reward", "Steps", "Reward/Step", "Ex Time"))
        for k, row in enumerate(table):
            print("Trial %4d: %12.2f : %7d: %12.2f : %7.2f" % (k + 1, row[0], int(row[1]), row[2], row[3]))
        print("-----------------------------------")
        for name in ("AVE", "STD", "MED", "L95", "U95", "MIN", "MAX"):
            print("Trials %3s: %12.2f : %7.2f: %12.2f : %7.2f" % ((name,) + tuple(summ[name])))
    return rec, summ


_SIMPLE_KW = dict(num_trials=1, num_steps=200, policy_type="cemppi", num_samples=20, horizon=15, λ=0.1, α=1.0, U0=(0.0,),
                  cov_mat=(1.5,), ais_its=5, λ_ais=0.1, ce_elite_threshold=0.8, ce_Σ_est="mle", cma_σ=0.75,
                  cma_elite_threshold=0.8, seed=None, x0=None, log_runs=True, device=0, quiet=False)


def simulate_mountaincar(**kw):
    """mountaincar_example.jl:49-207 (same keyword arguments and defaults); x0: per-trial start positions
    (the reference draws x ~ U(-0.6,-0.4) from an unseeded RNG; default here -0.5)."""
    a = dict(_SIMPLE_KW); _check_kw(a, kw); a.update(kw)
    if a["x0"] is not None:
        x = np.asarray(a["x0"], dtype=np.float64).reshape(a["num_trials"])
        a["x0"] = np.stack([x, np.zeros(a["num_trials"])], 1)
    return _simulate_simple("mountaincar", 2, **a)


def simulate_cartpole(**kw):
    """cartpole_example.jl:8-190 (same keyword arguments and defaults); x0: (num_trials, 4) start states
    [x, xdot, theta, thetadot].  The reference draws 0.1*rand(4) - 0.05 from an unseeded MersenneTwister (:111);
    default here: the same box drawn from numpy's default_rng(seed + k)."""
    a = dict(_SIMPLE_KW); _check_kw(a, kw); a.update(kw)
    if a["seed"] is None:
        a["seed"] = int(np.random.default_rng().integers(1, 10 ** 10))
    if a["x0"] is None:
        a["x0"] = np.stack([0.1 * np.random.default_rng(a["seed"] + k + 1).random(4) - 0.05 for k in range(a["num_trials"])])
    return _simulate_simple("cartpole", 4, **a)


def _check_kw(allowed, kw):
    bad = [k for k in kw if k not in allowed]
    if bad:
        raise TypeError("unexpected keyword argument(s): %s" % ", ".join(bad))
