"""One engine-vs-oracle parity case of the randomised sweep (test infrastructure: uses the oracle).  Shared by tests/dev/fuzz_parity.py (the
open-ended campaign) and tests/test_gpu_fuzz_slice.py (the fixed-seed slice that runs in the GPU suite).
A few costs per case may differ beyond 1e-7 without being a bug: rollouts that brake to a standstill chatter; a case
allows ncars*K/200 of them per slot as long as control and U agree to 1e-6 (the committed shape tests identify those rollouts from the
oracle's own trajectory instead).  Round 5: a third of the slots of the policies that do not rank costs START braked to (or towards) a standstill;
those slots are held to the north star's 1e-5 on the control with up to a quarter of their costs differing."""
import os
import numpy as np

# every use of the braked-start yardstick (a slot whose control is beyond 1e-5 of the oracle's but within ten times the oracle's distance from its own 1e-13
# neighbour) is recorded here; the slice test fails when there are more than a handful, so a regression of the general force rules cannot hide behind it
waivers = []


def tag_of(c):
    extra = "".join(" %s=%g" % (k, c[k]) for k in ("lam", "alpha", "lam_ais", "elite", "cma_sigma") if k in c)
    return "%s cars=%d K=%d T=%d N=%d B=%d split=%d est=%s rng=%s seed=%d%s" % (c["kind"], c["ncars"], c["K"], c["T"], c["N"], c["B"], c["split"], c["est"],
                                                                                  "dev" if c["device_rng"] else "inj", c["seed"], extra)


def run_case(O, Engine, MPOPISError, track, c, rng, steps=2, oracle_threads=8):
    """c: dict(kind, ncars, K, T, N, B, split, est, device_rng, seed).  Returns (status, messages): status in {"ok", "fail", "refused"}."""
    kind, ncars, K, T, N, B = c["kind"], c["ncars"], c["K"], c["T"], c["N"], c["B"]
    est, device_rng, seed = c["est"], c["device_rng"], c["seed"]
    lam, alpha, lam_ais, elite, cma_sigma = c.get("lam", 10.0), c.get("alpha", 1.0), c.get("lam_ais", 20.0), c.get("elite", 0.8), c.get("cma_sigma", 0.75)
    cs = 2 * ncars * T
    cov = np.tile([0.0625, 0.1], ncars)
    tag = tag_of(c)
    try:
        eng = Engine("car", ncars, kind, K, T, batch=B, lam=lam, alpha=alpha, ais_its=N, lam_ais=lam_ais, elite_threshold=elite, sigma_est=est, cma_sigma=cma_sigma,
                     cov=cov, track=track, seed=seed)
    except MPOPISError as e:
        return "refused", ["create refused: %s %s" % (tag, e)]
    msgs, notes = [], []
    try:
        eng.set_overlap(c["split"])
        envs, pols, braked = [], [], []
        # Braked start states put most rollouts at the sign(Vx) flip, where a rollout's 50-step cost differs by up to 1e-1 between any two evaluation
        # orders.  Policies that act on the RANK of a cost (:cemppi's elite set, :cmamppi's rank weights) turn that into a different elite set, and the
        # multi-car slots of the adaptive policies (every rollout of every car at the flip) amplify it over their iterations: engine and oracle part
        # ways by 1e-4 .. 1e-2 on the control there -- the round-4 engine fails the same cases of a campaign with the same three digits -- and a 1e-13
        # nudge of pol.U (the yardstick below) does not always reproduce a divergence that starts inside the rollouts.  Those slots keep the driving
        # start states; real closed-loop states are tests/test_gpu_midlap_parity.py's business.
        may_brake = kind not in ("cemppi", "cmamppi") and (ncars == 1 or kind in ("mppi", "gmppi"))
        for b in range(B):
            e = O.OracleEnv("car", ncars, track=track)
            for _ in range(int(rng.integers(0, 30))):
                e.step(np.clip(np.tile([0.05, 0.5], ncars) + 0.2 * rng.standard_normal(2 * ncars), -1, 1))
            braked.append(bool(may_brake and rng.integers(0, 3) == 0))
            if braked[-1]:                                     # a third of the slots: brake towards (or to) a standstill first -- the general force rules of
                for _ in range(int(rng.integers(1, 16))):      # the sub-step and the reference's sign(Vx) chatter (src/envs/car_racing.jl:311) from the start state on
                    e.step(np.clip(np.tile([0.0, -0.7], ncars) + 0.2 * rng.standard_normal(2 * ncars), -1, 1))
            envs.append(e)
            pols.append(O.OraclePolicy(kind, e, K, T, lam=lam, alpha=alpha, U0=np.zeros(2 * ncars), cov=cov, N=N, lam_ais=lam_ais, elite_threshold=elite,
                                       sigma_est=est, cma_sigma=cma_sigma, nthreads=oracle_threads))
        eng.set_state(np.stack([e.state for e in envs]))

        def oracle_self_distance(b, U_before, Zb, dib, dub, ref):
            """the oracle against itself, pol.U nudged by 1e-13 relative, same state and draws: (control, U) distance = the policy's own conditioning"""
            p2 = O.OraclePolicy(kind, envs[b], K, T, lam=lam, alpha=alpha, U0=np.zeros(2 * ncars), cov=cov, N=N, lam_ais=lam_ais, elite_threshold=elite,
                                sigma_est=est, cma_sigma=cma_sigma, nthreads=oracle_threads)
            p2.U = U_before * (1.0 + 1e-13)
            r2 = p2(envs[b], Zb, dib, dub)
            if r2["status"] != 0:
                return float("inf"), float("inf")
            return float(np.abs(r2["control"] - ref["control"]).max()), float(np.abs(p2.U - pols[b].U).max())
        for step in range(steps):
            if kind == "mppi":
                Z = rng.standard_normal((B, T, K, 2 * ncars))
            elif device_rng:
                Z = np.stack([np.stack([O.philox_normals(seed + b + 1, step, n, cs * K).reshape(K, cs) for n in range(N)]) for b in range(B)])
            else:
                Z = rng.standard_normal((B, N, K, cs))
            if device_rng:
                dd = [[O.philox_resample_draws(seed + b + 1, step, n | 0x80000000, K) for n in range(max(N - 1, 1))] for b in range(B)]
                di = np.array([[d[0] for d in row] for row in dd], dtype=np.int32)
                du = np.array([[d[1] for d in row] for row in dd])
            else:
                di = rng.integers(0, K, (B, max(N - 1, 1), K)).astype(np.int32)
                du = rng.random((B, max(N - 1, 1), K))
            U_before = [pols[b].U.copy() for b in range(B)]
            refs = [pols[b](envs[b], Z[b], di[b], du[b]) for b in range(B)]
            worst = min(r["status"] for r in refs)
            try:
                got = eng.policy_step(None if device_rng else Z, None if device_rng else di, None if device_rng else du, want_E=(kind != "mppi"))
            except MPOPISError as e:
                if e.code != worst:
                    msgs.append("FAIL status %s step %d engine %d oracle %d" % (tag, step, e.code, worst))
                break
            if worst:
                msgs.append("FAIL status %s engine ok, oracle %d" % (tag, worst))
                break
            U = eng.get_U()
            for b in range(B):
                r = refs[b]
                # relative to the cost's own scale, floored by the number of summed per-step rewards (each O(10)): a multi-car cost can cancel to ~1e-2 -- positive speed
                # rewards against the negative distance terms (multi-car_racing.jl:145-158) -- and 5e-9 of rounding on a sum of 200 terms then read as 4e-6 "relative"
                # (seen once in 6 000 cases: four cars, every one of the 34 flagged rollouts had |cost| < 0.07)
                rel = np.abs(got["cost"][b] - r["cost"]) / (np.abs(r["cost"]) + float(T * ncars))
                # alpha < 1 puts gamma U_orig' Σ'^-1 (V - U_orig) into every cost (:272): under the Σ-adapting policies Σ' is a K-sample scatter + 1e-8 I with
                # cond up to 1e8, and the engine's triangular solves and the oracle's explicit inverse differ by cond * eps on that term -- measured 1e-7 ... 7e-6
                # on the cost with the controls at 1e-11 (FUZZ_WILD): there the cost is held to the north star's 1e-5, not to 1e-7
                nbad = int((rel > (1e-7 if alpha == 1.0 else 1e-5)).sum())
                if alpha != 1.0:
                    # ... and part of that term, gamma U_orig' Σ'^-1 (pol.U - U_orig), is the SAME for every rollout of an iteration: with a collapsed Σ' (lambda_ais ~ 1:
                    # cond 1e10) it is a large number known to a few digits only -- seen: all 64 costs of a slot 6e-2 apart, the control agreeing to 3e-14 -- and it
                    # cancels in the weights (utils.jl:79-86 subtracts the minimum).  What is compared for alpha < 1 is therefore what the costs are used for:
                    # the weights of the final reweighting
                    nbad = int((np.abs(got["weights"][b] - r["weights"]) > 1e-7).sum())
                ea = float(np.abs(got["control"][b] - r["control"]).max())
                # pol.U relative to its own size: under :cmamppi with a large step factor the step size sigma runs away (every control saturates at the bounds, the
                # costs stop depending on the mean) and |pol.U| grows to 1e2 ... 1e4 -- an absolute 1e-5 there asks for 1e-9 relative after a 10^3 amplification
                uscale = max(1.0, float(np.abs(pols[b].U).max()))
                eu = float(np.abs(U[b] - pols[b].U).max()) / uscale
                idx_ok = True
                if kind == "pmcmppi" and r["iters_run"] > 1:
                    n_it = r["iters_run"]
                    idx_ok = bool(np.array_equal(got["res_idx0"][b][:n_it - 1], r["res_idx0"][:n_it - 1]))      # resampling indices: bit-exact
                # a slot that starts at (or near) a standstill: most of its rollouts chatter, their 50-step costs differ between any two evaluation
                # orders (tests/test_gpu_standstill.py: 6-14 % of them, by up to 1e-1); what is held there is the north star's bound on the control
                allow, tol = (max(2, ncars * K // 4), 1e-5) if braked[b] else (max(2, ncars * K // 200), 1e-6)
                if alpha == 1.0 and kind != "mppi" and nbad > allow and not braked[b]:
                    # more cost deviations than the flat allowance (seen once in 6 000 cases: four cars, H = 50, 34 of 1024 rollouts against 20 allowed, the control agreeing to
                    # 1e-12): are they all of the standstill class?  The oracle's own trajectories of its final samples say so -- some car of the rollout within one
                    # sub-step's brake impulse of Vx = 0 at a logged state (tests/test_gpu_standstill.py) -- and then they count as that class, not as failures
                    # First on IDENTICAL samples: from the second iteration on each side rolls out its OWN samples, ~1e-8 apart, and a sensitive rollout turns that
                    # into 1e-6 on its cost; the engine's final noise matrix through the oracle's model removes that.
                    cost_same, traj = pols[b].simulate_model(U_before[b], np.ascontiguousarray(got["E"][b].T), log=True)
                    rel_same = np.abs(got["cost"][b] - cost_same) / (np.abs(cost_same) + float(T * ncars))
                    stalled = np.abs(traj.reshape(K, T, ncars, 8)[:, :, :, 3]).min(axis=(1, 2)) < 0.12
                    unexplained = int(((rel_same > 1e-7) & ~stalled).sum())
                    if unexplained <= 2:
                        nbad = unexplained
                if alpha != 1.0:
                    tol = 1e-5                                   # (the Σ'^-1 term above perturbs the weights at 1e-7: controls measured up to 2.6e-6 -- the north star's bound holds)
                bad = got["iters_run"][b] != r["iters_run"] or nbad > allow or ea > tol or eu > 10 * tol or not idx_ok
                # (never for the non-adaptive policies: with N = 1 nothing feeds on perturbed weights, the 1e-5 on the control holds as it stands)
                if bad and braked[b] and got["iters_run"][b] == r["iters_run"] and kind not in ("mppi", "gmppi") and nbad <= allow and idx_ok:
                    # the yardstick: is the engine farther from the oracle than ten times the oracle's distance from its own 1e-13 neighbour?
                    sa, su = oracle_self_distance(b, U_before[b], Z[b], di[b], du[b], r)
                    if ea <= max(tol, 10.0 * sa) and eu <= max(10 * tol, 10.0 * su):
                        bad = False
                        msgs_note = "note %s step %d slot %d: braked start, control %.1e U %.1e within 10x the oracle's self-distance %.1e / %.1e" % (tag, step, b, ea, eu, sa, su)
                        waivers.append(msgs_note)
                        if len(notes) < 3:
                            notes.append(msgs_note)
                if bad and os.environ.get("FUZZ_DUMP"):
                    # everything needed to replay this slot on its own (tests/dev/fuzz_replay.py)
                    np.savez(os.environ["FUZZ_DUMP"], c=np.array([str(c)]), step=step, slot=b, x0=np.stack([e.state for e in envs]), U_before=np.stack(U_before),
                             cost_dev=got["cost"][b], cost_ref=r["cost"], E_dev=got["E"][b] if "E" in got else np.zeros(1), E_ref=r["E"] if "E" in r else np.zeros(1), U_last=r.get("U_last", np.zeros(1)),
                             control_dev=got["control"][b], control_ref=r["control"], weights_dev=got["weights"][b], weights_ref=r["weights"])
                if bad:
                    msgs.append("FAIL %s step %d slot %d iters %d %d cost-bad %d max rel %.2e ctrl %.2e U %.2e (|U| %.1e) idx %s" % (
                        tag, step, b, got["iters_run"][b], r["iters_run"], nbad, rel.max(), ea, eu, uscale, idx_ok))
            if msgs:
                break
    finally:
        eng.close()
    return ("fail" if msgs else "ok"), msgs
