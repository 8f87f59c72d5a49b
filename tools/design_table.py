"""Prints the DESIGN.md section-5 table from a bench line (profiles/r03_bench_line.json) so that the document quotes the JSON verbatim (CPU only)."""
import json, re, sys
d = json.loads([l for l in open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r03_bench_line.json") if l.startswith("{")][0])
cfg = {(c["config"][:2], c["trials"]): c for c in d["configs"]}
pct = lambda x: ("%.2g" % (100 * x)) if 100 * x < 100 else "%.0f" % (100 * x)
sp = lambda x: f"{x:,.0f}".replace(",", " ")
def hip(c, note):
    return "| | hip | %d | %s | %.3g | %s | %s %% | %s %% | %s |" % (c["trials"], note % c["ms_per_step"], c["rollouts_per_s"], sp(c["mpc_steps_per_s"]), pct(c["hbm_frac"]),
                                                                  pct(c["fp64_reference_algorithm_frac"]), "%s, %.0f µs" % (c["dominant"]["class"], c["dominant"]["avg_launch_us"]))
def cpu(c, name):
    u = c["cpu_one_trial"]
    return "| %s | cpu, %d threads | 1 | %s | %.3g | %.3g | | | |" % (name, u["threads"], sp(1e3 / u["mpc_steps_per_s"]) if u["mpc_steps_per_s"] < 1 else "%.3g" % (1e3 / u["mpc_steps_per_s"]), u["rollouts_per_s"], u["mpc_steps_per_s"])
rows = [cpu(cfg[("C2", 1)], "C2 `:gmppi` K=1024"), hip(cfg[("C2", 1)], "%.3f"), hip(cfg[("C2", 64)], "%.3f"),
        cpu(cfg[("C3", 1)], "C3 `:cemppi` K=150 N=10 `:ss`"), hip(cfg[("C3", 1)], "**%.2f** (round 2 kernels: 2.47)"), hip(cfg[("C3", 64)], "**%.2f** (2.93)"),
        cpu(cfg[("C4", 1)], "C4 3-car `:cmamppi` K=4096 N=10 (closed loop)"), hip(cfg[("C4", 1)], "**%.2f** (round 2 kernels: 6.57)"), hip(cfg[("C4", 8)], "**%.2f** (8.31)"),
        hip(cfg[("C4", 32)], "%.1f (18.9)")]
r64 = hip(cfg[("C4", 64)], "**%.1f** (31.3)")
rows.append(re.sub(r"\| [a-z]+, \d+ µs \|$", "| four part-chains overlap (automatic from 48 slots) |", r64))
cb, rp, rf = d["cpu_baseline"], d["repeats"]["ms_per_step"], d["roofline"]
rows.append("| C5 `:μΣaismppi` K=4096 N=10 (headline) | cpu, %d threads / 1 thread | 1 | %.0f / %.0f | %.3g / %.3g | %.1f | | | |" % (
    cb["cores"], 1e3 / cb["mpc_steps_per_s"], 40960 / cb["value_1thread"] * 1e3, cb["value"], cb["value_1thread"], cb["mpc_steps_per_s"]))
rows.append("| | hip, one stream (`value`) | 64 | **%.2f** (median of 11: %.2f, min %.2f, max %.2f) | **%.3g** | %s | %.1f %% (`step_frac`) | %.0f %% | rollout, %.1f µs ⇒ `roofline.frac` %.3f |" % (
    d["ms_per_step"], rp["median"], rp["min"], rp["max"], d["value"], sp(d["mpc_steps_per_s"]), 100 * rf["step_frac"], 100 * d["value"] * 3.5e5 / 78.6e12, rf["avg_launch_us"], rf["frac"]))
ms = rf["multi_stream"]
if isinstance(ms, dict):
    rows.append("| | hip, opt-in four-part schedule | 64 | %.2f | %.3g | | | | |" % (ms["ms_per_step"], ms["value"]))
print("\n".join(rows))
