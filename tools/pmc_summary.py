"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc_r01/*/pmc_counter_collection.csv) per kernel:
mean counter value per dispatch and mean duration.  Writes profiles/r04_pmc_summary.csv and
profiles/pmc_rollout.json (HBM bytes per launch of the rollout kernel, used by bench.py)."""
import csv, glob, hashlib, json, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_r01"
tag = sys.argv[2] if len(sys.argv) > 2 else "r06"
mode = sys.argv[3] if len(sys.argv) > 3 else "c5"          # "c4": the passes ran tools/prof_c4.py 64 (3-car :cmamppi) -> profiles/pmc_rollout_3car.json
MATCH, CS, CARS, OUTJ = (("k_rollout_cars<3", 300, 3, "profiles/pmc_rollout_3car.json") if mode == "c4" else ("k_rollout_car<1", 100, 1, "profiles/pmc_rollout.json"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "*", "pmc_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
names = sorted({c for k in acc for c in acc[k]})
rows = []
for k in sorted(acc, key=lambda k: -sum(dur[k])):
    if not k.startswith("mpopis::"):
        continue
    row = {"kernel": k, "dispatches_per_pass": len(dur[k]) // max(1, len(acc[k])), "avg_us": sum(dur[k]) / len(dur[k])}
    for c in names:
        v = acc[k].get(c)
        row[c] = sum(v) / len(v) if v else ""
    rows.append(row)
os.makedirs("profiles", exist_ok=True)
with open("profiles/%s_pmc_summary.csv" % tag, "w", newline="") as fo:
    w = csv.DictWriter(fo, fieldnames=["kernel", "dispatches_per_pass", "avg_us"] + names)
    w.writeheader()
    w.writerows(rows)
for r in rows[:8]:
    print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()})
ro = [r for r in rows if MATCH in r["kernel"]]
if ro:
    r = ro[0]
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts half the bytes of a
    # wide (16 B/lane) coalesced stream; this kernel's E loads are 8 B/lane (512 B per wave instruction) -- calibrated below
    # against the known byte count (the kernel reads E once: B*cs*K*8 bytes).
    cs, K = CS, 4096
    # trials per launch: 64 on one stream, 32 in the default two-stream schedule -- read it off the dispatch's grid (K/256 x B workgroups of 256)
    B = 64
    try:
        tr = list(csv.DictReader(open(glob.glob(os.path.join(root, "*", "pmc_counter_collection.csv"))[0])))
        g = [int(x["Grid_Size"]) for x in tr if MATCH in x["Kernel_Name"] and x.get("Grid_Size")]
        if g:
            B = max(1, round(sum(g) / len(g) / (K * CARS)))          # one lane per (sample, car) (multi-car waves leave 64 % CARS lanes idle: pass the trial count instead)
    except Exception:
        pass
    if len(sys.argv) > 4:
        B = int(sys.argv[4])                                         # trials per launch, stated by the caller
    known_read = B * cs * K * 8
    fetch_kib, write_kib = r.get("FETCH_SIZE") or 0.0, r.get("WRITE_SIZE") or 0.0
    sha = hashlib.sha256()
    for f in ("mpopis_amd/csrc/kernels_rollout.hip", "mpopis_amd/csrc/car_dynamics.h"):      # = bench.py PMC_SOURCES: ties this file to the kernel sources it measured
        sha.update(open(f, "rb").read())
    out = {"source_sha": sha.hexdigest()[:16], "kernel": r["kernel"], "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
           "known_read_bytes_per_launch": known_read, "fetch_raw_bytes": fetch_kib * 1024,
           "fetch_calibration_ratio_known_over_raw": known_read / (fetch_kib * 1024) if fetch_kib else None,
           "trials_per_launch": B, "hbm_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
           "hbm_bytes_per_rollout": (2 * fetch_kib * 1024 + write_kib * 1024) / (B * K),
           "valu_insts_per_rollout": (r.get("SQ_INSTS_VALU") or 0.0) / (B * K / 64.0),
           "fp64_valu_insts_per_rollout": sum((r.get(c) or 0.0) for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64")) / (B * K / 64.0),
           "fp64_flops_per_rollout": 64.0 * (2 * (r.get("SQ_INSTS_VALU_FMA_F64") or 0.0) + sum((r.get(c) or 0.0) for c in ("SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"))) / (B * K),
           # SQ_ACTIVE_INST_VALU counts quad-cycles summed over all SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
           "valu_busy_frac": ((r.get("SQ_ACTIVE_INST_VALU") or 0.0) * 4.0) / (((r.get("GRBM_GUI_ACTIVE") or 1.0) / 8.0) * 1024.0),
           "effective_clock_ghz": ((r.get("GRBM_GUI_ACTIVE") or 0.0) / 8.0) / (r["avg_us"] * 1e3),
           "note": "hbm_bytes_per_launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE x2 correction per MI355X_MICROARCH.md §HBM); %d trials per launch, K=4096, cs=%d" % (B, CS)}
    if r.get("SQ_VALU_MFMA_BUSY_CYCLES") not in (None, ""):
        out["mfma_busy_cycles"] = r["SQ_VALU_MFMA_BUSY_CYCLES"]
    out["avg_us_under_pmc"] = r["avg_us"]
    json.dump(out, open(OUTJ, "w"), indent=1)
    print(out)
