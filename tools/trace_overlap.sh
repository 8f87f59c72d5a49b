#!/bin/bash
# kernel timeline of a few bench steps (two-stream schedule): which kernels actually run concurrently?
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/${1:-trace}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/prof" -o tr -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$O/prof.log" 2>&1
ls -la "$O/prof"
python - "$O/prof/tr_kernel_trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "mpopis" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 140 kernels of the run = inside the instrumented/timed passes
sel = rows[len(rows)//3: len(rows)//3 + 90]
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%10.1f %10.1f %8.1f q=%s %s grid=%s" % (s/1e3, e/1e3, (e-s)/1e3, r.get("Queue_Id"), r["Kernel_Name"][:42].replace("mpopis::", ""), r.get("Grid_Size")))
PY
