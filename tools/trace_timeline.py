"""Timeline of one AIS iteration from a rocprofv3 --kernel-trace csv: kernels in start order with start offset, duration and queue (dev tool).
usage: python tools/trace_timeline.py <kernel_trace.csv> [anchor kernel substring = k_rollout] [which occurrence = 30] [how many kernels = 40]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_rollout"
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 30
count = int(sys.argv[4]) if len(sys.argv) > 4 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0 = idx[min(occ, len(idx) - 1)]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
for r in rows[i0:i0 + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("mpopis::", "").split("(")[0]
    print("%9.1f us  +%7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name[:60]))
