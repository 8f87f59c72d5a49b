#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python command, filtered:  tools/kstat.sh <pattern> <python args...>   (env passes through)
R=$(cd "$(dirname "$0")/.." && pwd)
PAT=$1; shift
O=$(mktemp -d /tmp/kstat.XXXX)
cd /tmp && export TMPDIR=/tmp
( cd "$R" && rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o p -- python "$@" > "$O/log" 2>&1 )
python - "$O/p_kernel_stats.csv" "$PAT" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print("%-60s calls %5s avg %9.1f us min %9.1f max %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf "$O"
