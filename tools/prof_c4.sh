#!/bin/bash
# rocprofv3 kernel stats of the C4 closed loop (3-car :cmamppi K=4096 H=50, B trials (default 8), 11 MPC steps) -> gpurun_out/<tag>/
#   usage: bash tools/prof_c4.sh <tag> [B]
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/${1:-c4}
B=${2:-8}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o c4 -- python "$R/tools/prof_c4.py" $B > "$O/prof.log" 2>&1
tail -2 "$O/prof.log"
head -14 "$O/prof/c4_kernel_stats.csv" | cut -c1-150; python "$R/tools/trace_timeline.py" "$O/prof/c4_kernel_trace.csv" k_rollout 30 36 > "$O/timeline.txt" 2>&1; rm -f "$O"/prof/*kernel_trace.csv
