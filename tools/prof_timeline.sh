#!/bin/bash
# kernel-trace timeline of one iteration of an arbitrary python command:  tools/prof_timeline.sh <tag> <anchor kernel> <occurrence> <count> <python args...>
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; ANCH=$2; OCC=$3; CNT=$4; shift 4
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
( cd "$R" && rocprofv3 --kernel-trace --output-format csv -d "$O/prof" -o p -- python "$@" > "$O/prof.log" 2>&1 )
python "$R/tools/trace_timeline.py" "$O/prof/p_kernel_trace.csv" "$ANCH" "$OCC" "$CNT" | tee "$O/timeline.txt"
rm -f "$O"/prof/*kernel_trace.csv
