#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python command:  tools/prof_any.sh <tag> <python args...>
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
( cd "$R" && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o p -- python "$@" > "$O/prof.log" 2>&1 )
tail -2 "$O/prof.log"
head -12 "$O/prof/p_kernel_stats.csv" | cut -c1-150
