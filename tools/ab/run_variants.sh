#!/bin/bash
# builds variants of ONE translation unit and runs a python script with each, alternating, on one box:
#   tools/ab/run_variants.sh <file.hip> "<flags of variant 1>" "<flags of variant 2>" ... -- <script.py> [args]      (REPS=2 repetitions)
cd "$(dirname "$0")/../.."
SRC=$1; shift
i=0; tags=()
while [ "$1" != "--" ]; do tools/ab/build_variant.sh V$i $SRC $1 > /dev/null || exit 1; echo "V$i = $SRC $1"; tags+=(V$i); i=$((i+1)); shift; done
shift
for rep in $(seq ${REPS:-2}); do for t in "${tags[@]}"; do echo -n "$t: "; MPOPIS_HIP_LIB=$PWD/tools/ab/lib$t.so python "$@" 2>&1 | tail -1; done; done
rm -f tools/ab/libV*.so
