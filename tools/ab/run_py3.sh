#!/bin/bash
# like run_py.sh for three builds: tools/ab/lib{A,B,C}.so
cd "$(dirname "$0")/../.."
cp mpopis_amd/lib/libmpopis_hip.so /tmp/lib_cur.so
for rep in 1 2; do for v in A B C; do cp tools/ab/lib$v.so mpopis_amd/lib/libmpopis_hip.so; echo "== $v"; python "$@" 2>&1 | tail -${TAILN:-4}; done; done
cp /tmp/lib_cur.so mpopis_amd/lib/libmpopis_hip.so
