#!/bin/bash
# A/B two builds of libmpopis_hip.so on the SAME GPU box: runs `python <script> <args>` with tools/ab/libA.so, then libB.so, twice
cd "$(dirname "$0")/../.."
cp mpopis_amd/lib/libmpopis_hip.so /tmp/lib_cur.so
for rep in 1 2; do for v in A B; do cp tools/ab/lib$v.so mpopis_amd/lib/libmpopis_hip.so; echo "== $v"; python "$@" 2>&1 | tail -${TAILN:-4}; done; done
cp /tmp/lib_cur.so mpopis_amd/lib/libmpopis_hip.so
