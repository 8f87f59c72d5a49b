#!/bin/bash
# A/B two builds of libmpopis_hip.so on the SAME GPU box (boxes differ by +-3%): tools/ab/libA.so vs tools/ab/libB.so
cd "$(dirname "$0")/../.."
cp mpopis_amd/lib/libmpopis_hip.so /tmp/lib_cur.so
for rep in 1 2; do for v in A B; do cp tools/ab/lib$v.so mpopis_amd/lib/libmpopis_hip.so; echo -n "$v: "; python tools/quick_bench.py ${1:-c5} 2>&1 | tail -2 | tr '\n' ' '; echo; done; done
cp /tmp/lib_cur.so mpopis_amd/lib/libmpopis_hip.so
