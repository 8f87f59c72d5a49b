#!/bin/bash
# compare several builds on one box: tools/ab/run_multi.sh A O2 Os ...   (files tools/ab/lib<name>.so or lib_<name>.so)
cd "$(dirname "$0")/../.."
cp mpopis_amd/lib/libmpopis_hip.so /tmp/lib_cur.so
for rep in 1 2; do for v in "$@"; do f=tools/ab/lib$v.so; [ -f $f ] || f=tools/ab/lib_$v.so; cp $f mpopis_amd/lib/libmpopis_hip.so; echo -n "$v: "; python tools/quick_bench.py c5 2>&1 | tail -2 | tr '\n' ' ' | grep -o "B=64.*" | cut -c1-40,90-140; done; done
cp /tmp/lib_cur.so mpopis_amd/lib/libmpopis_hip.so
