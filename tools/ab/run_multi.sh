#!/bin/bash
# compare several builds on one box: tools/ab/run_multi.sh A O2 Os ...   (files tools/ab/lib<name>.so or lib_<name>.so; selected with MPOPIS_HIP_LIB)
cd "$(dirname "$0")/../.."
for rep in 1 2; do for v in "$@"; do f=tools/ab/lib$v.so; [ -f $f ] || f=tools/ab/lib_$v.so; echo -n "$v: "; MPOPIS_HIP_LIB=$PWD/$f python tools/quick_bench.py c5 2>&1 | tail -2 | tr '\n' ' ' | grep -o "B=64.*" | cut -c1-40,90-140; done; done
