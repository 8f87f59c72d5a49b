#!/bin/bash
# A/B two builds of libmpopis_hip.so on the SAME GPU box (boxes differ by +-3%): tools/ab/libA.so vs tools/ab/libB.so.
# The variant is selected with MPOPIS_HIP_LIB (mpopis_amd/_lib.py); the product binary mpopis_amd/lib/libmpopis_hip.so is never touched.
cd "$(dirname "$0")/../.."
for rep in 1 2; do for v in A B; do echo -n "$v: "; MPOPIS_HIP_LIB=$PWD/tools/ab/lib$v.so python tools/quick_bench.py ${1:-c5} 2>&1 | tail -2 | tr '\n' ' '; echo; done; done
