#!/bin/bash
# A/B two builds of libmpopis_hip.so on the SAME GPU box: runs `python <script> <args>` with tools/ab/libA.so, then libB.so, twice (MPOPIS_HIP_LIB selects the build)
cd "$(dirname "$0")/../.."
for rep in 1 2; do for v in ${VARIANTS:-A B}; do echo "== $v"; MPOPIS_HIP_LIB=$PWD/tools/ab/lib$v.so python "$@" 2>&1 | tail -${TAILN:-4}; done; done
