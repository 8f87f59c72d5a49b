#!/bin/bash
# profiling build of the linalg micro-bench: kernels_linalg.hip with -DPOTRF_PROF (per-phase timestamps of the cooperative Cholesky)
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -DPOTRF_PROF -DLAN_PROF -Iinclude -Impopis_amd/csrc"
/opt/rocm/bin/hipcc $F -c mpopis_amd/csrc/kernels_linalg.hip -o /tmp/kl_prof.o 2>&1 | grep -E "error"
/opt/rocm/bin/hipcc $F -c tools/kbench_linalg.hip -o /tmp/kbl_prof.o 2>&1 | grep -E "error"
/opt/rocm/bin/hipcc $F -c mpopis_amd/csrc/kernels_invsqrt.hip -o /tmp/ki_prof.o 2>&1 | grep -E "error"
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/kbl_prof.o /tmp/kl_prof.o /tmp/ki_prof.o -o tools/kbench_linalg_prof_bin 2>&1 | grep -E "error|undefined"
ls -la tools/kbench_linalg_prof_bin
