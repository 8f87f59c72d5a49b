#!/bin/bash
# builds tools/kbench_linalg_bin against the current object files (run `python -m mpopis_amd.build` first)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -c tools/kbench_linalg.hip -o /tmp/kbl.o 2>&1 | grep -E "error"
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/kbl.o mpopis_amd/lib/obj/kernels_linalg.o mpopis_amd/lib/obj/kernels_invsqrt.o -o tools/kbench_linalg_bin 2>&1 | grep -E "error|undefined"
ls -la tools/kbench_linalg_bin
