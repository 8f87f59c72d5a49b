#!/bin/bash
# dev: which path the rollout kernels take (general sub-step, general nearest-point search) at the reset state and at mid-lap states.
# Builds a variant library with -DMPOPIS_PATH_STATS beside the real one (tools/ab/libstats.so) and runs tools/path_stats.py with it.
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DMPOPIS_PATH_STATS -c mpopis_amd/csrc/kernels_rollout.hip -Iinclude -Impopis_amd/csrc -o /tmp/kr_stats.o || exit 1
objs=$(ls mpopis_amd/lib/obj/*.o | grep -v kernels_rollout.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libstats.so $objs /tmp/kr_stats.o -ldl
echo "built tools/ab/libstats.so; on the GPU box:  MPOPIS_HIP_LIB=\$PWD/tools/ab/libstats.so python tools/path_stats.py [trials policy K N cars]"
[ -n "$RUN" ] && MPOPIS_HIP_LIB=$PWD/tools/ab/libstats.so python tools/path_stats.py "$@"
