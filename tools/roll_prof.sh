#!/bin/bash
# dev: per-wave start / end times of the 1-car rollout kernel (C5, 64 trials).  Builds a variant library with -DMPOPIS_ROLL_PROF beside the real one.
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DMPOPIS_ROLL_PROF $ROLL_PROF_FLAGS -c mpopis_amd/csrc/kernels_rollout.hip -Iinclude -Impopis_amd/csrc -o /tmp/kr_prof.o || exit 1
objs=$(ls mpopis_amd/lib/obj/*.o | grep -v kernels_rollout.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libprof${ROLL_PROF_TAG}.so $objs /tmp/kr_prof.o -ldl
