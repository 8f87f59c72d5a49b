#!/bin/bash
# dev: tools/ab/lib<TAG>.so = the current objects with ONE translation unit rebuilt with extra flags.
#   usage: tools/ab/build_variant.sh <TAG> <file.hip> [extra hipcc flags...]      e.g.  tools/ab/build_variant.sh A kernels_rollout.hip -DMPOPIS_PARAMS_IN_LDS=0
cd "$(dirname "$0")/../.."
TAG=$1; SRC=$2; shift 2
base=$(basename "$SRC" .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast "$@" -c mpopis_amd/csrc/$SRC -Iinclude -Impopis_amd/csrc -o /tmp/ab_${TAG}_$base.o || exit 1
objs=$(ls mpopis_amd/lib/obj/*.o | grep -v "/$base.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/lib$TAG.so $objs /tmp/ab_${TAG}_$base.o -ldl && echo "tools/ab/lib$TAG.so"
