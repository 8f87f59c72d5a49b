#!/bin/bash
# builds tools/kbench_c5_bin and the two private variants (drawing / staging compiled out of kernels_mfma.hip)
cd "$(dirname "$0")/.."
H=/opt/rocm/bin/hipcc; F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-function -Wno-unused-result"
O=mpopis_amd/lib/obj
$H $F -c tools/kbench_c5.hip -o /tmp/kbc5.o 2>&1 | grep -E "error"
$H --offload-arch=gfx950 /tmp/kbc5.o $O/kernels_mfma.o $O/kernels_sample.o $O/kernels_linalg.o -o tools/kbench_c5_bin 2>&1 | grep -E "error|undefined"
for V in nodraw:-DMPOPIS_DEV_NO_DRAW nostage:-DMPOPIS_DEV_NO_STAGE; do
    T=${V%%:*}; D=${V##*:}
    $H $F $D -c mpopis_amd/csrc/kernels_mfma.hip -o /tmp/km_$T.o 2>&1 | grep -E " error"
    $H --offload-arch=gfx950 /tmp/kbc5.o /tmp/km_$T.o $O/kernels_sample.o $O/kernels_linalg.o -o tools/kbench_c5_${T}_bin 2>&1 | grep -E "error|undefined"
done
ls tools/kbench_c5*bin
