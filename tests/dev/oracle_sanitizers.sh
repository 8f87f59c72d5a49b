#!/bin/bash
# dev: the CPU oracle (the checker every parity statement rests on) under AddressSanitizer + UBSan: the CPU oracle tests and a sweep of policy calls.
#   usage: bash tests/dev/oracle_sanitizers.sh          (no GPU needed; ~1 min)
R=$(cd "$(dirname "$0")/../.." && pwd)
gcc -O1 -g -fsanitize=address,undefined -fopenmp -std=gnu11 -ffp-contract=off -shared -fPIC -o /tmp/libmpopis_oracle_asan.so "$R/oracle/mpopis_oracle.c" -lm || exit 1
export MPOPIS_ORACLE_LIB=/tmp/libmpopis_oracle_asan.so ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD=$(gcc -print-file-name=libasan.so)
cd "$R" && python -m pytest tests/test_oracle_kat.py tests/test_golden.py tests/test_third_party_blocks.py tests/test_dynamics_shim.py -x -q -m "not gpu" 2>&1 | tail -2
python tests/dev/oracle_sanitizer_sweep.py 2>&1 | tail -3
