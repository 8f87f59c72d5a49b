#!/bin/bash
# like run_py.sh for three builds: tools/ab/lib{A,B,C}.so
VARIANTS="A B C" exec "$(dirname "$0")/run_py.sh" "$@"
