for cfg in "1 300" "8 300" "64 300" "2 145" "4 200" "3 129" "2 304"; do
  set -- $cfg
  echo "=== B=$1 n=$2"
  for mode in "MPOPIS_POTRF_REG=1" "MPOPIS_POTRF_REG=0" "MPOPIS_POTRF_REG=0 MPOPIS_POTRF_G=0"; do
    echo "-- $mode"; env $mode timeout 120 tools/kbench_linalg_bin $1 $2 2>&1 | grep -E "^potrf|potrf status|potrf max|L hash"
  done
done
