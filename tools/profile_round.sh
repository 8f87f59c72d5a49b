#!/bin/bash
# Evidence for one round, run ON THE GPU BOX (gpurun): bench line, rocprofv3 kernel-trace stats of the same command,
# and the PMC passes (each in its own run; never combined with sys/hip/hsa tracing).  Outputs under gpurun_out/<tag>/.
#   usage: tools/profile_round.sh <tag>        e.g.  gpurun --timeout 1500 -- 'bash tools/profile_round.sh r01f'
set -u
TAG=${1:-r01}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$TAG
mkdir -p "$O"
# the driver's command (BENCH_rNN.json: cmd).  stdout = the ONE compact line (< 4 KB); the full record goes to stderr (BENCH_DETAIL ...) and gpurun_out/bench_detail_latest.json
cd "$R" && python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_line.json" 2> "$O/bench.err"; wc -c "$O/bench_line.json"; tail -c 4200 "$O/bench_line.json"
cp "$R/gpurun_out/bench_detail_latest.json" "$O/bench_detail.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
# per-kernel durations need the chip to themselves: the profiled passes pin the schedule to ONE stream (MPOPIS_NSPLIT=1 = mpopis_set_overlap(h, 1),
# what bench.py's own one-stream pass does for roofline.frac); the default schedule for this shape is four part-chains that time-share the chip.
export MPOPIS_NSPLIT=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-midlap --repeats 0 > "$O/prof_bench.log" 2>&1
head -8 "$O"/prof/bench_kernel_stats.csv | cut -c1-160
# ... and the default schedule once, for the record (durations include time-sharing)
( unset MPOPIS_NSPLIT; rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_default" -o bench -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-midlap --repeats 0 > "$O/prof_bench_default.log" 2>&1 )
rm -f "$O"/prof_default/*kernel_trace.csv
for P in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" \
         "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"; do
    N=$(echo $P | cut -d" " -f1)
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$O/pmc/$N" -o pmc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-midlap --repeats 0 > "$O/pmc_$N.log" 2>&1
done
ls "$O"/pmc/*/ | head -20
# summaries on the box (gpurun merges at most 64 MiB back): per-kernel PMC table + pmc_rollout.json next to the raw passes, then drop the per-dispatch dumps
cd "$R" && python tools/pmc_summary.py "$O/pmc" "$TAG" > "$O/pmc_summary.log" 2>&1
cp "$R"/profiles/${TAG}_pmc_summary.csv "$R"/profiles/pmc_rollout.json "$O"/ 2>/dev/null
rm -f "$O"/pmc/*/pmc_counter_collection.csv "$O"/pmc/*/pmc_kernel_trace.csv "$O"/prof/*kernel_trace.csv

# ---- C4 (3-car :cmamppi K=4096 H=50 N=10) at 64 trials: the second K=4096/H=50 config -- kernel stats on ONE stream + the PMC passes of its rollout kernel ----
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/c4_b64" -o c4 -- python "$R/tools/prof_c4.py" 64 > "$O/c4_b64.log" 2>&1
head -12 "$O"/c4_b64/c4_kernel_stats.csv | cut -c1-170; rm -f "$O"/c4_b64/*kernel_trace.csv
for P in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" \
         "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    N=$(echo $P | cut -d" " -f1)
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$O/pmc_c4/$N" -o pmc -- python "$R/tools/prof_c4.py" 64 3 > "$O/pmc_c4_$N.log" 2>&1
done
cd "$R" && python tools/pmc_summary.py "$O/pmc_c4" "${TAG}_c4_b64" c4 64 > "$O/pmc_c4_summary.log" 2>&1
cp "$R"/profiles/${TAG}_c4_b64_pmc_summary.csv "$R"/profiles/pmc_rollout_3car.json "$O"/ 2>/dev/null
rm -f "$O"/pmc_c4/*/pmc_counter_collection.csv "$O"/pmc_c4/*/pmc_kernel_trace.csv
