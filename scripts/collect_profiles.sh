#!/bin/bash
# Collect the evidence files of a round on the GPU box (run through gpurun):
#   bash scripts/collect_profiles.sh r02 [tag]
# For each workload (c2 = BASELINE configs[1], c4 = configs[3], c5 = configs[4]): bench line, per-shape HIP-event
# kernel table, rocprofv3 --kernel-trace --stats CSV, and PMC passes (FETCH_SIZE; WRITE_SIZE + L2 hit; SQ busy /
# MFMA counters) folded by scripts/pmc_summary.py.  Everything lands in gpurun_out/<round>_<workload>_*.
set -uo pipefail
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in ${WORKLOADS:-c2 c4 c5}; do
  B="python3 $ROOT/bench.py --workload $w --no-cpu-baseline"
  if [ $w = c2 ]; then   # the headline line carries the CPU baseline too
    python3 $ROOT/bench.py --workload $w --steps 30 --warmup 10 > $OUT/${R}_${w}_bench.json 2> /dev/null
  else
    $B --steps 30 --warmup 10 > $OUT/${R}_${w}_bench.json 2> /dev/null
  fi
  $B --steps 8 --warmup 3 --kernel-table $OUT/${R}_${w}_kernel_table_hip_events.json > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $R -- $B --steps 10 --warmup 3 --no-prof > /dev/null 2>&1
  cp $OUT/prof_$w/${R}_kernel_stats.csv $OUT/${R}_${w}_rocprofv3_kernel_stats.csv
  P="$B --steps 4 --warmup 2 --no-prof"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${w}_f -- $P > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_${w}_w -- $P > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${w}_s -- $P > /dev/null 2>&1
  python3 $ROOT/scripts/pmc_summary.py $OUT/${R}_${w}_pmc_summary.json $OUT/pmc_${w}_f $OUT/pmc_${w}_w $OUT/pmc_${w}_s
  if [ $w = c2 ]; then   # main-stream idle time and the per-kernel table of the SAME trace (scripts/stream_gaps.py)
    find $OUT/prof_$w -name "*kernel_trace.csv" -exec cp {} $OUT/${R}_${w}_kernel_trace.csv \;
    python3 $ROOT/scripts/stream_gaps.py $OUT/${R}_${w}_kernel_trace.csv $OUT/${R}_${w}_stream_gaps.json > /dev/null
    python3 $ROOT/scripts/trace_table.py $OUT/${R}_${w}_kernel_trace.csv $OUT/${R}_${w}_trace_table.json > /dev/null
  fi
  rm -rf $OUT/pmc_${w}_f $OUT/pmc_${w}_w $OUT/pmc_${w}_s $OUT/prof_$w
  echo "$w done: $(grep -o '"value": [0-9.]*' $OUT/${R}_${w}_bench.json | head -1)"
done
# next-row f4: the default VQVAE_Deep through the drop-in module (bench line + per-shape table)
python3 $ROOT/bench.py --workload deep --steps 20 --warmup 5 --no-prof > $OUT/${R}_deep_bench.json 2> /dev/null
python3 $ROOT/bench.py --workload deep --steps 8 --warmup 3 --kernel-table $OUT/${R}_deep_kernel_table_hip_events.json > /dev/null 2>&1
echo "deep done: $(grep -o '"value": [0-9.]*' $OUT/${R}_deep_bench.json | head -1)"
