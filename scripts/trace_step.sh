#!/bin/bash
# rocprofv3 --kernel-trace of a short bench run; leaves the per-dispatch CSV in gpurun_out/<tag>_kernel_trace.csv
# (input of scripts/stream_gaps.py).   bash scripts/trace_step.sh <tag> [bench args...]
set -uo pipefail
TAG=${1:-trace}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python3 $ROOT/bench.py --no-cpu-baseline --no-prof --steps 12 --warmup 4 "$@" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_trace.err
find $OUT/prof_$TAG -name "*kernel_trace.csv" -exec cp {} $OUT/${TAG}_kernel_trace.csv \;
find $OUT/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
rm -rf $OUT/prof_$TAG
head -c 300 $OUT/${TAG}_bench.json; echo; wc -l $OUT/${TAG}_kernel_trace.csv
