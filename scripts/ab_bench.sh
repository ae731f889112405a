#!/bin/bash
# Same-box A/B of the whole train step: A = libvq2.so, B = libvq2_<tag>.so, interleaved rounds.
#   bash scripts/ab_bench.sh prev 3 [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; ROUNDS=${2:-3}; shift; shift
for r in $(seq $ROUNDS); do
  for v in A B; do
    if [ $v = B ]; then export VQ2_LIB=$ROOT/vq-vae-2-pytorch_amd/libvq2_$TAG.so; else unset VQ2_LIB; fi
    python3 $ROOT/bench.py --no-cpu-baseline --no-prof --steps 60 --warmup 15 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v round $r:', d['ms_per_step'], 'ms', d['value'], 'img/s')"
  done
done
