#!/bin/bash
# Same-box A/B of the train step between two ENVIRONMENT settings: bash scripts/ab_env.sh "VQ2_X=0" "VQ2_X=1" [rounds] [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; B=$2; ROUNDS=${3:-3}; shift; shift; shift
for r in $(seq $ROUNDS); do
  for v in "$A" "$B"; do
    env $v python3 $ROOT/bench.py --no-cpu-baseline --no-prof --steps 60 --warmup 15 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v round $r:', d['ms_per_step'], 'ms', d['value'], 'img/s')"
  done
done
