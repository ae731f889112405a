ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do
for v in "VQ2_WGRAD_STREAM_MAXPIX=40000" "VQ2_WGRAD_STREAM_MAXPIX=0" "VQ2_WGRAD_STREAM_MAXPIX=140000" "VQ2_WGRAD_STREAM=0" "VQ2_STATS_STREAM=0"; do
  env $v python3 $ROOT/bench.py --no-cpu-baseline --no-prof --steps 60 --warmup 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v round $r:', d['ms_per_step'], 'ms', d['value'], 'img/s')"
done; done
