cd $GRAFT_REPO_ROOT
python scripts/bench_infer.py 2>/dev/null
python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --no-prof 2>/dev/null
python bench.py --workload c5 --steps 20 --warmup 5 --no-cpu-baseline --no-prof 2>/dev/null
