cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-prof"
$B > gpurun_out/ab1.json
$B > gpurun_out/ab2.json
grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab[1-2].json
python bench.py --steps 10 --warmup 5 --no-cpu-baseline --kernel-table gpurun_out/kt_k.json > /dev/null 2>&1
