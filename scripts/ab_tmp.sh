cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-prof"
$B > gpurun_out/ab0.json
VQ2_WGRAD_STREAM=1 $B > gpurun_out/ab1.json
VQ2_WGRAD_STREAM=1 VQ2_BWD_LDS_FLOOR=84000 $B > gpurun_out/ab2.json
VQ2_BWD_LDS_FLOOR=84000 $B > gpurun_out/ab3.json
VQ2_WGRAD_STREAM=1 VQ2_BWD_LDS_FLOOR=56000 $B > gpurun_out/ab4.json
grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab[0-4].json
