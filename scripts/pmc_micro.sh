cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_k4_f -- python3 $R/scripts/microbench.py c4s2_64_128 t_128_64 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_k4_w -- python3 $R/scripts/microbench.py c4s2_64_128 t_128_64 > /dev/null 2>&1
python3 $R/scripts/pmc_summary.py $R/gpurun_out/k4_pmc.json $R/gpurun_out/pmc_k4_f $R/gpurun_out/pmc_k4_w
rm -rf $R/gpurun_out/pmc_k4_f $R/gpurun_out/pmc_k4_w
