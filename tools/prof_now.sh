mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_all -o all -- python $R/bench.py --no-cpu --no-strict --steps 3 --warmup 1 > $R/gpurun_out/now_prof_all.log 2>&1
python $R/tools/rocpd_stats.py $R/gpurun_out/prof_all/all_results.db > $R/gpurun_out/now_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/prof_all
