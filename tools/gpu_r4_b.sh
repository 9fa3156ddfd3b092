#!/bin/bash
# Round-4 call B: new LN-fold tests + the files touched by the robustness fixes, then ViT-Base / C3 timing with the fold on and off.
mkdir -p gpurun_out/r4b
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_ln_fold_gpu.py -x -q -s > gpurun_out/r4b/t_fold.log 2>&1
echo "fold rc=$?" >> gpurun_out/r4b/t_fold.log
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_ln_fold_gpu.py > gpurun_out/r4b/t_all.log 2>&1
echo "all rc=$?" >> gpurun_out/r4b/t_all.log
timeout 300 python bench.py --workload c5 --no-cpu --no-strict --steps 10 --warmup 3 > gpurun_out/r4b/c5_fold.json 2> gpurun_out/r4b/c5_fold.err
timeout 300 python bench.py --workload c5 --no-cpu --no-strict --steps 10 --warmup 3 --opt ln_fold=0 > gpurun_out/r4b/c5_nofold.json 2> gpurun_out/r4b/c5_nofold.err
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/r4b/p -o k -- python $R/bench.py --workload c5 --no-cpu --no-strict --steps 3 --warmup 2 > $R/gpurun_out/r4b/prof.log 2>&1
python $R/tools/rocpd_stats.py $R/gpurun_out/r4b/p/k_results.db > $R/gpurun_out/r4b/c5_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/r4b/p
