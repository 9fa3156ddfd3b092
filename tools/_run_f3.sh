mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -k "conv2d or token_mean or patch_embed" 2>&1 | tail -25 > gpurun_out/f3_ops.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "full" 2>&1 | tail -25 > gpurun_out/f3_full.log
rm -f gpurun_out/f3_bench.jsonl
for w in cswin mixer_full; do
  timeout 300 python bench.py --no-cpu --workload $w --steps 5 --warmup 2 >> gpurun_out/f3_bench.jsonl 2>> gpurun_out/f3_bench.err
done
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for w in cswin mixer_full; do
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o $w -- python $R/bench.py --no-cpu --workload $w --steps 3 --warmup 1 > $R/gpurun_out/prof_$w.log 2>&1
done
cd $R
cat gpurun_out/f3_ops.log gpurun_out/f3_full.log gpurun_out/f3_bench.jsonl; tail -5 gpurun_out/f3_bench.err
