#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -8 > gpurun_out/test_all.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
rm -f gpurun_out/bench_others.jsonl
for w in c3 c4 mixer da; do
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --cpu-sample 16 >> gpurun_out/bench_others.jsonl 2>> gpurun_out/bench_others.err
done
timeout 400 python bench.py --workload c5 --steps 3 --warmup 1 --cpu-sample 8 >> gpurun_out/bench_others.jsonl 2>> gpurun_out/bench_others.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in c2 c3 c4 c5; do
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o $w -- python $R/bench.py --no-cpu --workload $w --steps 3 --warmup 1 > $R/gpurun_out/prof_$w.log 2>&1
done
cd $R
