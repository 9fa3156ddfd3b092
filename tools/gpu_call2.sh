#!/bin/bash
# second GPU call: full GPU test suite, bandwidth yardsticks, c2 option sweep, other workloads, profiles
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x --deselect tests/test_chan_attn_gpu.py::test_full_size_properties 2>&1 | tail -60 > gpurun_out/test_all.log
timeout 300 python tools/mem_bw.py > gpurun_out/mem_bw.log 2>&1
for nt in 0 1 2 3; do for rev in 0 1; do for ch in 0 64 128; do
  timeout 200 python bench.py --no-cpu --steps 10 --warmup 3 --nt $nt --reverse $rev --chunk-images $ch >> gpurun_out/c2_sweep.jsonl 2>> gpurun_out/c2_sweep.err
done; done; done
for w in c3 c4 mixer da; do
  timeout 300 python bench.py --no-cpu --workload $w --steps 5 --warmup 2 >> gpurun_out/bench_others.jsonl 2>> gpurun_out/bench_others.err
done
timeout 300 python bench.py --no-cpu --workload c3 --steps 5 --warmup 2 --precision 0 >> gpurun_out/bench_others.jsonl 2>> gpurun_out/bench_others.err
timeout 300 python bench.py --no-cpu --workload c3 --steps 5 --warmup 2 --precision 2 >> gpurun_out/bench_others.jsonl 2>> gpurun_out/bench_others.err
timeout 400 python bench.py --no-cpu --workload c5 --steps 3 --warmup 1 >> gpurun_out/bench_others.jsonl 2>> gpurun_out/bench_others.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload c3 --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c4 -o c4 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload c4 --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_c4.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la gpurun_out > gpurun_out/ls.txt
