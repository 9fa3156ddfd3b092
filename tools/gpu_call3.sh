#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x --deselect tests/test_chan_attn_gpu.py::test_full_size_properties 2>&1 | tail -40 > gpurun_out/test_all.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
rm -f gpurun_out/bench_others.jsonl
for w in c3 c4 mixer da; do
  timeout 300 python bench.py --no-cpu --workload $w --steps 5 --warmup 2 >> gpurun_out/bench_others.jsonl 2>> gpurun_out/bench_others.err
done
timeout 400 python bench.py --no-cpu --workload c5 --steps 3 --warmup 1 >> gpurun_out/bench_others.jsonl 2>> gpurun_out/bench_others.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c2 -o c2 -- python $R/bench.py --no-cpu --steps 10 --warmup 3 > $R/gpurun_out/prof_c2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c3 -o c3 -- python $R/bench.py --no-cpu --workload c3 --steps 5 --warmup 2 > $R/gpurun_out/prof_c3.log 2>&1

timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_c2 -o c2 -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > $R/gpurun_out/pmc_fetch_c2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_c2 -o c2 -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > $R/gpurun_out/pmc_write_c2.log 2>&1
cd $R
ls -la gpurun_out > gpurun_out/ls.txt
