#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "sdpa or lepe or window or vit or cswin or head_index" 2>&1 | tail -15 > gpurun_out/test_attn.log
rm -f gpurun_out/bench_others.jsonl
for w in c3 c4; do
  timeout 300 python bench.py --no-cpu --workload $w --steps 5 --warmup 2 >> gpurun_out/bench_others.jsonl 2>> gpurun_out/bench_others.err
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c3 -o c3 -- python $R/bench.py --no-cpu --workload c3 --steps 5 --warmup 2 > $R/gpurun_out/prof_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c4 -o c4 -- python $R/bench.py --no-cpu --workload c4 --steps 3 --warmup 1 > $R/gpurun_out/prof_c4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_sq_c3 -o c3 -- python $R/bench.py --no-cpu --workload c3 --steps 2 --warmup 1 > $R/gpurun_out/pmc_sq_c3.log 2>&1
cd $R
