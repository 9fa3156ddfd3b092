#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_chan_attn_gpu.py tests/test_ops_gpu.py -m gpu -q --tb=short -x -k "single_pass or edge or chunk or sdpa16 or full_size" 2>&1 | tail -15 > gpurun_out/test_fused.log
rm -f gpurun_out/c2_fused.jsonl
timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 >> gpurun_out/c2_fused.jsonl 2>> gpurun_out/c2_fused.err
timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 >> gpurun_out/c2_fused.jsonl 2>> gpurun_out/c2_fused.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c2 -o c2 -- python $R/bench.py --no-cpu --steps 10 --warmup 3 > $R/gpurun_out/prof_c2.log 2>&1
cd $R
