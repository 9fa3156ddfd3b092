#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_bench.log
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_gpu_parity.py -m gpu -q --tb=short -x 2>&1 | tail -8 > gpurun_out/test_ops.log
rm -f gpurun_out/bench_others.jsonl
for w in c3 c5 mixer; do
  timeout 300 python bench.py --no-cpu --workload $w --steps 5 --warmup 2 >> gpurun_out/bench_others.jsonl 2>> gpurun_out/bench_others.err
done
