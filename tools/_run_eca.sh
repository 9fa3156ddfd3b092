mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_chan_attn_gpu.py tests/test_ops_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -15 > gpurun_out/eca_tests.log
rm -f gpurun_out/eca_bench.jsonl
for o in "eca_single=1" "eca_single=0"; do
 for nt in 3 2 1 0; do
  timeout 300 python bench.py --no-cpu --only ECA --steps 20 --warmup 5 --opt $o --nt $nt 2>> gpurun_out/eca_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o nt=$nt', d['ms_per_step'], d['roofline'])" >> gpurun_out/eca_bench.jsonl
 done
done
cat gpurun_out/eca_tests.log gpurun_out/eca_bench.jsonl; tail -3 gpurun_out/eca_bench.err
