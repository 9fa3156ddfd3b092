mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_chan_attn_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -15 > gpurun_out/se_tests.log
rm -f gpurun_out/se_bench.jsonl
for o in "se_single=1" "se_single=0"; do
 for nt in 3 1; do
  timeout 300 python bench.py --no-cpu --only SELayer --steps 20 --warmup 5 --opt $o --nt $nt 2>> gpurun_out/se_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o nt=$nt', d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/se_bench.jsonl
 done
done
timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 > gpurun_out/se_c2.json 2>>gpurun_out/se_bench.err
cat gpurun_out/se_tests.log gpurun_out/se_bench.jsonl gpurun_out/se_c2.json; tail -3 gpurun_out/se_bench.err
