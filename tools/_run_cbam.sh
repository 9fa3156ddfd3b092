mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_chan_attn_gpu.py -m gpu -q --tb=short -x -k "cbam_single" 2>&1 | tail -15 > gpurun_out/cbam_tests.log
rm -f gpurun_out/cbam_bench.jsonl
for o in "cbam_single=1" "cbam_single=0"; do
  timeout 300 python bench.py --no-cpu --only CBAM --steps 20 --warmup 5 --opt $o 2>> gpurun_out/cbam_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o', d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/cbam_bench.jsonl
done
cat gpurun_out/cbam_tests.log gpurun_out/cbam_bench.jsonl; tail -3 gpurun_out/cbam_bench.err
