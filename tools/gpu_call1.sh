#!/bin/bash
# first GPU call: probe the box, parity-test the channel-attention family, bench + chunk sweep + rocprof
mkdir -p gpurun_out
{
echo "== host"; nproc; lscpu | grep -E 'Model name|Socket|Thread|Core' ; free -g | head -2
echo "== gpu"; rocminfo | grep -E 'Marketing Name|gfx|Compute Unit|Max Clock' | head -12
ls /root/reference 2>&1 | head -2
} > gpurun_out/probe.txt 2>&1
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_chan_attn_gpu.py -m gpu -x -q > gpurun_out/test_chan.log 2>&1
echo "chan tests exit $?" >> gpurun_out/probe.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "se64 or cbam64 or eca64 or se256 or cbam256 or eca256" > gpurun_out/test_parity_chan.log 2>&1
echo "parity tests exit $?" >> gpurun_out/probe.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
for c in 2 8 16 24 32 48 64 128 256; do
  timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 --chunk-images $c >> gpurun_out/bench_chunk_sweep.jsonl 2>> gpurun_out/bench_chunk_sweep.err
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c2 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_c2.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_c2 -name "*.db" -size +20M -delete 2>/dev/null
ls -la gpurun_out gpurun_out/prof_c2 >> gpurun_out/probe.txt 2>&1
