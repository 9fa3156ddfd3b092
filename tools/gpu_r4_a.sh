#!/bin/bash
# Round-4 call A: per-block dispatch sequences (rocprofv3 --kernel-trace of `bench.py --only <block>`) + the vendor-GEMM yardstick.
mkdir -p gpurun_out/r4a
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 200 python tools/blas_probe.py > gpurun_out/r4a/blas_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for blk in "XCABlock" "XCA(" "CSWinBlock s1" "CSWinBlock s2" "CSWinBlock s3" "CSWinBlock s4" "MixerLayer" "DoubleAttention(256" "ViT Attention" "VisionTransformer"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/r4a/p_$tag -o k -- python $R/bench.py --no-cpu --no-strict --steps 4 --warmup 2 --only "$blk" > $R/gpurun_out/r4a/log_$tag.txt 2>&1
  python $R/tools/rocpd_seq.py $R/gpurun_out/r4a/p_$tag/k_results.db 0 "$blk" > $R/gpurun_out/r4a/seq_$tag.txt 2>&1
  rm -rf $R/gpurun_out/r4a/p_$tag
done
