#!/bin/bash
# Round-3 measurement set (run on an MI355X box from the repo root through gpurun): default bench line with the CPU leg, rocprofv3
# kernel trace + stats of the same command, MFMA-utilisation counters of the ViT blocks, and per-block PMC traffic (two passes per
# block, kernel-trace only) assembled into profiles-ready files under gpurun_out/ (pmc_traffic.json carries the csrc fingerprint that
# bench.py checks before relaying `traffic`).
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 900 python bench.py > gpurun_out/r3_bench_all.json 2> gpurun_out/r3_bench_all.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_all -o all -- python $R/bench.py --no-cpu --no-strict --steps 3 --warmup 1 > $R/gpurun_out/r3_prof_all.log 2>&1
python $R/tools/rocpd_stats.py $R/gpurun_out/prof_all/all_results.db > $R/gpurun_out/r3_all_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/prof_all
for wl in c3 c5; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $R/gpurun_out/mfma_$wl -o m -- python $R/bench.py --workload $wl --no-cpu --no-strict --steps 3 --warmup 1 > $R/gpurun_out/r3_mfma_$wl.log 2>&1
  python $R/tools/pmc_mfma.py $R/gpurun_out/mfma_$wl/m_results.db > $R/gpurun_out/r3_${wl}_mfma_util.txt 2>&1
  rm -rf $R/gpurun_out/mfma_$wl
done
rm -f $R/gpurun_out/r3_pmc_blocks.jsonl
for blk in "SELayer" "CBAM" "ECALayer" "ViT Attention" "CSWinBlock s1" "CSWinBlock s2" "CSWinBlock s3" "CSWinBlock s4" "XCABlock" "XCA(" "DoubleAttention(64" "DoubleAttention(256" "MixerLayer" "VisionTransformer"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f_$tag -o f -- python $R/bench.py --no-cpu --no-strict --steps 3 --warmup 1 --only "$blk" > $R/gpurun_out/pmc_f_$tag.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w_$tag -o w -- python $R/bench.py --no-cpu --no-strict --steps 3 --warmup 1 --only "$blk" > $R/gpurun_out/pmc_w_$tag.log 2>&1
  name=$(python -c "import json,sys; d=json.loads([l for l in open('$R/gpurun_out/pmc_f_$tag.log') if l.startswith('{')][-1]); print(d['config']['blocks'][0]['block'])")
  python $R/tools/pmc_block_traffic.py "$name" $R/gpurun_out/pmc_f_$tag/f_results.db $R/gpurun_out/pmc_w_$tag/w_results.db 8 >> $R/gpurun_out/r3_pmc_blocks.jsonl 2>> $R/gpurun_out/r3_pmc_blocks.err
  rm -rf $R/gpurun_out/pmc_f_$tag $R/gpurun_out/pmc_w_$tag $R/gpurun_out/pmc_f_$tag.log $R/gpurun_out/pmc_w_$tag.log
done
cd $R
python tools/pmc_collect.py gpurun_out/r3_pmc_blocks.jsonl gpurun_out/pmc_traffic.json
