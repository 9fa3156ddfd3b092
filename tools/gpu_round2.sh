#!/bin/bash
# Round-2 measurement set (run on an MI355X box from the repo root through gpurun): default bench line with the CPU leg, rocprofv3
# kernel trace of the same command, and per-block PMC traffic (two passes per block, kernel-trace only) for roofline.traffic.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 900 python bench.py > gpurun_out/r2_bench_all.json 2> gpurun_out/r2_bench_all.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_all -o all -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > $R/gpurun_out/r2_prof_all.log 2>&1
python $R/tools/rocpd_stats.py $R/gpurun_out/prof_all/all_results.db > $R/gpurun_out/r2_stats_all.txt 2>&1
rm -rf $R/gpurun_out/prof_all
rm -f $R/gpurun_out/r2_pmc_blocks.jsonl
for blk in "SELayer" "CBAM" "ECALayer" "ViT Attention" "CSWinBlock s1" "CSWinBlock s2" "CSWinBlock s3" "CSWinBlock s4" "XCABlock" "XCA(" "VisionTransformer"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f_$tag -o f -- python $R/bench.py --no-cpu --steps 3 --warmup 1 --only "$blk" > $R/gpurun_out/pmc_f_$tag.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w_$tag -o w -- python $R/bench.py --no-cpu --steps 3 --warmup 1 --only "$blk" > $R/gpurun_out/pmc_w_$tag.log 2>&1
  name=$(python -c "import json,sys; d=json.loads([l for l in open('$R/gpurun_out/pmc_f_$tag.log') if l.startswith('{')][-1]); print(d['config']['blocks'][0]['block'])")
  python $R/tools/pmc_block_traffic.py "$name" $R/gpurun_out/pmc_f_$tag/f_results.db $R/gpurun_out/pmc_w_$tag/w_results.db 8 >> $R/gpurun_out/r2_pmc_blocks.jsonl 2>> $R/gpurun_out/r2_pmc_blocks.err
  rm -rf $R/gpurun_out/pmc_f_$tag $R/gpurun_out/pmc_w_$tag
done
cd $R
# DoubleAttention (row a6; its own workload, not part of the default step): bench line + traffic of the (256,128,128)@56x56 block
timeout 300 python bench.py --workload da --no-cpu > gpurun_out/r2_bench_da.json 2> gpurun_out/r2_bench_da.err
cd /tmp
blk="DoubleAttention(256"
tag=DoubleAttention_256
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f_$tag -o f -- python $R/bench.py --workload da --no-cpu --steps 3 --warmup 1 --only "$blk" > $R/gpurun_out/pmc_f_$tag.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w_$tag -o w -- python $R/bench.py --workload da --no-cpu --steps 3 --warmup 1 --only "$blk" > $R/gpurun_out/pmc_w_$tag.log 2>&1
name=$(python -c "import json,sys; d=json.loads([l for l in open('$R/gpurun_out/pmc_f_$tag.log') if l.startswith('{')][-1]); print(d['config']['blocks'][0]['block'])")
python $R/tools/pmc_block_traffic.py "$name" $R/gpurun_out/pmc_f_$tag/f_results.db $R/gpurun_out/pmc_w_$tag/w_results.db 8 >> $R/gpurun_out/r2_pmc_blocks.jsonl 2>> $R/gpurun_out/r2_pmc_blocks.err
rm -rf $R/gpurun_out/pmc_f_$tag $R/gpurun_out/pmc_w_$tag
cd $R
