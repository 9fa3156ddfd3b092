#!/bin/bash
# Round-4 final re-measurement after the fused Mixer token-mixing kernel (run on an MI355X box from the repo root through gpurun):
# full GPU test suite, kernel table of the step, the MixerLayer dispatch table, PMC traffic of every block (the stamped file), and --
# with the fresh traffic file in place -- the default bench line.
mkdir -p gpurun_out/r4
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4
rm -f $O/*
timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_all -o all -- python $R/bench.py --no-cpu --no-strict --steps 3 --warmup 1 > $O/prof_all.log 2>&1
python $R/tools/rocpd_stats.py $O/prof_all/all_results.db > $O/all_kernel_stats.txt 2>&1
rm -rf $O/prof_all
for blk in "MixerLayer"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 200 rocprofv3 --kernel-trace -d $O/p_$tag -o k -- python $R/bench.py --no-cpu --no-strict --steps 6 --warmup 2 --only "$blk" > $O/log_$tag.txt 2>&1
  python $R/tools/rocpd_seq.py $O/p_$tag/k_results.db 0 "$blk" > $O/seq_$tag.txt 2>&1
  rm -rf $O/p_$tag $O/log_$tag.txt
done
BLOCKS=("SELayer" "CBAM" "ECALayer" "ViT Attention" "CSWinBlock s1" "CSWinBlock s2" "CSWinBlock s3" "CSWinBlock s4" "XCABlock" "XCA(" "DoubleAttention(64" "DoubleAttention(256" "MixerLayer" "VisionTransformer")
rm -f $O/pmc_blocks.jsonl
for blk in "${BLOCKS[@]}"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f_$tag -o f -- python $R/bench.py --no-cpu --no-strict --steps 3 --warmup 1 --only "$blk" > $O/pmc_f_$tag.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w_$tag -o w -- python $R/bench.py --no-cpu --no-strict --steps 3 --warmup 1 --only "$blk" > $O/pmc_w_$tag.log 2>&1
  name=$(python -c "import json,sys; d=json.loads([l for l in open('$O/pmc_f_$tag.log') if l.startswith('{')][-1]); print(d['config']['blocks'][0]['block'])")
  python $R/tools/pmc_block_traffic.py "$name" $O/pmc_f_$tag/f_results.db $O/pmc_w_$tag/w_results.db 8 >> $O/pmc_blocks.jsonl 2>> $O/pmc_blocks.err
  rm -rf $O/pmc_f_$tag $O/pmc_w_$tag $O/pmc_f_$tag.log $O/pmc_w_$tag.log
done
cd $R
python tools/pmc_collect.py $O/pmc_blocks.jsonl $O/pmc_traffic.json > $O/pmc_collect.log 2>&1
cp $O/pmc_traffic.json profiles/pmc_traffic.json
( time timeout 300 python bench.py > $O/bench_all.json 2> $O/bench_all.err ) 2> $O/bench_all.time
