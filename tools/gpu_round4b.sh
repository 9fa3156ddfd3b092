#!/bin/bash
# Round-4 re-measurement after the LPI patch kernel / left-over-rows split / ln_stats change (run on an MI355X box from the repo root
# through gpurun): the parts of tools/gpu_round4.sh that depend on the kernels, plus the full GPU test suite and the LPI probe.
#   1. pytest -m gpu                                                          -> r4/pytest_gpu.log
#   2. default bench line with the CPU leg                                   -> r4/bench_all.json
#   3. rocprofv3 kernel trace of the same step: --stats style table           -> r4/all_kernel_stats.txt
#   4. per block: dispatch-ordered kernel table of ONE forward                -> r4/seq_<block>.txt
#   5. MFMA-utilisation counters of the XCiT workload                         -> r4/c4_mfma_util.txt
#   6. per block PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes)       -> r4/pmc_blocks.jsonl, pmc_traffic.json (stamped with the csrc hash)
#   7. tools/bin/lpi_probe (cross-compiled tools/lpi_probe.hip)               -> r4/lpi_probe.txt
mkdir -p gpurun_out/r4
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4
rm -f $O/*
timeout 400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
( time timeout 900 python bench.py > $O/bench_all.json 2> $O/bench_all.err ) 2> $O/bench_all.time
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_all -o all -- python $R/bench.py --no-cpu --no-strict --steps 3 --warmup 1 > $O/prof_all.log 2>&1
python $R/tools/rocpd_stats.py $O/prof_all/all_results.db > $O/all_kernel_stats.txt 2>&1
rm -rf $O/prof_all
BLOCKS=("SELayer" "CBAM" "ECALayer" "ViT Attention" "CSWinBlock s1" "CSWinBlock s2" "CSWinBlock s3" "CSWinBlock s4" "XCABlock" "XCA(" "DoubleAttention(64" "DoubleAttention(256" "MixerLayer" "VisionTransformer")
for blk in "${BLOCKS[@]}"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 200 rocprofv3 --kernel-trace -d $O/p_$tag -o k -- python $R/bench.py --no-cpu --no-strict --steps 6 --warmup 2 --only "$blk" > $O/log_$tag.txt 2>&1
  python $R/tools/rocpd_seq.py $O/p_$tag/k_results.db 0 "$blk" > $O/seq_$tag.txt 2>&1
  rm -rf $O/p_$tag $O/log_$tag.txt
done
for wl in c4; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $O/mfma_$wl -o m -- python $R/bench.py --workload $wl --no-cpu --no-strict --steps 3 --warmup 1 > $O/mfma_$wl.log 2>&1
  python $R/tools/pmc_mfma.py $O/mfma_$wl/m_results.db > $O/${wl}_mfma_util.txt 2>&1
  rm -rf $O/mfma_$wl $O/mfma_$wl.log
done
rm -f $O/pmc_blocks.jsonl
for blk in "${BLOCKS[@]}"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f_$tag -o f -- python $R/bench.py --no-cpu --no-strict --steps 3 --warmup 1 --only "$blk" > $O/pmc_f_$tag.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w_$tag -o w -- python $R/bench.py --no-cpu --no-strict --steps 3 --warmup 1 --only "$blk" > $O/pmc_w_$tag.log 2>&1
  name=$(python -c "import json,sys; d=json.loads([l for l in open('$O/pmc_f_$tag.log') if l.startswith('{')][-1]); print(d['config']['blocks'][0]['block'])")
  python $R/tools/pmc_block_traffic.py "$name" $O/pmc_f_$tag/f_results.db $O/pmc_w_$tag/w_results.db 8 >> $O/pmc_blocks.jsonl 2>> $O/pmc_blocks.err
  rm -rf $O/pmc_f_$tag $O/pmc_w_$tag $O/pmc_f_$tag.log $O/pmc_w_$tag.log
done
cd $R
python tools/pmc_collect.py $O/pmc_blocks.jsonl $O/pmc_traffic.json > $O/pmc_collect.log 2>&1
tools/bin/lpi_probe > $O/lpi_probe.txt 2>&1
