#!/bin/bash
# Round-6 evidence set (run on an MI355X box from the repo root through gpurun).  Everything lands under gpurun_out/r6f/; the files that
# are evidence get copied into profiles/ afterwards (r06_*).
#   1. full GPU test suite                                                             -> r6f/pytest_gpu.log
#   2. rocprofv3 kernel trace of the default step: --stats style table                 -> r6f/all_kernel_stats.txt
#   3. per block: dispatch-ordered kernel table of ONE forward (tools/rocpd_seq.py)    -> r6f/seq_<block>.txt
#   4. MFMA-utilisation counters of the ViT blocks                                     -> r6f/c3_mfma_util.txt, c5_mfma_util.txt
#   5. per block PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes), 14 blocks     -> r6f/pmc_blocks.jsonl, pmc_traffic.json (stamped with the csrc hash)
#   6. next rows: cswin / xcit / mixer_full / zoo / zoo2 / f1 at B = 256: bench lines + PMC traffic per block
#   7. with the fresh traffic file in place: the default bench line (+ --detail)       -> r6f/bench_all.json, bench_detail.json
mkdir -p gpurun_out/r6f
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6f
rm -f $O/*
cd $R
export MI355_MARGIN_OUT=$O/r06_parity_margin.md
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
echo "rc=$?" >> $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_all -o all -- python $R/bench.py --no-cpu --no-strict --no-calib --steps 3 --warmup 1 > $O/prof_all.log 2>&1
python $R/tools/rocpd_stats.py $O/prof_all/all_results.db > $O/all_kernel_stats.txt 2>&1
rm -rf $O/prof_all
BLOCKS=("SELayer" "CBAM" "ECALayer" "ViT Attention" "CSWinBlock s1" "CSWinBlock s2" "CSWinBlock s3" "CSWinBlock s4" "XCABlock" "XCA(" "DoubleAttention(64" "DoubleAttention(256" "MixerLayer" "VisionTransformer")
for blk in "${BLOCKS[@]}"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 200 rocprofv3 --kernel-trace -d $O/p_$tag -o k -- python $R/bench.py --no-cpu --no-strict --steps 6 --warmup 2 --only "$blk" > $O/log_$tag.txt 2>&1
  python $R/tools/rocpd_seq.py $O/p_$tag/k_results.db 0 "$blk" > $O/seq_$tag.txt 2>&1
  rm -rf $O/p_$tag $O/log_$tag.txt
done
for wl in c3 c5; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $O/mfma_$wl -o m -- python $R/bench.py --workload $wl --no-cpu --no-strict --no-calib --steps 3 --warmup 1 > $O/mfma_$wl.log 2>&1
  python $R/tools/pmc_mfma.py $O/mfma_$wl/m_results.db > $O/${wl}_mfma_util.txt 2>&1
  rm -rf $O/mfma_$wl $O/mfma_$wl.log
done
rm -f $O/pmc_blocks.jsonl
for blk in "${BLOCKS[@]}"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f_$tag -o f -- python $R/bench.py --no-cpu --no-strict --steps 3 --warmup 1 --only "$blk" > $O/pmc_f_$tag.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w_$tag -o w -- python $R/bench.py --no-cpu --no-strict --steps 3 --warmup 1 --only "$blk" > $O/pmc_w_$tag.log 2>&1
  name=$(python -c "import json,sys; d=json.loads([l for l in open('$O/pmc_f_$tag.log') if l.startswith('{')][-1]); print(d['blocks'][0]['block'])")
  python $R/tools/pmc_block_traffic.py "$name" $O/pmc_f_$tag/f_results.db $O/pmc_w_$tag/w_results.db 8 >> $O/pmc_blocks.jsonl 2>> $O/pmc_blocks.err
  rm -rf $O/pmc_f_$tag $O/pmc_w_$tag $O/pmc_f_$tag.log $O/pmc_w_$tag.log
done
cd $R
python tools/pmc_collect.py $O/pmc_blocks.jsonl $O/pmc_traffic.json > $O/pmc_collect.log 2>&1
cp $O/pmc_traffic.json profiles/pmc_traffic.json
# ---- next rows: one bench line per workload, then PMC traffic per block of each -------------------------------------------------------------
for wl in cswin xcit mixer_full zoo zoo2 f1; do
  timeout 300 python bench.py --workload $wl --no-cpu --no-calib --detail $O/next_${wl}_detail.json > $O/next_$wl.json 2> $O/next_$wl.err
done
cd /tmp
rm -f $O/pmc_next.jsonl
for wl in cswin xcit mixer_full zoo zoo2 f1; do
  python - "$O/next_$wl.json" > $O/names_$wl.txt <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
for b in d["blocks"]:
    print(b["block"])
PY
  while IFS= read -r name; do
    tag=$(echo "${wl}_$name" | tr -c 'A-Za-z0-9' '_' | cut -c1-60)
    timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/nf_$tag -o f -- python $R/bench.py --workload $wl --no-cpu --no-strict --no-calib --steps 3 --warmup 1 --only "$name" > $O/nf_$tag.log 2>&1
    timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/nw_$tag -o w -- python $R/bench.py --workload $wl --no-cpu --no-strict --no-calib --steps 3 --warmup 1 --only "$name" > $O/nw_$tag.log 2>&1
    python $R/tools/pmc_block_traffic.py "$wl: $name" $O/nf_$tag/f_results.db $O/nw_$tag/w_results.db 8 >> $O/pmc_next.jsonl 2>> $O/pmc_next.err
    rm -rf $O/nf_$tag $O/nw_$tag $O/nf_$tag.log $O/nw_$tag.log
  done < $O/names_$wl.txt
done
cd $R
( time timeout 400 python bench.py --detail $O/bench_detail.json > $O/bench_all.json 2> $O/bench_all.err ) 2> $O/bench_all.time
