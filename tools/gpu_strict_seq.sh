#!/bin/bash
# dispatch-ordered kernel table of one STRICT-mode (precision 0) forward per block -> gpurun_out/strict_seq/seq_<block>.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/strict_seq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BLOCKS=("CSWinBlock s1" "CSWinBlock s2" "DoubleAttention(64" "DoubleAttention(256" "MixerLayer" "ViT Attention")
for blk in "${BLOCKS[@]}"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 120 rocprofv3 --kernel-trace -d $O/p_$tag -o k -- python $R/bench.py --no-cpu --no-strict --no-calib --no-models --precision 0 --steps 6 --warmup 2 --only "$blk" > $O/log_$tag.txt 2>&1
  timeout 60 python $R/tools/rocpd_seq.py $O/p_$tag/k_results.db 0 "$blk" > $O/seq_$tag.txt 2>&1
  rm -rf $O/p_$tag
done
