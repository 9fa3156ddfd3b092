#!/bin/bash
# quick GPU check: tests selected by -k expression ($1), then per-block dispatch sequences for the blocks in $2 (| separated)
mkdir -p gpurun_out/q
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 900 python -m pytest tests -m gpu -q -x -k "$1" > gpurun_out/q/t.log 2>&1
echo "rc=$?" >> gpurun_out/q/t.log
cd /tmp && export TMPDIR=/tmp
IFS='|' read -ra BL <<< "$2"
for blk in "${BL[@]}"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/q/p_$tag -o k -- python $R/bench.py --no-cpu --no-strict --steps 6 --warmup 2 --only "$blk" > $R/gpurun_out/q/log_$tag.txt 2>&1
  python $R/tools/rocpd_seq.py $R/gpurun_out/q/p_$tag/k_results.db 0 "$blk" > $R/gpurun_out/q/seq_$tag.txt 2>&1
  rm -rf $R/gpurun_out/q/p_$tag
done
