#!/bin/bash
# Round 5, fourth GPU session: GPU suite on the scratch-free statistics gates, A/B of the 8-wave x 4-tile fused MLP (CSWin stage 1).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5d
mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
cd $R
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
echo "rc=$?" >> $O/pytest_gpu.log
for rep in 1 2 3; do
  for v in 0 1; do
    timeout 120 python bench.py --only "CSWinBlock s1" --no-cpu --no-strict --opt mlp_tt4=$v > $O/s1_tt4_${v}_$rep.json 2> /dev/null
  done
done
timeout 120 python bench.py --workload zoo --no-cpu --no-strict > $O/zoo.json 2> $O/zoo.err
