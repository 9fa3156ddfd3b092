#!/bin/bash
# Round 5, third GPU session: full GPU suite after the SE vote fix, bench line with the corrected MFMA yardstick and the PCI-matched
# telemetry, r03-vs-HEAD A/B of this lease.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5c
mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
cd $R
timeout 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
echo "rc=$?" >> $O/pytest_gpu.log
( time timeout 400 python bench.py --detail $O/bench_detail.json > $O/bench_all.json 2> $O/bench_all.err ) 2> $O/bench_all.time
bash tools/ab_r03.sh lease3 > $O/ab.log 2>&1
