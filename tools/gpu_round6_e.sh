#!/bin/bash
# Round-6 lease E: whole GPU suite on the tree with the statistics fold; c4 A/B.  gpurun_out/r6e/
mkdir -p gpurun_out/r6e
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6e
rm -f $O/*
cd $R
export MI355_MARGIN_OUT=$O/r06_parity_margin.md
( time timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_gpu.time
echo "rc=$?" >> $O/pytest_gpu.log
for i in 1 2; do
  timeout 200 python bench.py --no-cpu --no-strict --no-calib --workload c4 > $O/bench_c4_$i.json 2> $O/bench_c4_$i.err
done
( time timeout 400 python bench.py --detail $O/bench_detail.json > $O/bench_all.json 2> $O/bench_all.err ) 2> $O/bench_all.time
