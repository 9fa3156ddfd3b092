#!/bin/bash
# Round-6 lease A (run on an MI355X box from the repo root through gpurun); everything lands under gpurun_out/r6a/.
#   1. new tests of the round (range fallback, contention, aliases, ragged gloo gather) + the parity-margin table
#   2. the whole GPU suite
#   3. default bench line with --detail; the same step with range_fallback = 0 (cost of the default boundary behaviour), no CPU leg
#   4. next-row workloads on HEAD at B = 256: cswin, xcit, mixer_full, zoo, zoo2, f1
mkdir -p gpurun_out/r6a
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6a
rm -f $O/*
cd $R
export MI355_MARGIN_OUT=$O/r06_parity_margin.md
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_parity_margin_gpu.py -q -x -s > $O/pytest_round6.log 2>&1
echo "rc=$?" >> $O/pytest_round6.log
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_parity_margin_gpu.py > $O/pytest_gpu.log 2>&1
echo "rc=$?" >> $O/pytest_gpu.log
( time timeout 400 python bench.py --detail $O/bench_detail.json > $O/bench_all.json 2> $O/bench_all.err ) 2> $O/bench_all.time
timeout 200 python bench.py --no-cpu --no-strict --opt range_fallback=0 > $O/bench_rf0.json 2> $O/bench_rf0.err
timeout 200 python bench.py --no-cpu --no-strict > $O/bench_rf1.json 2> $O/bench_rf1.err
timeout 200 python bench.py --no-cpu --no-strict --opt range_fallback=0 > $O/bench_rf0b.json 2> $O/bench_rf0b.err
for wl in cswin xcit mixer_full zoo zoo2 f1; do
  timeout 300 python bench.py --workload $wl --no-cpu --no-calib --detail $O/next_${wl}_detail.json > $O/next_$wl.json 2> $O/next_$wl.err
done
