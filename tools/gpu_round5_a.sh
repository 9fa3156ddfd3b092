#!/bin/bash
# Round 5, first GPU session: what telemetry the box offers, the full GPU test suite, the new bench line (+ --detail), the r03-vs-HEAD
# A/B of this lease, and the mixer_early A/B.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5a
mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
cd $R
{ ls -la /sys/class/drm/ 2>&1 | head -20; for c in /sys/class/drm/card[0-9]*/device; do echo "== $c"; cat $c/vendor 2>&1; ls $c/hwmon/*/ 2>&1 | head -40; cat $c/pp_dpm_sclk 2>&1 | head; done; } > $O/sysfs.txt 2>&1
( timeout 30 amd-smi metric --json > $O/amdsmi_metric.json 2> $O/amdsmi_metric.err; timeout 30 rocm-smi --showclocks --showpower --showtemp --json > $O/rocmsmi.json 2>&1 ) 
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "rc=$?" >> $O/pytest_gpu.log
( time timeout 400 python bench.py --detail $O/bench_detail.json > $O/bench_all.json 2> $O/bench_all.err ) 2> $O/bench_all.time
bash tools/ab_r03.sh lease1 > $O/ab.log 2>&1
for rep in 1 2; do
  for v in 0 1; do
    timeout 120 python bench.py --only MixerLayer --no-cpu --no-strict --opt mixer_early=$v > $O/mixer_early${v}_$rep.json 2> /dev/null
  done
done
