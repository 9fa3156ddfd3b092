#!/bin/bash
# Canonical measurement set of a round (run on an MI355X box from the repo root, e.g. through gpurun):
#   GPU test suite, the default bench line (c2, with CPU baseline), the other workloads, rocprofv3 kernel traces for
#   c2..c5 and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only) behind roofline.traffic.
# Outputs land in gpurun_out/: stats_<workload>.txt (tools/rocpd_stats.py run on the box) and the PMC .db files that
# tools/pmc_traffic.py / tools/pmc_mfma.py turn into profiles/*.json|txt.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -6 > gpurun_out/test_all.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
rm -f gpurun_out/bench_others.jsonl
timeout 300 python bench.py --workload c3 --steps 10 --warmup 3 --cpu-sample 16 >> gpurun_out/bench_others.jsonl 2>> gpurun_out/bench_others.err
for w in c4 c5 mixer da cswin mixer_full xcit zoo zoo2; do
  timeout 300 python bench.py --no-cpu --workload $w --steps 5 --warmup 2 >> gpurun_out/bench_others.jsonl 2>> gpurun_out/bench_others.err
done
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for w in c2 c3 c4 c5 cswin mixer_full xcit mixer zoo2; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o $w -- python $R/bench.py --no-cpu --workload $w --steps 3 --warmup 1 > $R/gpurun_out/prof_$w.log 2>&1
  python $R/tools/rocpd_stats.py $R/gpurun_out/prof_$w/${w}_results.db > $R/gpurun_out/stats_$w.txt 2>&1     # the .db files together exceed what gpurun copies back
  rm -rf $R/gpurun_out/prof_$w
done
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_c2 -o c2 -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > $R/gpurun_out/pmc_fetch_c2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_c2 -o c2 -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > $R/gpurun_out/pmc_write_c2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_mfma_c3 -o c3 -- python $R/bench.py --no-cpu --workload c3 --steps 3 --warmup 1 > $R/gpurun_out/pmc_mfma_c3.log 2>&1
cd $R
