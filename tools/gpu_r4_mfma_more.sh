mkdir -p gpurun_out/r4
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4
cd /tmp && export TMPDIR=/tmp
for wl in c4 mixer da; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $O/mfma_$wl -o m -- python $R/bench.py --workload $wl --no-cpu --no-strict --steps 3 --warmup 1 > $O/mfma_$wl.log 2>&1
  python $R/tools/pmc_mfma.py $O/mfma_$wl/m_results.db > $O/${wl}_mfma_util.txt 2>&1
  rm -rf $O/mfma_$wl $O/mfma_$wl.log
done
