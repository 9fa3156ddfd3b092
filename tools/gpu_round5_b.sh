#!/bin/bash
# Round 5, second GPU session: full GPU suite on the scratch-free SE / ECA kernels and the small-output GEMM, the MFMA shape probe,
# the bench line, the r03-vs-HEAD A/B of this lease.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5b
mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "rc=$?" >> $O/pytest_gpu.log
timeout 120 tools/bin/mfma_probe > $O/mfma_probe.txt 2>&1
( time timeout 400 python bench.py --detail $O/bench_detail.json > $O/bench_all.json 2> $O/bench_all.err ) 2> $O/bench_all.time
bash tools/ab_r03.sh lease2 > $O/ab.log 2>&1
