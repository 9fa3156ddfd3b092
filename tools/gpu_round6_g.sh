#!/bin/bash
# Round-6 lease G: the LayerNorm-emitting projection (CSWin stage 3).  gpurun_out/r6g/
mkdir -p gpurun_out/r6g
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6g
rm -f $O/*
cd $R
timeout 900 python -m pytest tests/test_round6_kernels_gpu.py tests/test_full_size_gpu.py -q > $O/pytest_g.log 2>&1
echo "rc=$?" >> $O/pytest_g.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "cswin" > $O/pytest_g2.log 2>&1
echo "rc=$?" >> $O/pytest_g2.log
for i in 1 2 3; do
  timeout 200 python bench.py --no-cpu --no-strict --no-calib --workload c4 > $O/bench_c4_$i.json 2> $O/bench_c4_$i.err
  timeout 200 python bench.py --no-cpu --no-strict --no-calib --workload c4 --opt gemm_wreg=0 > $O/bench_c4_wreg0_$i.json 2> $O/bench_c4_wreg0_$i.err
done
timeout 300 python bench.py --workload cswin --no-cpu --no-calib --no-strict > $O/next_cswin.json 2> $O/next_cswin.err
timeout 300 python bench.py --workload cswin --no-cpu --no-calib --no-strict --opt gemm_wreg=0 > $O/next_cswin_wreg0.json 2> $O/next_cswin_wreg0.err
cd /tmp && export TMPDIR=/tmp
for blk in "CSWinBlock s3"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 200 rocprofv3 --kernel-trace -d $O/p_$tag -o k -- python $R/bench.py --no-cpu --no-strict --steps 6 --warmup 2 --only "$blk" > $O/log_$tag.txt 2>&1
  python $R/tools/rocpd_seq.py $O/p_$tag/k_results.db 0 "$blk" > $O/seq_$tag.txt 2>&1
  rm -rf $O/p_$tag $O/log_$tag.txt
done
