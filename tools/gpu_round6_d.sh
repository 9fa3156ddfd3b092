#!/bin/bash
# Round-6 lease D: mlp_wide (fused LayerNorm + MLP at C = 256 / 384), xca_tr with V requested up front.  gpurun_out/r6d/
mkdir -p gpurun_out/r6d
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6d
rm -f $O/*
cd $R
timeout 900 python -m pytest tests/test_round6_kernels_gpu.py -q -x > $O/pytest_d.log 2>&1
echo "rc=$?" >> $O/pytest_d.log
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_round6_gpu.py -q > $O/pytest_d2.log 2>&1
echo "rc=$?" >> $O/pytest_d2.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "xc or cswin" > $O/pytest_d3.log 2>&1
echo "rc=$?" >> $O/pytest_d3.log
for i in 1 2; do
  timeout 200 python bench.py --no-cpu --no-strict --no-calib --workload c4 > $O/bench_c4_$i.json 2> $O/bench_c4_$i.err
  timeout 200 python bench.py --no-cpu --no-strict --no-calib --workload c4 --opt mlp_wide=0 > $O/bench_c4_wide0_$i.json 2> $O/bench_c4_wide0_$i.err
done
cd /tmp && export TMPDIR=/tmp
BLOCKS=("XCABlock" "CSWinBlock s3")
for blk in "${BLOCKS[@]}"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 200 rocprofv3 --kernel-trace -d $O/p_$tag -o k -- python $R/bench.py --no-cpu --no-strict --steps 6 --warmup 2 --only "$blk" > $O/log_$tag.txt 2>&1
  python $R/tools/rocpd_seq.py $O/p_$tag/k_results.db 0 "$blk" > $O/seq_$tag.txt 2>&1
  rm -rf $O/p_$tag $O/log_$tag.txt
done
