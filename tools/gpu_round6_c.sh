#!/bin/bash
# Round-6 lease C: xca_tr_kernel + the corrected range-wait test; XCABlock / XCA sequences.  gpurun_out/r6c/
mkdir -p gpurun_out/r6c
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6c
rm -f $O/*
cd $R
timeout 900 python -m pytest tests/test_round6_kernels_gpu.py tests/test_round6_gpu.py tests/test_full_size_gpu.py -q > $O/pytest_c.log 2>&1
echo "rc=$?" >> $O/pytest_c.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_ops_gpu.py -q -k "xc" > $O/pytest_c2.log 2>&1
echo "rc=$?" >> $O/pytest_c2.log
for i in 1 2; do
  timeout 200 python bench.py --no-cpu --no-strict --no-calib --workload c4 > $O/bench_c4_$i.json 2> $O/bench_c4_$i.err
  timeout 200 python bench.py --no-cpu --no-strict --no-calib --workload c4 --opt xca_tr=0 > $O/bench_c4_tr0_$i.json 2> $O/bench_c4_tr0_$i.err
done
cd /tmp && export TMPDIR=/tmp
BLOCKS=("XCABlock" "XCA(")
for blk in "${BLOCKS[@]}"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 200 rocprofv3 --kernel-trace -d $O/p_$tag -o k -- python $R/bench.py --no-cpu --no-strict --steps 6 --warmup 2 --only "$blk" > $O/log_$tag.txt 2>&1
  python $R/tools/rocpd_seq.py $O/p_$tag/k_results.db 0 "$blk" > $O/seq_$tag.txt 2>&1
  rm -rf $O/p_$tag $O/log_$tag.txt
done
