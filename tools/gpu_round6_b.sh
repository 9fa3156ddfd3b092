#!/bin/bash
# Round-6 lease B: the weight-stationary GEMM and the one-event range wait.  gpurun_out/r6b/
mkdir -p gpurun_out/r6b
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6b
rm -f $O/*
cd $R
timeout 900 python -m pytest tests/test_round6_kernels_gpu.py tests/test_round6_gpu.py "tests/test_round4_gpu.py::test_guarded_forward_catches_a_saturated_fused_intermediate" "tests/test_round5_gpu.py::test_fused_mixer_kernel_reports_a_saturating_intermediate" tests/test_full_size_gpu.py -q -x > $O/pytest_b.log 2>&1
echo "rc=$?" >> $O/pytest_b.log
for i in 1 2; do
  timeout 200 python bench.py --no-cpu --no-strict --no-calib --opt range_fallback=0 > $O/bench_rf0_$i.json 2> $O/bench_rf0_$i.err
  timeout 200 python bench.py --no-cpu --no-strict --no-calib > $O/bench_rf1_$i.json 2> $O/bench_rf1_$i.err
  timeout 200 python bench.py --no-cpu --no-strict --no-calib --opt gemm_wreg=0 > $O/bench_wreg0_$i.json 2> $O/bench_wreg0_$i.err
done
cd /tmp && export TMPDIR=/tmp
BLOCKS=("XCABlock" "XCA(" "CSWinBlock s3" "CSWinBlock s1" "ViT Attention")
for blk in "${BLOCKS[@]}"; do
  tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
  timeout 200 rocprofv3 --kernel-trace -d $O/p_$tag -o k -- python $R/bench.py --no-cpu --no-strict --steps 6 --warmup 2 --only "$blk" > $O/log_$tag.txt 2>&1
  python $R/tools/rocpd_seq.py $O/p_$tag/k_results.db 0 "$blk" > $O/seq_$tag.txt 2>&1
  rm -rf $O/p_$tag $O/log_$tag.txt
done
