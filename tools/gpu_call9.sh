#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_gemm_a -o g -- python $R/tools/gemm_bench.py 7,10 > $R/gpurun_out/pmc_gemm_a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $R/gpurun_out/pmc_gemm_b -o g -- python $R/tools/gemm_bench.py 7,10 > $R/gpurun_out/pmc_gemm_b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_gemm_c -o g -- python $R/tools/gemm_bench.py 7,10 > $R/gpurun_out/pmc_gemm_c.log 2>&1
cd $R
tail -3 gpurun_out/pmc_gemm_b.log
