#!/bin/bash
# SQ counters of the kernels of one bench block (run on the GPU box from the repo root): tools/pmc_blocks.sh "<block substring>" <tag>
# Each counter group is its own rocprofv3 pass (kernel trace + pmc only).
R=${GRAFT_REPO_ROOT:-$PWD}
blk="$1"; tag=${2:-blk}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmcb_$tag$i -o p -- python $R/bench.py --no-cpu --no-strict --steps 2 --warmup 1 --only "$blk" > $R/gpurun_out/pmcb_$tag$i.log 2>&1
  python $R/tools/pmc_dump.py $R/gpurun_out/pmcb_$tag$i/p_results.db > $R/gpurun_out/pmcb_${tag}_g$i.txt 2>&1
  rm -rf $R/gpurun_out/pmcb_$tag$i
done
cat $R/gpurun_out/pmcb_${tag}_g*.txt > $R/gpurun_out/pmcb_$tag.txt
