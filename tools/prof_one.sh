#!/bin/bash
# rocprofv3 kernel statistics of one python script (run on the GPU box from the repo root): tools/prof_one.sh <script.py> [tag]
R=${GRAFT_REPO_ROOT:-$PWD}
tag=${2:-one}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag -- bash -c "python $R/$1" > $R/gpurun_out/prof_$tag.log 2>&1
python $R/tools/rocpd_stats.py $R/gpurun_out/prof_$tag/${tag}_results.db 2>&1 | head -${3:-14}
rm -rf $R/gpurun_out/prof_$tag
