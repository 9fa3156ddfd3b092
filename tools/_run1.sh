R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for blk in "XCABlock" "CSWinBlock s3" "MixerLayer"; do
tag=$(echo "$blk" | tr -c 'A-Za-z0-9' '_')
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o p -- python $R/bench.py --no-cpu --no-strict --steps 5 --warmup 2 --only "$blk" > $R/gpurun_out/prof_$tag.log 2>&1
echo "== $blk"; grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/prof_$tag.log | head -1
python $R/tools/rocpd_stats.py $R/gpurun_out/prof_$tag/p_results.db 2>&1 | head -16 | cut -c1-150
rm -rf $R/gpurun_out/prof_$tag
done
