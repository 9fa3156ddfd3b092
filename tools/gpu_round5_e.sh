#!/bin/bash
# Round 5: in-kernel Mixer statistics -- tests + same-box A/B of the block.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5e
mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
cd $R
timeout 600 python -m pytest tests -m gpu -q -x -k "mixer" > $O/pytest_mixer.log 2>&1
echo "rc=$?" >> $O/pytest_mixer.log
for rep in 1 2 3; do
  for v in 0 1; do
    timeout 120 python bench.py --only MixerLayer --no-cpu --no-strict --opt mixer_stats=$v > $O/mixer_stats${v}_$rep.json 2> /dev/null
  done
done
