#!/bin/bash
# Same-lease A/B of the ROUND-3 build against HEAD (is the driver's +31 % on proj / +18 % on fc2 the box or the
# code?).  Run from the repo root on an MI355X box through gpurun; needs tools/bin/libmi355attn_r03.so and tools/bin/r03_tree (built in
# the build container from `git archive b226749`, git-ignored, shipped with the tree).  Op level: the four ViT-Base GEMMs through both
# libraries IN ONE PROCESS (tools/ab_so.py, interleaved rounds).  Block level: `bench.py --only <block>` of the r03 tree and of HEAD,
# alternating processes A/B/A/B.  Output: gpurun_out/ab_<lease>/ ; tools/ab_r03_table.py turns any number of such directories into a table.
R=${GRAFT_REPO_ROOT:-$PWD}
L=${1:-$(date +%s)}
O=$R/gpurun_out/ab_$L
mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
cd $R
for op in proj fc2 fc1 qkv; do
  timeout 120 python tools/ab_so.py $op tools/bin/libmi355attn_r03.so > $O/op_$op.txt 2>&1
done
for rep in 1 2; do
  for blk in "XCABlock" "MixerLayer" "VisionTransformer"; do
    timeout 200 python tools/bin/r03_tree/bench.py --only "$blk" --no-cpu --no-strict --steps 20 --warmup 5 > $O/r03_${blk}_$rep.json 2> $O/r03_${blk}_$rep.err
    timeout 200 python bench.py --only "$blk" --no-cpu --no-strict --steps 20 --warmup 5 > $O/head_${blk}_$rep.json 2> $O/head_${blk}_$rep.err
  done
done
# the box's yardsticks, so that leases can be compared with each other
timeout 120 python - > $O/yardstick.json 2> $O/yardstick.err <<'PY'
import json, sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "pytorch-attention_amd")); sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
src = torch.randn(256 * 256 * 56 * 56, device=dev)           # the C2 footprint (822 MB), as in bench.py
print(json.dumps(bench.yardsticks(dev, src)))
PY
