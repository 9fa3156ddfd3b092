#!/bin/bash
# same-lease A/B of option sets on one bench block: tools/ab_opts.sh <block substring> "<opts A>" "<opts B>" [rounds]
# e.g. tools/ab_opts.sh ViT-Base "--opt gemm_w4=0 --opt operand_pad=0" "" 3
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
blk="$1"; A="$2"; B="$3"; n="${4:-3}"
for i in $(seq 1 "$n"); do
  for tag in A B; do
    if [ "$tag" = A ]; then o="$A"; else o="$B"; fi
    python bench.py --steps 10 --warmup 3 --no-calib --no-cpu --no-strict --only "$blk" $o 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$tag [$o]', round(d['ms_per_step'], 4), 'ms')
"
  done
done
