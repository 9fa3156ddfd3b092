#!/usr/bin/env python
"""In-forward kernel tally of ViT-Base (B = 256) under option sets, alternated in one process:
python tools/vit_tally.py "gemm_w4=0,operand_pad=0" "" [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-attention_amd"))
import torch
import mi355attn
import bench_workloads as W
dev = torch.device("cuda", 0)
blk = W.workload_c5(256, dev)["blocks"][0]
m, x = blk["module"], blk["x"]
sets = [a for a in sys.argv[1:3]]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
defaults = {}
def apply(spec):
    for k, v in defaults.items():
        mi355attn.set_option(k, v)
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        defaults.setdefault(k, mi355attn.get_option(k))
        mi355attn.set_option(k, int(v))
with torch.no_grad():
    for r in range(rounds):
        for spec in sets:
            apply(spec)
            for _ in range(2):
                m(x)
            torch.cuda.synchronize()
            rows = mi355attn.kernel_trace(lambda: m(x))
            tot = sum(t for _, _, t, _, _ in rows)
            print("== round %d  [%s]  %.1f us traced" % (r, spec, tot))
            for tag, cnt, t, mn, mx in rows[:9]:
                print("   %8.1f us  x%-3d %7.1f avg  %s" % (t, cnt, t / cnt, tag[:90]))
