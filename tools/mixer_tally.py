#!/usr/bin/env python
"""Kernel tally of one MixerLayer(512, 196) forward at B = 256 with the token mixing fused (option mixer_fused = 1) and as three launches."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-attention_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mi355attn
import bench_workloads as W
blk = W.workload_mixer(256, "cuda")["blocks"][0]
m, x = blk["module"].eval(), blk["x"]
with torch.no_grad():
    for opt in (1, 0):
        mi355attn.set_option("mixer_fused", opt)
        for _ in range(3):
            m(x)
        rows = mi355attn.kernel_trace(lambda: [m(x) for _ in range(5)])
        print(f"mixer_fused = {opt}: {sum(r[2] for r in rows) / 5:.1f} us per forward")
        for tag, cnt, tot, mn, mx in rows:
            print(f"   {tot / cnt:8.1f} us (min {mn:6.1f})  x{cnt // 5}  {tag}")
