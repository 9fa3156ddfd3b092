#!/usr/bin/env python
"""A/B of option "gemm_pa_tail" (gemm16.hip: left-over rows of the two-accumulator GEMM on the small-tile ring kernel): the two
shapes of the bench it applies to, interleaved repeats in one process, then the two blocks they sit in."""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch-attention_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mi355attn
from mi355attn import functional as F


def time_us(fn, iters=40, warm=8):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / iters


def main():
    torch.manual_seed(0)
    for name, M, N, K in (("mixer fc2", 50176, 512, 2048), ("xcit fc2", 50176, 384, 1536), ("vit fc2 (no split: 158 left)", 50432, 768, 3072)):
        x = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda")
        res = {0: [], 30: []}
        for _ in range(3):
            for pct in (0, 30):
                mi355attn.set_option("gemm_pa_tail", pct)
                res[pct].append(time_us(lambda: F.linear16(x, w, b, resid=r, precision=1)))
        rows = mi355attn.kernel_trace(lambda: F.linear16(x, w, b, resid=r, precision=1))
        print(f"{name:32s} off {min(res[0]):7.1f} us  on {min(res[30]):7.1f} us   (all: {[round(v, 1) for v in res[0]]} / {[round(v, 1) for v in res[30]]})")
        for tag, cnt, tot, mn, mx in rows:
            print(f"      {tot:8.1f} us  {tag}")
    import bench_workloads as W
    blocks = W.workload_mixer(256, "cuda")["blocks"] + [b for b in W.workload_c4(256, "cuda")["blocks"] if b["name"].startswith("XCABlock")]
    for blk in blocks:
        mod, xin, extra = blk["module"].eval(), blk["x"], blk.get("fwd_args", ())
        res = {0: [], 30: []}
        with torch.no_grad():
            for _ in range(3):
                for pct in (0, 30):
                    mi355attn.set_option("gemm_pa_tail", pct)
                    res[pct].append(time_us(lambda: mod(xin, *extra), iters=20, warm=4))
        print(f"{blk['name']:32s} off {min(res[0]):7.1f} us  on {min(res[30]):7.1f} us   (all: {[round(v, 1) for v in res[0]]} / {[round(v, 1) for v in res[30]]})")

if __name__ == "__main__":
    main()
