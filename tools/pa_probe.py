#!/usr/bin/env python
"""Timing probes for the two-accumulator GEMM: fp32-output epilogue with / without residual, 16-bit output, per K."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-attention_amd"))
import torch
import mi355attn
from mi355attn import StreamTimer
from mi355attn import functional as F
dev = torch.device("cuda", 0)
M = 256 * 197
def run(N, K, out16, resid_on, gelu, v, reps=10):
    torch.manual_seed(0)
    x16 = torch.randn(M, K, device=dev).half()
    w16 = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev) if resid_on else None
    act = F.ACT_GELU if gelu else F.ACT_NONE
    mi355attn.set_option("gemm_variant", v)
    for _ in range(3):
        F.linear16(x16, w16, b, act=act, resid=resid, out16=out16, precision=1)
    torch.cuda.synchronize()
    tm = StreamTimer(dev); tm.start()
    for _ in range(reps):
        F.linear16(x16, w16, b, act=act, resid=resid, out16=out16, precision=1)
    ms = tm.stop_ms() / reps
    mi355attn.set_option("gemm_variant", 0)
    return ms
for (N, K) in ((768, 768), (768, 1536), (768, 3072), (2304, 768)):
    for v in (15, 16):
        r = {}
        r["f32+res"] = run(N, K, False, True, False, v)
        r["f32"] = run(N, K, False, False, False, v)
        r["h16"] = run(N, K, True, False, False, v)
        print(N, K, "v%d" % v, {k: round(x * 1e3, 1) for k, x in r.items()}, "TF(f32+res)=%.0f" % (2.0 * M * N * K / r["f32+res"] / 1e9), flush=True)
