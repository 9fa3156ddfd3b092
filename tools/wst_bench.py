#!/usr/bin/env python
"""gemm16_wst (weights stationary in registers, K = 768) against the tile kernels on the ViT-Base shapes: correctness vs an fp64 product of
sampled rows, time per call (HIP events on the launch stream, interleaved rounds).  python tools/wst_bench.py [rounds]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-attention_amd"))
import mi355attn  # noqa: E402
from mi355attn import StreamTimer  # noqa: E402
from mi355attn import functional as F  # noqa: E402

dev = torch.device("cuda", 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
VARS = (0, 2, 4)
torch.manual_seed(0)
for name, M, N, act in (("qkv", 50432, 2304, F.ACT_NONE), ("fc1", 50432, 3072, F.ACT_GELU), ("qkv_ragged", 50432 - 200 + 7, 2304, F.ACT_NONE),
                        ("fc1_small", 4096, 3072, F.ACT_GELU)):
    K = 768
    x16 = F.cast16(torch.randn(M, K, device=dev), 1)
    w16 = F.cast16((torch.randn(N, K, device=dev) / K ** 0.5).contiguous(), 1)
    b = torch.randn(N, device=dev)
    outs, tags = {}, {}
    for v in VARS:
        mi355attn.set_option("gemm_wst", v)
        tags[v] = [t for t, *_ in mi355attn.kernel_trace(lambda: outs.__setitem__(v, F.linear16(x16, w16, b, act=act, out16=True, precision=1)))]
    torch.cuda.synchronize()
    rows = torch.tensor([0, 1, 15, 16, 17, M // 2, M // 2 + 5, M - 17, M - 16, M - 2, M - 1], device=dev)
    ref = x16[rows].double() @ w16.double().t() + b.double()
    if act == F.ACT_GELU:
        ref = torch.nn.functional.gelu(ref)
    errs = {v: float(((outs[v][rows].double() - ref).abs().max() / ref.abs().max())) for v in outs}
    diffs = {v: float((outs[0].float() - outs[v].float()).abs().max() / outs[0].float().abs().max()) for v in outs}
    print("%-10s M=%d N=%d  kernels %s   max rel err vs fp64 %s   vs tile kernel (all rows) %s" % (
        name, M, N, [tags[v][0].split(" ")[0] for v in VARS], {v: "%.2e" % e for v, e in errs.items()}, {v: "%.2e" % e for v, e in diffs.items()}))
    for r in range(rounds):
        ts = []
        for v in VARS:
            mi355attn.set_option("gemm_wst", v)
            F.linear16(x16, w16, b, act=act, out16=True, precision=1)
            torch.cuda.synchronize()
            tm = StreamTimer(dev)
            tm.start()
            for _ in range(10):
                F.linear16(x16, w16, b, act=act, out16=True, precision=1)
            ts.append(tm.stop_ms() / 10 * 1e3)
        flop = 2.0 * M * N * K
        print("    round %d: " % r + "   ".join("opt %d: %.1f us (%.0f TFLOP/s)" % (v, tt, flop / tt / 1e6) for v, tt in zip(VARS, ts)))
    mi355attn.set_option("gemm_wst", 0)
