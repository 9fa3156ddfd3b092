#!/usr/bin/env python
"""gemm16_wslab (a column slab of W stationary in registers, K = 256 / 384 / 512, 16-bit output) against the tile kernels on the qkv / fc1 shapes of
XCiT, CSWin stages 3-4 and the Mixer: bits vs the tile kernels, error vs an fp64 product of sampled rows, time per call (HIP events on the launch
stream, interleaved rounds).  python tools/wslab_bench.py [rounds]"""
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
VARS = (0, 2)
SHAPES = (("xcit_qkv", 50176, 1152, 384, F.ACT_NONE), ("xcit_fc1", 50176, 1536, 384, F.ACT_GELU),
          ("cswin3_qkv", 50176, 768, 256, F.ACT_NONE), ("cswin3_fc1", 50176, 1024, 256, F.ACT_GELU),
          ("cswin4_qkv", 12544, 1536, 512, F.ACT_NONE), ("cswin4_fc1", 12544, 2048, 512, F.ACT_GELU),
          ("mixer_fc1", 50176, 2048, 512, F.ACT_GELU), ("ragged", 50176 - 45, 1152, 384, F.ACT_GELU))
torch.manual_seed(0)
for name, M, N, K, act in SHAPES:
    x16 = F.cast16(torch.randn(M, K, device=dev), 1)
    w16 = F.cast16((torch.randn(N, K, device=dev) / K ** 0.5).contiguous(), 1)
    b = torch.randn(N, device=dev)
    outs, tags = {}, {}
    for v in VARS:
        mi355attn.set_option("gemm_wslab", v)
        tags[v] = [t for t, *_ in mi355attn.kernel_trace(lambda: outs.__setitem__(v, F.linear16(x16, w16, b, act=act, out16=True, precision=1)))]
    torch.cuda.synchronize()
    rows = torch.tensor([0, 1, 15, 16, 17, 31, 32, M // 2, M // 2 + 5, M - 33, M - 17, M - 16, M - 2, M - 1], device=dev)
    ref = x16[rows].double() @ w16.double().t() + b.double()
    if act == F.ACT_GELU:
        ref = torch.nn.functional.gelu(ref)
    errs = {v: float(((outs[v][rows].double() - ref).abs().max() / ref.abs().max())) for v in outs}
    same = {v: bool(torch.equal(outs[0], outs[v])) for v in outs}
    print("%-11s M=%d N=%d K=%d kernels %s  err vs fp64 %s  same bits as option 0 %s" % (
        name, M, N, K, [tags[v][0].split(" ")[0] for v in VARS], {v: "%.1e" % e for v, e in errs.items()}, same), flush=True)
    for r in range(rounds):
        ts = []
        for v in VARS:
            mi355attn.set_option("gemm_wslab", v)
            F.linear16(x16, w16, b, act=act, out16=True, precision=1)
            torch.cuda.synchronize()
            tm = StreamTimer(dev)
            tm.start()
            for _ in range(10):
                F.linear16(x16, w16, b, act=act, out16=True, precision=1)
            ts.append(tm.stop_ms() / 10 * 1e3)
        flop = 2.0 * M * N * K
        print("    round %d: " % r + "   ".join("opt %d: %.1f us (%.0f TFLOP/s)" % (v, tt, flop / tt / 1e6) for v, tt in zip(VARS, ts)), flush=True)
    mi355attn.set_option("gemm_wslab", 0)
