#!/usr/bin/env python
"""What the fp32 + residual epilogues of ViT-Base's proj / fc2 cost against a plain 16-bit-output GEMM of the same shape, same process,
interleaved ("say how much of the 60 us that explains with a counter, not an argument"): three epilogues per shape
through mi355_linear16_ws_fwd -- 16-bit out (the vendor yardstick's traffic), fp32 out, fp32 out + fp32 residual -- with the bytes each moves
and the kernel the dispatcher picked (mi355attn.kernel_trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-attention_amd"))
import torch, mi355attn
from mi355attn import functional as F
dev = torch.device("cuda", 0)
torch.manual_seed(0)
M = 256 * 197
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, N, K, act in (("proj", 768, 768, F.ACT_NONE), ("fc2", 768, 3072, F.ACT_GELU)):
    x16 = torch.randn(M, K, device=dev).half()
    w16 = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.randn(N, device=dev) * 0.1
    r = torch.randn(M, N, device=dev)
    cfgs = {"out16": lambda: F.linear16(x16, w16, b, act=act, out16=True, precision=1),
            "out32": lambda: F.linear16(x16, w16, b, act=act, precision=1),
            "out32+resid": lambda: F.linear16(x16, w16, b, act=act, resid=r, precision=1)}
    mb = {"out16": (M * K * 2 + M * N * 2) / 1e6, "out32": (M * K * 2 + M * N * 4) / 1e6, "out32+resid": (M * K * 2 + 2 * M * N * 4) / 1e6}
    tags = {k: [t for t, *_ in mi355attn.kernel_trace(f)][0].split(" ")[0] for k, f in cfgs.items()}
    for rnd in range(3):
        print(name, "round", rnd, {k: round(timeit(f), 1) for k, f in cfgs.items()}, "us", flush=True)
    print(name, "MB moved", {k: round(v) for k, v in mb.items()}, "kernels", tags, flush=True)
