#!/usr/bin/env python
"""Streaming-bandwidth yardsticks on the current GPU: read-only sweep and float4 copy vs working-set size.

Shows where the Infinity Cache (256 MiB) / L2 (8 x 4 MiB) stop helping a re-read -- the number that decides whether
chunking the pool->scale passes of the channel-attention family can pay.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-attention_amd"))
import torch  # noqa: E402
from mi355attn import StreamTimer, _ffi  # noqa: E402
from mi355attn import functional as F  # noqa: E402

dev = torch.device("cuda", 0)
sink = torch.zeros(4, device=dev)
res = []
for mb in (8, 16, 32, 64, 128, 192, 256, 384, 512, 822, 1644):
    n = mb * (1 << 20) // 4
    src = torch.randn(n, device=dev)
    dst = torch.empty_like(src)
    lib = _ffi.lib()
    reps = max(5, min(200, 8192 // mb))
    for _ in range(3):
        _ffi.check(lib.mi355_stream_read(_ffi.dptr(src), n * 4, _ffi.dptr(sink), _ffi.stream_ptr(dev)), "read")
    torch.cuda.synchronize()
    tm = StreamTimer(dev); tm.start()
    for _ in range(reps):
        lib.mi355_stream_read(_ffi.dptr(src), n * 4, _ffi.dptr(sink), _ffi.stream_ptr(dev))
    rd = n * 4 / (tm.stop_ms() / reps * 1e-3) / 1e9
    for _ in range(3):
        F.stream_copy(src, dst)
    torch.cuda.synchronize()
    tm = StreamTimer(dev); tm.start()
    for _ in range(reps):
        F.stream_copy(src, dst)
    cp = 2 * n * 4 / (tm.stop_ms() / reps * 1e-3) / 1e9
    # torch's own copy kernel as a cross-check
    tm = StreamTimer(dev); tm.start()
    for _ in range(reps):
        dst.copy_(src)
    tcp = 2 * n * 4 / (tm.stop_ms() / reps * 1e-3) / 1e9
    res.append(dict(MiB=mb, read_GBps=round(rd, 1), copy_GBps=round(cp, 1), torch_copy_GBps=round(tcp, 1)))
    print(res[-1], flush=True)
    del src, dst
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "mem_bw.json"), "w"), indent=1)
