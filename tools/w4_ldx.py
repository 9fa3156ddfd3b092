#!/usr/bin/env python
"""TEMPORARY: do the row strides of the 16-bit operands matter (L2 channel mapping of the 8-row DMA pieces)?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-attention_amd"))
import torch
import mi355attn
from mi355attn import StreamTimer
from mi355attn._ffi import lib, dptr, stream_ptr
libc = ctypes.CDLL(None)
dev = torch.device("cuda", 0)
M = 256 * 197
# name, N, K, out16, gelu, resid
for name, N, K, out16, gelu, res in (("qkv", 2304, 768, 1, 0, 0), ("fc1", 3072, 768, 1, 1, 0), ("proj", 768, 768, 0, 0, 1), ("fc2", 768, 3072, 0, 0, 1)):
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, dtype=torch.float16 if out16 else torch.float32, device=dev)
    resid = torch.randn(M, N, device=dev) if res else None
    nws = lib().mi355_linear16_workspace_bytes(M, N, K)
    ws = torch.zeros(max(nws, 16), dtype=torch.uint8, device=dev)
    for pa_, pw in ((0, 0), (128, 0), (0, 128), (128, 128), (64, 64), (192, 192)):
        xb = torch.randn(M, K + pa_, device=dev).half()
        wb = (torch.randn(N, K + pw, device=dev) / K ** 0.5).half()
        libc.setenv(b"MI355_LDW", str(pw).encode(), 1)
        for v in ((0, 17) if out16 and not gelu else (0,)):
            mi355attn.set_option("gemm_variant", v)
            def go():
                rc = lib().mi355_linear16_ws_fwd(dptr(xb), dptr(wb), dptr(b), None, dptr(resid), dptr(y), M, N, K, K + pa_, N, 1 if gelu else 0, out16, 1, dptr(ws), nws, stream_ptr(dev))
                assert rc == 0, rc
            for _ in range(3):
                go()
            torch.cuda.synchronize()
            tm = StreamTimer(dev); tm.start()
            for _ in range(10):
                go()
            ms = tm.stop_ms() / 10
            print("%s lda = K + %3d ldw = K + %3d variant %2d: %.4f ms  %.0f TF" % (name, pa_, pw, v, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
