#!/usr/bin/env python
"""Time mi355_linear16_fwd under every tile/schedule variant on the ViT-Base GEMM shapes (B=256) and check that all
variants agree bit-for-bit with variant 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-attention_amd"))
import torch  # noqa: E402
import mi355attn  # noqa: E402
from mi355attn import StreamTimer  # noqa: E402
from mi355attn import functional as F  # noqa: E402

dev = torch.device("cuda", 0)
M = 256 * 197
# (name, N, K, out16, with_resid, gelu)
shapes = [("qkv", 2304, 768, True, False, False), ("proj", 768, 768, False, True, False), ("fc1", 3072, 768, True, False, True),
          ("fc2", 768, 3072, False, True, True), ("fc1_nogelu", 3072, 768, True, False, False)]
if os.environ.get("GEMM_SET") == "narrow":
    # CSWin stage 3 / 4 and XCiT-S (dim 384) at B = 256: (name, N, K, out16, resid, gelu, M)
    shapes = [("s3_qkv", 768, 256, True, False, False, 50176), ("s3_proj", 256, 256, False, True, False, 50176),
              ("s3_fc1", 1024, 256, True, False, True, 50176), ("s3_fc2", 256, 1024, False, True, False, 50176),
              ("s4_qkv", 1536, 512, True, False, False, 12544), ("s4_proj", 512, 512, False, True, False, 12544),
              ("s4_fc1", 2048, 512, True, False, True, 12544), ("s4_fc2", 512, 2048, False, True, False, 12544),
              ("x_qkv", 1152, 384, True, False, False, 50176), ("x_proj", 384, 384, False, True, False, 50176),
              ("x_fc1", 1536, 384, True, False, True, 50176), ("x_fc2", 384, 1536, False, True, False, 50176),
              ("m_fc1", 2048, 512, True, False, True, 50176), ("m_fc2", 512, 2048, False, True, False, 50176),
              ("m_tok1", 256, 256, True, False, True, 131072)]
if os.environ.get("GEMM_SHAPES"):
    shapes = [s_ for s_ in shapes if s_[0] in os.environ["GEMM_SHAPES"].split(",")]
if os.environ.get("GEMM_LONGK"):
    shapes.append(("longK", 2304, 6144, True, False, False))
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,4,5,6,7".split(","))]
res = []
for shp in shapes:
    name, N, K, out16, with_resid, gelu = shp[:6]
    M = shp[6] if len(shp) > 6 else 256 * 197
    torch.manual_seed(0)
    x16 = torch.randn(M, K, device=dev).half()
    w16 = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev) if with_resid else None
    ref = None
    for v in variants:
        mi355attn.set_option("gemm_variant", v)
        act = F.ACT_GELU if gelu else F.ACT_NONE
        try:
            y = F.linear16(x16, w16, b, act=act, resid=resid, out16=out16, precision=1)
        except mi355attn.Mi355Error:
            continue                                         # this variant does not take the shape
        torch.cuda.synchronize()
        same = True if ref is None else bool(torch.equal(y, ref))
        relerr = 0.0 if ref is None else float((y.float() - ref.float()).norm() / ref.float().norm())
        if ref is None:
            ref = y.clone()
        for _ in range(3):
            F.linear16(x16, w16, b, act=act, resid=resid, out16=out16, precision=1)
        torch.cuda.synchronize()
        tm = StreamTimer(dev); tm.start()
        for _ in range(10):
            F.linear16(x16, w16, b, act=act, resid=resid, out16=out16, precision=1)
        ms = tm.stop_ms() / 10
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        res.append(dict(shape=name, N=N, K=K, variant=v, ms=round(ms, 4), tflops=round(tf, 1), same_as_v0=same, rel_vs_v0=relerr))
        print(res[-1], flush=True)
mi355attn.set_option("gemm_variant", 0)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_bench.json"), "w"), indent=1)
