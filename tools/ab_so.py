#!/usr/bin/env python
"""Same-process A/B of two builds of libmi355attn.so on one op (box-to-box variance on the pool is 10-20 %, so cross-run comparisons of
small changes are meaningless): `python tools/ab_so.py sdpa16|se|cbam|eca|lpi|lnlpi|lnt|stripe1|stripe2|pmlp1|pmlp2|fc1|qkv|proj|fc2 [baseline.so]`.
The baseline library defaults to tools/bin/libmi355attn_base.so (a copy of an earlier build)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-attention_amd"))
import torch  # noqa: E402
import mi355attn  # noqa: E402

vp, ci, cf, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
op = sys.argv[1] if len(sys.argv) > 1 else "sdpa16"
base = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tools", "bin", "libmi355attn_base.so")
libs = {"base": ctypes.CDLL(base), "new": ctypes.CDLL(mi355attn.LIB_PATH)}
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


if op == "sdpa16":
    qkv = torch.randn(256, 197, 2304, device=dev).half()
    outs = {}
    def mk(lib, out):
        lib.mi355_sdpa16_fwd.restype = ci
        lib.mi355_sdpa16_fwd.argtypes = [vp, vp, ci, ci, ci, ci, cf, ci, vp]
        return lambda: lib.mi355_sdpa16_fwd(qkv.data_ptr(), out.data_ptr(), 256, 197, 12, 64, 0.125, 1, st)
    fns = {}
    for k, lib in libs.items():
        outs[k] = torch.empty(256, 197, 768, device=dev, dtype=torch.float16)
        fns[k] = mk(lib, outs[k])
elif op in ("se", "cbam", "eca"):
    x = torch.randn(256, 256, 56, 56, device=dev)
    w1, w2 = torch.randn(16, 256, device=dev) / 16, torch.randn(256, 16, device=dev) / 4
    wc = torch.randn(1, 2, 7, 7, device=dev) / 7
    taps = torch.randn(5, device=dev)
    outs, fns, wss = {}, {}, {}
    for k, lib in libs.items():
        outs[k] = torch.empty_like(x)
        lib.mi355_set_option.argtypes = [ctypes.c_char_p, ctypes.c_long]
        lib.mi355_set_option(b"ws_persistent", 1)
        if op == "se":
            lib.mi355_se_workspace_bytes.restype = sz; lib.mi355_se_workspace_bytes.argtypes = [ci] * 4
            lib.mi355_se_fwd.restype = ci; lib.mi355_se_fwd.argtypes = [vp, vp, vp, vp] + [ci] * 5 + [vp, sz, vp]
            n = lib.mi355_se_workspace_bytes(256, 256, 56, 56)
            wss[k] = torch.zeros(n, dtype=torch.uint8, device=dev)
            fns[k] = (lambda lib=lib, k=k, n=n: lib.mi355_se_fwd(x.data_ptr(), w1.data_ptr(), w2.data_ptr(), outs[k].data_ptr(), 256, 256, 16, 56, 56, wss[k].data_ptr(), n, st))
        elif op == "cbam":
            lib.mi355_cbam_workspace_bytes.restype = sz; lib.mi355_cbam_workspace_bytes.argtypes = [ci] * 4
            lib.mi355_cbam_fwd.restype = ci; lib.mi355_cbam_fwd.argtypes = [vp] * 5 + [ci] * 7 + [vp, sz, vp]
            n = lib.mi355_cbam_workspace_bytes(256, 256, 56, 56)
            wss[k] = torch.zeros(n, dtype=torch.uint8, device=dev)
            fns[k] = (lambda lib=lib, k=k, n=n: lib.mi355_cbam_fwd(x.data_ptr(), w1.data_ptr(), w2.data_ptr(), wc.data_ptr(), outs[k].data_ptr(), 256, 256, 16, 7, 56, 56, 0, wss[k].data_ptr(), n, st))
        else:
            lib.mi355_eca_workspace_bytes.restype = sz; lib.mi355_eca_workspace_bytes.argtypes = [ci] * 4
            lib.mi355_eca_fwd.restype = ci; lib.mi355_eca_fwd.argtypes = [vp, vp, vp] + [ci] * 5 + [vp, sz, vp]
            n = lib.mi355_eca_workspace_bytes(256, 256, 56, 56)
            wss[k] = torch.zeros(max(n, 256), dtype=torch.uint8, device=dev)
            fns[k] = (lambda lib=lib, k=k, n=n: lib.mi355_eca_fwd(x.data_ptr(), taps.data_ptr(), outs[k].data_ptr(), 256, 256, 5, 56, 56, wss[k].data_ptr(), n, st))
elif op == "lpi":
    B, H, W, C = 256, 14, 14, 384
    x = torch.randn(B, H * W, C, device=dev)
    w1, w2 = torch.randn(C, 1, 3, 3, device=dev) / 3, torch.randn(C, 1, 3, 3, device=dev) / 3
    b1, b2, bw, bb, bm = (torch.randn(C, device=dev) * 0.1 for _ in range(5))
    bv = torch.rand(C, device=dev) + 0.5
    gm = torch.rand(C, device=dev)
    outs, fns, wss = {}, {}, {}
    for k, lib in libs.items():
        outs[k] = torch.empty_like(x)
        lib.mi355_lpi_fwd.restype = ci
        lib.mi355_lpi_fwd.argtypes = [vp] * 7 + [cf] + [vp] * 5 + [ci] * 4 + [vp, sz, vp]
        wss[k] = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
        fns[k] = (lambda lib=lib, k=k: lib.mi355_lpi_fwd(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), bw.data_ptr(), bb.data_ptr(), bm.data_ptr(), bv.data_ptr(),
                                                          1e-5, w2.data_ptr(), b2.data_ptr(), gm.data_ptr(), x.data_ptr(), outs[k].data_ptr(), B, H, W, C,
                                                          wss[k].data_ptr(), 1 << 20, st))
elif op == "lnlpi":
    # LayerNorm statistics + LPI (mi355_ln_lpi_fwd) at the XCABlock shape of the bench step
    B, H, W, C = 256, 14, 14, 384
    x = torch.randn(B, H * W, C, device=dev)
    w1, w2 = torch.randn(C, 1, 3, 3, device=dev) / 3, torch.randn(C, 1, 3, 3, device=dev) / 3
    b1, b2, bw, bb, bm, lw, lb = (torch.randn(C, device=dev) * 0.1 for _ in range(7))
    bv = torch.rand(C, device=dev) + 0.5
    gm = torch.rand(C, device=dev)
    outs, fns, wss = {}, {}, {}
    for k, lib in libs.items():
        outs[k] = torch.empty_like(x)
        lib.mi355_ln_lpi_fwd.restype = ci
        lib.mi355_ln_lpi_fwd.argtypes = [vp] * 3 + [cf] + [vp] * 6 + [cf] + [vp] * 5 + [ci] * 4 + [vp, sz, vp]
        wss[k] = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
        fns[k] = (lambda lib=lib, k=k: lib.mi355_ln_lpi_fwd(x.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-5, w1.data_ptr(), b1.data_ptr(), bw.data_ptr(), bb.data_ptr(),
                                                             bm.data_ptr(), bv.data_ptr(), 1e-5, w2.data_ptr(), b2.data_ptr(), gm.data_ptr(), x.data_ptr(),
                                                             outs[k].data_ptr(), B, H, W, C, wss[k].data_ptr(), 1 << 20, st))
elif op == "lnt":
    # the Mixer's transposed 16-bit LayerNorm (mi355_layernorm16_t_fwd) at B = 256, 196 tokens, 512 channels
    B, N, C, NP = 256, 196, 512, 224
    x = torch.randn(B, N, C, device=dev)
    lw, lb = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    outs, fns = {}, {}
    for k, lib in libs.items():
        outs[k] = torch.zeros(B, C, NP, device=dev, dtype=torch.float16)
        lib.mi355_layernorm16_t_fwd.restype = ci
        lib.mi355_layernorm16_t_fwd.argtypes = [vp] * 4 + [ci] * 4 + [cf, ci, vp]
        fns[k] = (lambda lib=lib, k=k: lib.mi355_layernorm16_t_fwd(x.data_ptr(), lw.data_ptr(), lb.data_ptr(), outs[k].data_ptr(), B, N, C, NP, 1e-5, 1, st))
elif op in ("stripe1", "stripe2"):
    # first half of a CSWinBlock (mi355_cswin_stripe_attn_fwd) at the stage-1 / stage-2 shapes of the bench step
    C, reso, hb, split = (64, 56, 1, 1) if op == "stripe1" else (128, 28, 2, 2)
    B = 256
    x = torch.randn(B, reso * reso, C, device=dev)
    w = (torch.randn(3 * C, C, device=dev) / C ** 0.5).half()
    bq = torch.randn(3 * C, device=dev) * 0.1
    gw = [torch.randn(C // 2, 1, 3, 3, device=dev) / 3 for _ in range(2)]
    gb = [torch.randn(C // 2, device=dev) * 0.1 for _ in range(2)]
    outs, fns = {}, {}
    for k, lib in libs.items():
        outs[k] = torch.empty(B, reso * reso, C, device=dev, dtype=torch.float16)
        lib.mi355_cswin_stripe_attn_fwd.restype = ci
        lib.mi355_cswin_stripe_attn_fwd.argtypes = [vp] * 8 + [ci] * 5 + [cf, cf, ci, vp]
        fns[k] = (lambda lib=lib, k=k: lib.mi355_cswin_stripe_attn_fwd(x.data_ptr(), w.data_ptr(), bq.data_ptr(), gw[0].data_ptr(), gb[0].data_ptr(),
                                                                       gw[1].data_ptr(), gb[1].data_ptr(), outs[k].data_ptr(), B, reso, C, hb, split,
                                                                       32 ** -0.5, 1e-5, 1, st))
elif op in ("pmlp1", "pmlp2"):
    # second half of a CSWinBlock (mi355_proj_mlp_fused_fwd)
    C = 64 if op == "pmlp1" else 128
    HD = 4 * C
    M = 256 * (3136 if C == 64 else 784)
    x = torch.randn(M, C, device=dev)
    ctx = torch.randn(M, C, device=dev).half()
    wp = (torch.randn(C, C, device=dev) / C ** 0.5).half()
    w1 = (torch.randn(HD, C, device=dev) / C ** 0.5).half()
    w2 = (torch.randn(C, HD, device=dev) / HD ** 0.5).half()
    if C == 128:
        w2 = w2.reshape(C, HD // 32, 32).permute(1, 0, 2).contiguous()
    bp, b1, b2 = torch.randn(C, device=dev) * 0.1, torch.randn(HD, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
    outs, fns = {}, {}
    for k, lib in libs.items():
        outs[k] = torch.empty(M, C, device=dev)
        lib.mi355_proj_mlp_fused_fwd.restype = ci
        lib.mi355_proj_mlp_fused_fwd.argtypes = [vp] * 10 + [ctypes.c_long, ci, ci, ci, cf, ci, vp]
        fns[k] = (lambda lib=lib, k=k: lib.mi355_proj_mlp_fused_fwd(x.data_ptr(), ctx.data_ptr(), wp.data_ptr(), bp.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                                                    w2.data_ptr(), b2.data_ptr(), None, outs[k].data_ptr(), M, C, HD, 1, 1e-5, 1, st))
elif op.startswith("ln") and op[2:].isdigit():
    # 16-bit LayerNorm (mi355_layernorm16_fwd) on 50 176 rows (256 x 196 tokens) of the given width: ln256 (CSWin s3), ln384 (XCiT), ln512 (Mixer), ln768
    rows, cols = 256 * (197 if op == "ln768" else 196), int(op[2:])
    x = torch.randn(rows, cols, device=dev)
    lw, lb = torch.rand(cols, device=dev) + 0.5, torch.randn(cols, device=dev) * 0.1
    outs, fns = {}, {}
    for k, lib in libs.items():
        outs[k] = torch.empty(rows, cols, device=dev, dtype=torch.float16)
        lib.mi355_layernorm16_fwd.restype = ci
        lib.mi355_layernorm16_fwd.argtypes = [vp] * 4 + [ci, ci, cf, ci, vp]
        fns[k] = (lambda lib=lib, k=k: lib.mi355_layernorm16_fwd(x.data_ptr(), lw.data_ptr(), lb.data_ptr(), outs[k].data_ptr(), rows, cols, 1e-5, 1, st))
elif op in ("fc1", "qkv", "proj", "fc2"):
    # the four GEMMs of a ViT-Base layer at B = 256 (mi355_linear16_ws_fwd): fc1 = bias + GELU, 16-bit out; fc2 / proj = fp32 + residual
    M = 256 * 197
    N, K, act, o16, res = {"fc1": (3072, 768, 1, 1, 0), "qkv": (2304, 768, 0, 1, 0), "proj": (768, 768, 0, 0, 1), "fc2": (768, 3072, 0, 0, 1)}[op]
    x16 = torch.randn(M, K, device=dev).half()
    w16 = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.randn(N, device=dev) * 0.1
    r = torch.randn(M, N, device=dev) if res else None
    outs, fns, wss = {}, {}, {}
    for k, lib in libs.items():
        outs[k] = torch.empty(M, N, device=dev, dtype=torch.float16 if o16 else torch.float32)
        lib.mi355_linear16_workspace_bytes.restype = sz; lib.mi355_linear16_workspace_bytes.argtypes = [ci] * 3
        n = lib.mi355_linear16_workspace_bytes(M, N, K)
        wss[k] = torch.zeros(max(n, 16), dtype=torch.uint8, device=dev)
        lib.mi355_linear16_ws_fwd.restype = ci
        lib.mi355_linear16_ws_fwd.argtypes = [vp] * 6 + [ci] * 8 + [vp, sz, vp]
        fns[k] = (lambda lib=lib, k=k, n=n: lib.mi355_linear16_ws_fwd(x16.data_ptr(), w16.data_ptr(), b.data_ptr(), None, r.data_ptr() if res else None,
                                                                       outs[k].data_ptr(), M, N, K, K, N, act, o16, 1, wss[k].data_ptr(), n, st))
else:
    raise SystemExit("unknown op " + op)

for k, f in fns.items():
    rc = f()
    assert rc == 0, (k, rc)
torch.cuda.synchronize()
for rnd in range(4):
    print("round", rnd, {k: round(timeit(f), 1) for k, f in fns.items()}, "us", flush=True)
d = (outs["base"].float() - outs["new"].float()).abs()
print("max abs diff", float(d.max()), "fraction of elements differing", float((d > 0).float().mean()))
