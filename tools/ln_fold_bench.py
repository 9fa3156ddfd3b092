"""Per-piece cost of the LayerNorm fold at the ViT-Base shape (M = 50432 rows, C = 768): what each LayerNorm launch costs, what the
producers (proj / fc2 with the emitting epilogue) and the consumers (qkv / fc1 with the row-scale epilogue) pay, and the finalize
kernel -- HIP events on the launch stream, interleaved A/B in one process.    python tools/ln_fold_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-attention_amd"))
import torch
import mi355attn
from mi355attn import StreamTimer, functional as F

dev = torch.device("cuda", 0)
M, C, H = 256 * 197, 768, 3072
torch.manual_seed(0)
x = torch.randn(M, C, device=dev)
ln = torch.nn.LayerNorm(C).to(dev)
ctx16 = torch.randn(M, C, device=dev).half()
h16 = torch.randn(M, H, device=dev).half()
wproj = (torch.randn(C, C, device=dev) / C ** 0.5).half()
wfc2 = (torch.randn(C, H, device=dev) / H ** 0.5).half()
bias = torch.randn(C, device=dev) * 0.1
qkv = torch.nn.Linear(C, 3 * C).to(dev)
fc1 = torch.nn.Linear(C, H).to(dev)
cvec = x.mean(-1).contiguous()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tm = StreamTimer(dev)
    tm.start()
    for _ in range(reps):
        fn()
    return tm.stop_ms() / reps * 1e3


with torch.no_grad():
    st = F.ln_center16(x, 1e-5, 1)
    wq, sq, bq = F.lnfold_weights(ln, qkv, 1)
    w1, s1, b1 = F.lnfold_weights(ln, fc1, 1)
    u16 = F.layernorm16(x, ln.weight, ln.bias, ln.eps, 1)
    res = {}
    for rnd in range(2):
        res.setdefault("layernorm16", []).append(timeit(lambda: F.layernorm16(x, ln.weight, ln.bias, ln.eps, 1)))
        res.setdefault("ln_center16", []).append(timeit(lambda: F.ln_center16(x, 1e-5, 1)))
        res.setdefault("proj plain", []).append(timeit(lambda: F.linear16(ctx16, wproj, bias, resid=x, precision=1)))
        res.setdefault("proj emit+finalize", []).append(timeit(lambda: F.linear16_emit(ctx16, wproj, bias, x, cvec, 1e-5, precision=1)))
        res.setdefault("fc2 plain (GELU)", []).append(timeit(lambda: F.linear16(h16, wfc2, bias, act=F.ACT_GELU, resid=x, precision=1)))
        res.setdefault("fc2 emit+finalize (GELU)", []).append(timeit(lambda: F.linear16_emit(h16, wfc2, bias, x, cvec, 1e-5, act=F.ACT_GELU, precision=1)))
        res.setdefault("fc2 plain (no act)", []).append(timeit(lambda: F.linear16(h16, wfc2, bias, resid=x, precision=1)))
        res.setdefault("fc2 emit+finalize (no act)", []).append(timeit(lambda: F.linear16_emit(h16, wfc2, bias, x, cvec, 1e-5, precision=1)))
        res.setdefault("qkv plain", []).append(timeit(lambda: F.linear16(u16, F.weight16(qkv.weight, 1), qkv.bias, out16=True, precision=1)))
        res.setdefault("qkv fold", []).append(timeit(lambda: F.linear16_lnfold(st, wq, bq, sq, precision=1)))
        res.setdefault("fc1 plain (GELU)", []).append(timeit(lambda: F.linear16(u16, F.weight16(fc1.weight, 1), fc1.bias, act=F.ACT_GELU, out16=True, precision=1)))
        res.setdefault("fc1 fold (GELU)", []).append(timeit(lambda: F.linear16_lnfold(st, w1, b1, s1, act=F.ACT_GELU, precision=1)))
    # the finalize kernel alone
    stats = torch.zeros(mi355attn.lib().mi355_ln_fold_stats_bytes(M, C) // 4, device=dev)
    rowtau = torch.empty(M, 2, device=dev)
    f = mi355attn._ffi
    res["finalize alone"] = [timeit(lambda: f.check(f.lib().mi355_ln_finalize_fwd(f.dptr(stats), f.dptr(x), f.dptr(st.a16), f.dptr(cvec), f.dptr(rowtau),
                                                                                    M, C, 1e-5, 1e30, 1, f.dptr(None), f.stream_ptr(dev)), "fin"))]
for k, v in res.items():
    print("%-28s %s us" % (k, " / ".join("%.1f" % t for t in v)))
