import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-attention_amd"))
import torch
from mi355attn import functional as f
torch.manual_seed(197)
for (B, N, h, d) in [(2, 197, 12, 64), (3, 64, 4, 32), (1, 130, 2, 64)]:
    qkv16 = f.cast16(torch.randn(B, N, 3 * h * d).cuda(), 1)
    a1 = f.sdpa16(qkv16, h, d ** -0.5, precision=1); a2 = f.sdpa16(qkv16, h, d ** -0.5, precision=1)
    r1 = f.sdpa(qkv16.float(), h, d ** -0.5, precision=1); r2 = f.sdpa(qkv16.float(), h, d ** -0.5, precision=1)
    print((B, N, h, d), "sdpa16 deterministic:", torch.equal(a1, a2), " sdpa deterministic:", torch.equal(r1, r2))
    diff = (a1.float() - r1.half().float()).abs()
    nz = (diff > 0).sum().item()
    print("   mismatches", nz, "of", diff.numel(), " max abs diff", diff.max().item(), " max |ref|", r1.abs().max().item())
    if nz:
        idx = (diff > 0).nonzero()[:5]
        for i in idx:
            i = tuple(i.tolist())
            print("    at", i, "got", a1[i].item(), "ref32", r1[i].item(), "ref16", r1.half()[i].item())
        # which query rows / heads
        rows = (diff > 0).nonzero()
        print("    n rows", rows[:, 1].unique().numel(), "first rows", rows[:, 1].unique()[:20].tolist(), " heads(cols//d)", (rows[:, 2] // d).unique().tolist()[:12])
