import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "pytorch-attention_amd")); sys.path.insert(0, R)
import mi355attn
from mi355attn import functional as F
from mi355attn.modules import DoubleAttention
torch.manual_seed(0)
m = DoubleAttention(256, 128, 128).eval().cuda()
x = torch.randn(256, 256, 56, 56, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
with torch.no_grad():
    print("fused   ms", t(lambda: m(x)))
    mi355attn.set_option("da_fused", 0)
    print("unfused ms", t(lambda: m(x)))
    mi355attn.set_option("da_fused", 1)
