import sys, os, torch
sys.path.insert(0, "pytorch-attention_amd"); sys.path.insert(0, ".")
import mi355attn
from mi355attn import StreamTimer, functional as F
from mi355attn.modules import CSWinBlock
torch.manual_seed(1234)
m = CSWinBlock(64, 56, 2, split_size=1, qkv_bias=True).eval().cuda()
x = torch.randn(256, 3136, 64, device="cuda")
a0, a1 = m.attns
def run():
    return F.cswin_stripe_attention(x, m.norm1, m.qkv, a0.get_v, a1.get_v, 56, 2, 1, a0.scale, 1)
with torch.no_grad():
    for _ in range(3): run()
    torch.cuda.synchronize()
    tm = StreamTimer(torch.device("cuda", 0)); tm.start()
    for _ in range(10): run()
    print("MI355_ABL", os.environ.get("MI355_ABL"), "stripe kernel ms", round(tm.stop_ms() / 10, 4))
