import sys, torch
sys.path.insert(0, "pytorch-attention_amd"); sys.path.insert(0, ".")
import mi355attn
from mi355attn.modules import CBAM
torch.manual_seed(11)
shape = (2, 64, 32, 32)
B, C, H, W = shape
m = CBAM(C, 16, 7).eval().cuda()
x = torch.randn(*shape).cuda()
mi355attn.set_option("cbam_single", 0)
with torch.no_grad():
    ref = m(x).clone()
mi355attn.set_option("cbam_single", 1)
with torch.no_grad():
    o = m(x).clone()
torch.cuda.synchronize()
d = (o - ref).abs()
bad = (d > 1e-4 * ref.abs().max()).nonzero()
print("n bad", len(bad), "of", o.numel())
import collections
cnt_c = collections.Counter(bad[:, 1].tolist()); cnt_y = collections.Counter(bad[:, 2].tolist()); cnt_x = collections.Counter(bad[:, 3].tolist())
print("channels", sorted(cnt_c.items()))
print("rows", sorted(cnt_y.items()))
print("cols", sorted(cnt_x.items()))
for r in bad[:12].tolist():
    b, c, yy, xx = r
    print(r, "x", float(x[b, c, yy, xx]), "single", float(o[b, c, yy, xx]), "ref", float(ref[b, c, yy, xx]), "ratio s", float(o[b, c, yy, xx] / x[b, c, yy, xx]), "ratio r", float(ref[b, c, yy, xx] / x[b, c, yy, xx]))
# is the bad value equal to some other x element (misplaced data)?
b, c, yy, xx = bad[0].tolist()
g = o[b, c, yy, xx] / (ref[b, c, yy, xx - 1] / x[b, c, yy, xx - 1])
print("implied x for bad elem using neighbour gate:", float(g))
