"""GPU sanity run: the single-read exchange kernels (SE / CBAM / ECA) at batch sizes far below the number of resident workgroups
(B = 1, 3, 7, 33) and four shapes, against the oracle.  Polls are bounded, so a broken hand-off reports an error instead of hanging.
    python tools/edge_batches.py"""
import sys, torch
sys.path.insert(0, "pytorch-attention_amd"); sys.path.insert(0, ".")
import oracle as O
from mi355attn.modules import SELayer, CBAM, ECALayer, GCT, LCT
torch.manual_seed(0)
for B in (1, 3, 7, 33):
    for (C, H, W) in ((256, 56, 56), (64, 32, 32), (512, 14, 14), (96, 28, 28)):
        x = torch.randn(B, C, H, W)
        for name, ctor in (("se", lambda: SELayer(C)), ("cbam", lambda: CBAM(C)), ("eca", lambda: ECALayer(C))):
            torch.manual_seed(1)
            m = ctor().eval()
            sd = m.state_dict()
            if name == "se": ref = O.se_forward(x, sd["fc.0.weight"], sd["fc.2.weight"])
            elif name == "cbam": ref = O.cbam_forward(x, sd["ca.fc.0.weight"], sd["ca.fc.2.weight"], sd["sa.conv.weight"])
            else: ref = O.eca_forward(x, sd["conv.weight"])
            with torch.no_grad():
                y = m.cuda()(x.cuda()).cpu()
            err = float((y - ref).norm() / ref.norm())
            assert err < 1e-5, (name, B, C, H, W, err)
    print("B", B, "ok", flush=True)
print("edge ok")
