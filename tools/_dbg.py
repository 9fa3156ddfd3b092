import sys, torch
sys.path.insert(0, "pytorch-attention_amd"); sys.path.insert(0, ".")
import mi355attn
from mi355attn import _ffi
from mi355attn.modules import CBAM
def run(prealloc):
    torch.manual_seed(11)
    m = CBAM(64, 16, 7).eval().cuda()
    static_x = torch.randn(6, 64, 28, 28, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    if prealloc:
        with torch.cuda.stream(s), torch.no_grad():
            m(static_x)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    nws = len(_ffi._ws_dedicated)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s), torch.no_grad():
        out = m(static_x)
    print("prealloc", prealloc, "new workspaces during capture:", len(_ffi._ws_dedicated) - nws)
    res = []
    for rep in range(4):
        x = torch.randn(6, 64, 28, 28, device="cuda")
        static_x.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        got = out.clone()
        with torch.no_grad():
            want = m(x)
        res.append(bool(torch.equal(got, want)))
    print("   ", res)
    _ffi._ws_dedicated.clear()
run(True)
run(False)
