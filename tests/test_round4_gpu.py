"""Round-4 boundary items ("Next round" 7 and the advisor's findings):

  * GCT / LCT / Gaussian GCT are recorded under hipGraph capture with their single-read exchange kernels (launch tag and ticket in the
    workspace, like SE / CBAM since round 3): replays, back-to-back replays and interleaved eager launches, bit-equal;
  * a kernel whose dynamic LDS depends on the shape (DoubleAttention(64,32,32): HW * 64 + 43 KB) runs a small image first and a large
    one afterwards (the attribute used to stay at the first value);
  * PAM at DANet's own width (dim = 512 > the attention kernel's widest head) against the oracle;
  * XCiT(drop_rate > 0) is the reference's eval-mode identity, and refused in train mode;
  * guarded_forward: a saturated intermediate of a fused kernel (non-finite output for finite input) triggers the strict re-run, and the
    re-run overrides sub-modules that were built with an explicit 16-bit precision;
  * the fp16 range guard still reports a saturating value when the same lane has seen an input inf.
"""
import warnings

import pytest
import torch

import oracle as O
from conftest import assert_parity

pytestmark = pytest.mark.gpu


def test_gct_lct_record_their_exchange_kernels_under_graph_capture():
    import mi355attn
    from mi355attn.modules import GCT, GaussianGCT, LCT
    torch.manual_seed(3)
    gct, lct, gg = GCT(64), LCT(64, 8), GaussianGCT(64)
    gct1 = GCT(64, mode="l1")
    with torch.no_grad():
        gct.gamma.add_(0.5); gct.beta.add_(0.1); gct1.gamma.add_(0.3)
        lct.w.mul_(1.5); lct.b.add_(0.2)
    mods = [m.cuda() for m in (gct, gct1, lct, gg)]
    static_x = torch.randn(6, 64, 28, 28, device="cuda")
    with torch.no_grad():
        for m in mods:
            m(static_x)                                    # loads the code objects, makes the eager workspaces known
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        outs = [m(static_x) for m in mods]
    for rep in range(4):
        x = torch.randn(6, 64, 28, 28, device="cuda")
        static_x.copy_(x)
        g.replay()
        if rep == 2:
            g.replay()                                     # two replays back to back: the epoch advances inside the graph
        torch.cuda.synchronize()
        got = [o.clone() for o in outs]
        with torch.no_grad():
            want = [m(x) for m in mods]                    # eager launches in between
        for a, b, m in zip(got, want, mods):
            assert torch.equal(a, b), f"replay {rep}: {type(m).__name__} differs from the eager launch"
    # the single-read kernels really were what ran: they must agree with the two-pass kernels only to fp32 noise, not bit for bit in general
    mi355attn.set_option("zoo_single", 0)
    try:
        with torch.no_grad():
            two = [m(x) for m in mods]
    finally:
        mi355attn.set_option("zoo_single", 1)
    for a, b, m in zip(got, two, mods):
        assert_parity(a.cpu(), b.cpu(), 2e-6, f"{type(m).__name__} single-read (replayed) vs two-pass")
    xc = x.cpu()
    assert_parity(got[0].cpu(), O.gct_forward(xc, gct.alpha.cpu(), gct.gamma.cpu(), gct.beta.cpu(), 1e-5, "l2"), 1e-5, "GCT l2 replay vs oracle")
    assert_parity(got[2].cpu(), O.lct_forward(xc, lct.w.cpu(), lct.b.cpu(), 8), 1e-5, "LCT replay vs oracle")
    assert_parity(got[3].cpu(), O.gct_gauss_forward(xc), 1e-5, "Gaussian GCT replay vs oracle")
    mi355attn.sync_status(wait=True)


def test_dynamic_lds_limit_follows_the_largest_request():
    """double_attn_small asks for HW * 64 + 43 008 bytes of dynamic LDS: 16 x 16 first (59 KB), then 32 x 32 (109 KB) in the same process."""
    from mi355attn.modules import DoubleAttention
    torch.manual_seed(1234)
    m = DoubleAttention(64, 32, 32).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    keys = ("convA.weight", "convA.bias", "convB.weight", "convB.bias", "convV.weight", "convV.bias", "proj.weight", "proj.bias")
    m = m.cuda()
    for hw in (16, 32, 24, 32):
        torch.manual_seed(hw)
        x = torch.randn(4, 64, hw, hw)
        with torch.no_grad():
            y = m(x.cuda())
        torch.cuda.synchronize()
        ref = O.double_attention_forward(x, *[sd[k] for k in keys])
        assert_parity(y.cpu(), ref, 1e-3, f"DoubleAttention(64,32,32) at {hw}x{hw}")


@pytest.mark.parametrize("dim,hw", [(512, 24), (384, 16), (320, 16)])
def test_pam_wider_than_the_attention_kernel(dim, hw):
    from mi355attn.modules import PAM
    torch.manual_seed(1234)
    m = PAM(dim).eval()
    with torch.no_grad():
        m.alpha.fill_(0.7)
        for c in (m.b, m.c, m.d):
            c.weight.mul_(0.5)                                  # keeps the unscaled logits (a sum over `dim` channels) in a sane range
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.manual_seed(4321)
    x = torch.randn(2, dim, hw, hw) * 0.5
    ref = O.pam_forward(x, sd, dtype=torch.float64).float()
    with torch.no_grad():
        y = m.cuda()(x.cuda())
    torch.cuda.synchronize()
    assert_parity(y.cpu(), ref, 1e-3, f"PAM({dim}) at {hw}x{hw}")


def test_xcit_dropout_rates_are_eval_identities():
    from torch import nn
    from mi355attn.modules.xcit import XCiT
    kw = dict(img_size=64, patch_size=16, embed_dim=128, depth=2, num_heads=4, eta=1.0, cls_attn_layers=1, norm_layer=nn.LayerNorm, num_classes=10)
    torch.manual_seed(1234)
    plain = XCiT(**kw).eval().cuda()
    torch.manual_seed(1234)
    dropped = XCiT(drop_rate=0.1, attn_drop_rate=0.2, **kw).eval().cuda()
    assert list(plain.state_dict().keys()) == list(dropped.state_dict().keys())
    x = torch.randn(3, 3, 64, 64, device="cuda")
    with torch.no_grad():
        a, b = plain(x), dropped(x)
    assert torch.equal(a, b)
    dropped.train()
    with pytest.raises(RuntimeError, match="eval"):
        dropped(x)
    with pytest.raises(ValueError):
        XCiT(drop_rate=1.5, **kw)


def test_guarded_forward_catches_a_saturated_fused_intermediate():
    """CSWinBlock stage 1 runs LayerNorm + fc1 + GELU + fc2 in ONE kernel whose 16-bit hidden activations never leave the registers
    (no range word to report into).  With fc1 scaled so that the hidden units pass 65504 the block's output is inf / NaN for a finite
    input: guarded_forward must re-run in strict mode -- including the sub-modules built with an explicit precision=1."""
    import mi355attn
    from mi355attn.modules import CSWinBlock
    torch.manual_seed(1234)
    blk = CSWinBlock(64, 56, 2, split_size=1, qkv_bias=True, precision=1).eval()
    with torch.no_grad():
        blk.mlp.fc1.weight.mul_(3.0e4)                         # hidden pre-activations ~ 3e4 * O(1): beyond the fp16 range
        blk.mlp.fc2.weight.mul_(1.0e-4)                        # keeps the fp32 reference finite and moderate
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    torch.manual_seed(4321)
    x = torch.randn(2, 3136, 64)
    ref = O.cswin_block_forward(x, sd, 56, 2, 1)
    assert torch.isfinite(ref).all()
    blk = blk.cuda()
    with torch.no_grad():
        y_fast = blk(x.cuda())
    torch.cuda.synchronize()
    try:
        mi355attn.range_status(wait=True)
    except mi355attn.Mi355RangeError:
        pass
    assert not torch.isfinite(y_fast).all(), "the construction no longer saturates the fused kernel: strengthen it"
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        with torch.no_grad():
            y = mi355attn.guarded_forward(blk, x.cuda())
    assert any("strict mode" in str(w.message) for w in rec)
    assert torch.isfinite(y).all()
    assert_parity(y.cpu(), ref, 2e-4, "CSWinBlock s1 after the strict re-run")
    assert blk.precision == 1 and blk.mlp.precision == 1, "explicit precisions must be restored after the re-run"


def test_range_guard_sees_a_saturation_next_to_an_input_inf():
    """rg_absmax skips a group of four that holds an inf (the input's own, not reported) instead of poisoning the lane's running maximum:
    a later finite value >= 65520 in the same lane is still reported."""
    import mi355attn
    from mi355attn import functional as F
    try:
        mi355attn.range_status(wait=True)
    except mi355attn.Mi355RangeError:
        pass
    x = torch.zeros(1 << 16, device="cuda")
    x[0] = float("inf")                                        # the first group of four of lane 0 ...
    F.cast16(x, 1)
    mi355attn.range_status(wait=True)                          # ... is the input's own inf: no report
    n = x.numel()
    x[n - 4096:] = 7.0e4                                       # finite values that saturate, later in the same grid-stride walks
    F.cast16(x, 1)
    with pytest.raises(mi355attn.Mi355RangeError):
        mi355attn.range_status(wait=True)
