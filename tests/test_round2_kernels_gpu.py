"""Op-level checks of the kernels added in round 2, below the module-level golden / full-size tests:

  * persistent 256x256 GEMM (gemm16_p8.hip): ragged M / N edges, one tile, fewer tiles than CUs, many rounds, every epilogue
    combination, bf16, bit identity with the round-1 kernel where no K split applies, split-K last round against fp64;
  * fused first half of a CSWinBlock (cswin_fused.hip) on geometries beyond the two model stages (short windows, one row of
    windows, bf16) against the oracle's LayerNorm -> qkv -> LePE attention;
  * fused second half (proj + residual + LN + MLP + residual) against the unfused composition of the same library ops and fp64.
"""
import pytest
import torch

import oracle as O
from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _ref_linear(x16, w16, b, act, gamma, resid):
    y = x16.double() @ w16.double().t()
    if b is not None:
        y = y + b.double()
    if act:
        y = torch.nn.functional.gelu(y)
    if gamma is not None:
        y = y * gamma.double()
    if resid is not None:
        y = y + resid.double()
    return y


P8_SHAPES = [  # (M, N, K): all have >= one full round of 256x256 tiles on a 256-CU part unless forced through gemm_variant 15
    (50432, 2304, 768), (50432, 768, 768), (12544, 1536, 512), (50176, 1152, 384),
    (65536 + 40, 256 + 8, 256),          # ragged in both directions, 2 tile columns, the second almost empty
    (300, 264, 128), (256, 256, 64), (1, 8, 64), (4097, 520, 192),
]


@pytest.mark.parametrize("prec,dt", [(1, torch.float16), (2, torch.bfloat16)])
@pytest.mark.parametrize("M,N,K", P8_SHAPES)
def test_persistent_gemm_matches_round1_kernel_and_fp64(M, N, K, prec, dt):
    """gemm_variant 15 forces the persistent kernel on any shape; with the K split off it must be bit-identical to variant 7
    (same products, same K order) for every epilogue, and both must sit on the fp64 product of the rounded operands."""
    import mi355attn
    from mi355attn import functional as F
    torch.manual_seed(M + N + K)
    x16 = torch.randn(M, K, device="cuda").to(dt)
    w16 = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
    b = torch.randn(N, device="cuda")
    gamma = torch.rand(N, device="cuda") + 0.5
    resid = torch.randn(M, N, device="cuda")
    combos = [dict(bias=b, out16=True), dict(bias=b, act=F.ACT_GELU, out16=True), dict(bias=b, resid=resid), dict(bias=None, gamma=gamma, resid=resid, act=F.ACT_GELU)]
    try:
        mi355attn.set_option("gemm_splitk", 0)
        for kw in combos:
            mi355attn.set_option("gemm_variant", 7)
            y7 = F.linear16(x16, w16, precision=prec, **kw)
            mi355attn.set_option("gemm_variant", 15)
            y15 = F.linear16(x16, w16, precision=prec, **kw)
            y15b = F.linear16(x16, w16, precision=prec, **kw)
            assert torch.equal(y15, y15b), "run-to-run"
            assert torch.equal(y7, y15), f"persistent kernel differs from variant 7 with {sorted(kw)}"
        if M * N <= 3_000_000:
            ref = _ref_linear(x16.cpu(), w16.cpu(), b.cpu(), True, None, None)
            assert_parity(F.linear16(x16, w16, b, act=F.ACT_GELU, precision=prec).cpu(), ref.float(), 2e-5 if prec == 1 else 2e-5, "fp64 product")
    finally:
        mi355attn.set_option("gemm_variant", 0)
        mi355attn.set_option("gemm_splitk", 1)


@pytest.mark.parametrize("M,N,K", [(50432, 768, 3072), (12544, 512, 2048), (20000, 1024, 1536), (770, 520, 4096)])
def test_persistent_gemm_split_last_round(M, N, K):
    """K >= 1536 with a partially filled last round: the left-over tiles are cut along K (partial sums through the workspace).  The
    result equals the unsplit one to fp32 summation-order noise, is deterministic, and sits on the fp64 product (sampled rows)."""
    import mi355attn
    from mi355attn import functional as F
    torch.manual_seed(7)
    x16 = torch.randn(M, K, device="cuda").half()
    w16 = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda")
    resid = torch.randn(M, N, device="cuda")
    try:
        mi355attn.set_option("gemm_variant", 15)
        mi355attn.set_option("gemm_splitk", 1)
        ys = F.linear16(x16, w16, b, resid=resid, precision=1)
        ys2 = F.linear16(x16, w16, b, resid=resid, precision=1)
        mi355attn.set_option("gemm_splitk", 0)
        yu = F.linear16(x16, w16, b, resid=resid, precision=1)
    finally:
        mi355attn.set_option("gemm_variant", 0)
        mi355attn.set_option("gemm_splitk", 1)
    assert torch.equal(ys, ys2), "split-K result is not deterministic"
    assert_parity(ys.cpu(), yu.cpu(), 2e-6, "split vs unsplit")
    rows = torch.tensor([0, 1, M // 3, M // 2 + 5, M - 2, M - 1])
    ref = _ref_linear(x16[rows].cpu(), w16.cpu(), b.cpu(), False, None, resid[rows].cpu())
    assert_parity(ys[rows].cpu(), ref.float(), 2e-6, "fp64 product, sampled rows")


STRIPES = [  # (C, reso, heads, split, B): tokens per stripe = reso * split <= 64
    (64, 56, 2, 1, 3), (128, 28, 4, 2, 3), (64, 28, 2, 2, 2), (64, 8, 2, 2, 5), (128, 16, 4, 4, 2), (64, 16, 2, 1, 1), (128, 8, 4, 8, 2), (64, 7, 2, 7, 3),
]


@pytest.mark.parametrize("prec,tol", [(1, 1e-3), (2, 8e-3)])
@pytest.mark.parametrize("C,reso,heads,split,B", STRIPES)
def test_fused_stripe_attention_vs_oracle(C, reso, heads, split, B, prec, tol):
    """LayerNorm -> qkv -> two stripe branches of LePE attention in one kernel, against the oracle's composition (fp32)."""
    from mi355attn import functional as F
    from mi355attn.modules import CSWinBlock
    if reso == split:
        pytest.skip("reso == split is the single-branch last stage (cswin.py:146-147), not a stripe pair")
    torch.manual_seed(1234)
    m = CSWinBlock(C, reso, heads, split_size=split, qkv_bias=True).eval()
    with torch.no_grad():                                   # non-trivial LayerNorm affine part: it is folded into the projection
        m.norm1.weight.add_(0.2 * torch.randn(C))
        m.norm1.bias.add_(0.2 * torch.randn(C))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.manual_seed(4321)
    x = torch.randn(B, reso * reso, C)
    u = O.layernorm(x, sd["norm1.weight"], sd["norm1.bias"])
    qkv = O.linear(u, sd["qkv.weight"], sd["qkv.bias"]).reshape(B, reso * reso, 3, C).permute(2, 0, 1, 3)
    half = C // 2
    ref = torch.cat([O.lepe_attention_forward(qkv[..., :half], sd["attns.0.get_v.weight"], sd["attns.0.get_v.bias"], reso, 0, split, heads // 2),
                     O.lepe_attention_forward(qkv[..., half:], sd["attns.1.get_v.weight"], sd["attns.1.get_v.bias"], reso, 1, split, heads // 2)], dim=2)
    m = m.cuda()
    assert F.cswin_stripe_ok(C, reso, split, heads, prec)
    with torch.no_grad():
        ctx = F.cswin_stripe_attention(x.cuda(), m.norm1, m.qkv, m.attns[0].get_v, m.attns[1].get_v, reso, heads, split, m.attns[0].scale, prec)
        ctx2 = F.cswin_stripe_attention(x.cuda(), m.norm1, m.qkv, m.attns[0].get_v, m.attns[1].get_v, reso, heads, split, m.attns[0].scale, prec)
    assert torch.equal(ctx, ctx2)
    assert_parity(ctx.float().cpu(), ref, tol, f"stripe attention C={C} reso={reso} split={split}")


@pytest.mark.parametrize("prec,tol", [(1, 3e-4), (2, 3e-3)])
@pytest.mark.parametrize("C", [64, 128])
@pytest.mark.parametrize("M", [1, 33, 802, 70000])
def test_fused_proj_mlp_vs_fp64(M, C, prec, tol):
    """x1 = x + proj(ctx16); y = x1 + fc2(gelu(fc1(LN(x1)))) in one kernel against fp64 on the same 16-bit ctx."""
    from mi355attn import functional as F
    from torch import nn
    torch.manual_seed(M + C)
    ln, proj, fc1, fc2 = nn.LayerNorm(C), nn.Linear(C, C), nn.Linear(C, 4 * C), nn.Linear(4 * C, C)
    with torch.no_grad():
        ln.weight.add_(0.2 * torch.randn(C)); ln.bias.add_(0.2 * torch.randn(C))
    mods = [mm.eval().cuda() for mm in (ln, proj, fc1, fc2)]
    ln, proj, fc1, fc2 = mods
    x = torch.randn(M, C, device="cuda")
    ctx16 = torch.randn(M, C, device="cuda").to(F.dtype16(prec))
    with torch.no_grad():
        y = F.mlp_fused(x, ln, fc1, fc2, precision=prec, ctx16=ctx16, proj=proj)
        y2 = F.mlp_fused(x, ln, fc1, fc2, precision=prec, ctx16=ctx16, proj=proj)
        d = lambda t: t.detach().double().cpu()
        x1 = d(x) + d(ctx16) @ d(proj.weight).t() + d(proj.bias)
        u = torch.nn.functional.layer_norm(x1, (C,), d(ln.weight), d(ln.bias), ln.eps)
        ref = x1 + torch.nn.functional.gelu(u @ d(fc1.weight).t() + d(fc1.bias)) @ d(fc2.weight).t() + d(fc2.bias)
    assert torch.equal(y, y2)
    assert_parity(y.cpu(), ref.float(), tol, f"proj + MLP fused M={M} C={C}")


def test_split_round_poll_timeout_is_reported():
    """The reducer chunk of the split last round polls a counter the partner chunks increment; with a zero poll budget it gives up at
    once, the launch still terminates, and the failure surfaces through mi355_sync_status / the next split launch (MI355_ESYNC)."""
    import mi355attn
    from mi355attn import functional as F
    torch.manual_seed(3)
    M, N, K = 50432, 768, 3072
    x16 = torch.randn(M, K, device="cuda").half()
    w16 = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    mi355attn.sync_status(wait=True)
    old = mi355attn.get_option("spin_limit")
    try:
        mi355attn.set_option("spin_limit", 0)
        F.linear16(x16, w16, precision=1)
        torch.cuda.synchronize()
        mi355attn.set_option("spin_limit", old)
        try:
            mi355attn.sync_status()
            timed_out = False
        except mi355attn.Mi355Error as e:
            assert "split last round" in str(e)
            timed_out = True
    finally:
        mi355attn.set_option("spin_limit", old)
    # a partner can legitimately have published before the very first poll: then there is nothing to report and the result is right
    y = F.linear16(x16, w16, precision=1)
    mi355attn.sync_status(wait=True)
    ref = x16[:64].double().cpu() @ w16.double().cpu().t()
    assert_parity(y[:64].cpu(), ref.float(), 2e-6, "after the zero-budget launch (timed out: %s)" % timed_out)


DA_SHAPES = [  # (B, C, H, W): c_m = c_n = 128 -- the two-pass path; pixel counts with and without a ragged last 32-pixel tile
    (3, 256, 56, 56), (1, 256, 56, 56), (2, 256, 14, 14), (5, 128, 28, 28), (2, 256, 10, 10), (1, 128, 2, 2), (2, 256, 6, 10), (37, 256, 8, 8), (2, 256, 9, 12), (300, 128, 6, 6),
]


@pytest.mark.parametrize("prec,tol", [(1, 1e-3), (2, 8e-3)])
@pytest.mark.parametrize("B,C,H,W", DA_SHAPES)
def test_double_attention_two_pass_vs_oracle(B, C, H, W, prec, tol):
    """DoubleAttention(C, 128, 128) through the two-pass kernels against the oracle, and against the seven-launch pipeline of the same
    library (option da_fused = 0); run-to-run identical; an image's result does not depend on its batch neighbours' values."""
    import mi355attn
    from mi355attn import functional as F
    from mi355attn.modules import DoubleAttention
    torch.manual_seed(100 + B + C + H)
    m = DoubleAttention(C, 128, 128).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    keys = ("convA.weight", "convA.bias", "convB.weight", "convB.bias", "convV.weight", "convV.bias", "proj.weight", "proj.bias")
    x = torch.randn(B, C, H, W) * 1.5
    ref = O.double_attention_forward(x, *[sd[k] for k in keys])
    args = [x.cuda()] + [sd[k].cuda() for k in keys]
    y = F.double_attention_forward(*args, precision=prec)
    y2 = F.double_attention_forward(*args, precision=prec)
    assert torch.equal(y, y2), "run-to-run"
    assert_parity(y.cpu(), ref, tol, f"two-pass DoubleAttention B={B} C={C} {H}x{W}")
    try:
        mi355attn.set_option("da_fused", 0)
        yu = F.double_attention_forward(*args, precision=prec)
    finally:
        mi355attn.set_option("da_fused", 1)
    tol_u = tol if prec == 1 else 4e-2                          # bf16 through seven roundings of fp32 intermediates: looser on max |diff|
    assert_parity(yu.cpu(), ref, tol_u, "seven-launch pipeline")
    assert_parity(y.cpu(), yu.cpu(), tol_u, "two-pass vs seven-launch")
    if B > 1:
        xs = args[0].clone()
        xs[1:] = torch.randn_like(xs[1:]) * 3
        y3 = F.double_attention_forward(xs, *args[1:], precision=prec)
        # the pixel ranges per image depend on B only, not on the data: image 0 is bit-identical
        assert torch.equal(y3[0], y[0]), "image 0 changed with its neighbours' values"


@pytest.mark.parametrize("prec,tol", [(1, 1e-3), (2, 8e-3)])
@pytest.mark.parametrize("B,HW,ps,E", [(256, 224, 16, 768), (3, 224, 16, 768), (300, 128, 8, 512), (5, 224, 16, 264)])
def test_patch_embed_on_the_16bit_engine(B, HW, ps, E, prec, tol):
    """mi355_patch_embed_ws_fwd: im2col into the operand format + persistent GEMM with the position rows as a periodic residual table,
    against fp64 on sampled images and against the implicit-GEMM kernel of mi355_patch_embed_fwd (same operand roundings, fp32
    summation order differs); the cls row (LAST row of every image) must be cls + pos[P] exactly up to one fp32 addition."""
    import math
    from mi355attn import functional as F
    from mi355attn._ffi import lib
    torch.manual_seed(E + B)
    Cin = 3
    img = torch.randn(B, Cin, HW, HW, device="cuda")
    w, b = (torch.randn(E, Cin, ps, ps) / math.sqrt(Cin * ps * ps)).cuda(), torch.randn(E).cuda()
    P = (HW // ps) ** 2
    cls, pos = torch.randn(E).cuda(), torch.randn(P + 1, E).cuda()
    assert lib().mi355_patch_embed_workspace_bytes(B, Cin, HW, HW, ps, E, prec) > 0, "shape chosen to take the 16-bit engine"
    tok = F.patch_embed(img, w, b, cls, pos, ps, precision=prec)
    tok2 = F.patch_embed(img, w, b, cls, pos, ps, precision=prec)
    assert torch.equal(tok, tok2)
    assert tok.shape == (B, P + 1, E)
    # the implicit-GEMM kernel through the entry point without a workspace
    old = torch.empty_like(tok)
    from mi355attn._ffi import check, dptr, stream_ptr
    check(lib().mi355_patch_embed_fwd(dptr(img), dptr(w.reshape(E, -1)), dptr(b), dptr(cls), dptr(pos), dptr(old), B, Cin, HW, HW, ps, E,
                                      prec, stream_ptr(img.device)), "mi355_patch_embed_fwd")
    assert_parity(tok.cpu(), old.cpu(), 2e-6 if prec == 1 else 2e-5, "16-bit engine vs implicit GEMM")
    pick = [0, B // 2, B - 1]
    ref = O.vit_patch_embed_forward(img[pick].cpu(), w.cpu(), b.cpu(), torch.float64)
    ref = torch.cat([ref, cls.cpu().double().expand(len(pick), 1, E)], dim=1) + pos.cpu().double()
    assert_parity(tok[pick].cpu(), ref.float(), tol, "patch_embed on the 16-bit engine")
    assert_parity(tok[:, P].cpu(), (cls + pos[P]).cpu().expand(B, E), 1e-6, "cls rows")


def test_double_attention_two_pass_nan_poisons_one_image_only():
    """One NaN pixel in image 1: softmax over H*W of every B row of that image is NaN in the reference, hence G and the whole output
    of image 1; the other images must not notice (bit-identical to the clean run)."""
    from mi355attn import functional as F
    from mi355attn.modules import DoubleAttention
    torch.manual_seed(77)
    m = DoubleAttention(256, 128, 128).eval()
    sd = {k: v.detach().clone().cuda() for k, v in m.state_dict().items()}
    keys = ("convA.weight", "convA.bias", "convB.weight", "convB.bias", "convV.weight", "convV.bias", "proj.weight", "proj.bias")
    x = torch.randn(3, 256, 20, 20, device="cuda")
    clean = F.double_attention_forward(x, *[sd[k] for k in keys], precision=1)
    x[1, 17, 3, 5] = float("nan")
    y = F.double_attention_forward(x, *[sd[k] for k in keys], precision=1)
    ref = O.double_attention_forward(x.cpu(), *[sd[k].cpu() for k in keys])
    assert torch.isnan(ref[1]).all() and torch.isfinite(ref[0]).all()
    assert torch.isnan(y[1]).all(), "image with a NaN pixel must be NaN everywhere, as in the reference"
    assert torch.equal(y[0], clean[0]) and torch.equal(y[2], clean[2])
