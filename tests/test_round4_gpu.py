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
    from conftest import no_range_fallback
    with torch.no_grad(), no_range_fallback():                 # the raw fp16 result (round 6: module(x) itself would fall back to strict)
        y_fast = blk(x.cuda())
    torch.cuda.synchronize()
    try:
        mi355attn.range_status(wait=True)
    except mi355attn.Mi355RangeError:
        pass
    assert not torch.isfinite(y_fast).all(), "the construction no longer saturates the fused kernel: strengthen it"
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        with torch.no_grad(), no_range_fallback():             # guarded_forward is the explicit form of the fallback: tested on its own
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


# ---- left-over rows of the two-accumulator GEMM on ring-pipelined small tiles (option "gemm_pa_tail"; gemm16.hip) -------------------
TAIL_SHAPES = [  # (M, N, K, tail kernel expected)
    (128 * 392, 512, 2048, "32x64"),      # MixerLayer fc2 at B = 256: 784 tiles = 3.06 rounds -> 49 152 rows + 1 024 rows (the default's case)
    (256 * 196, 384, 1536, "32x64"),      # XCiT-S fc2: 588 swapped tiles = 2.30 rounds -> 43 520 rows + 6 656 rows (1 248 ring workgroups, five per CU in turn)
    (128 * 264, 256, 1024, "32x64"),      # one column tile, eight left-over tiles, the shortest reduction that splits
    (128 * 160, 512, 1088, "32x64"),      # 320 tiles: one round + 64 (4 096 rows); 17 K-tiles (more than the ring holds)
]


@pytest.mark.parametrize("prec,dt", [(1, torch.float16), (2, torch.bfloat16)])
@pytest.mark.parametrize("M,N,K,which", TAIL_SHAPES)
def test_left_over_rows_on_the_ring_kernel_are_bit_identical(M, N, K, which, prec, dt):
    import mi355attn
    from mi355attn import functional as F
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("round arithmetic of the shapes above assumes 256 CUs")
    torch.manual_seed(M + N + K)
    x16 = torch.randn(M, K, device="cuda").to(dt)
    w16 = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
    b = torch.randn(N, device="cuda")
    resid = torch.randn(M, N, device="cuda")
    old = mi355attn.get_option("gemm_pa_tail")
    try:
        for kw in (dict(bias=b, resid=resid), dict(bias=None, resid=resid), dict(bias=b, resid=resid, act=F.ACT_GELU)):
            mi355attn.set_option("gemm_pa_tail", 0)
            y_off = F.linear16(x16, w16, precision=prec, **kw)
            mi355attn.set_option("gemm_pa_tail", 30)
            tags = []
            y_on = [None]

            def run():
                y_on[0] = F.linear16(x16, w16, precision=prec, **kw)
            tags = [t for t, *_ in mi355attn.kernel_trace(run)]
            assert any(f"tail {which}" in t for t in tags), f"the ring kernel did not run: {tags}"
            y_on2 = F.linear16(x16, w16, precision=prec, **kw)
            torch.cuda.synchronize()
            assert torch.equal(y_on[0], y_on2), "run-to-run difference"
            assert torch.equal(y_off, y_on[0]), f"split launch differs from the single launch with {sorted(kw)}"
        mi355attn.set_option("gemm_variant", 7)
        y7 = F.linear16(x16, w16, b, resid=resid, precision=prec)
        mi355attn.set_option("gemm_variant", 0)
        assert torch.equal(y7, F.linear16(x16, w16, b, resid=resid, precision=prec)), "differs from the round-1 tile kernel"
        # the last rows (the ring kernel's) against an fp64 product of the same 16-bit operands
        rows = slice(M - 96, M)
        ref = (x16[rows].double() @ w16.double().t() + b.double() + resid[rows].double()).float()
        assert_parity(F.linear16(x16, w16, b, resid=resid, precision=prec)[rows], ref, 2e-5, "fp64 product, last rows")
    finally:
        mi355attn.set_option("gemm_variant", 0)
        mi355attn.set_option("gemm_pa_tail", old)


def test_left_over_rows_split_is_recorded_under_graph_capture():
    """Two plain launches on one stream: nothing to step aside for under capture."""
    import mi355attn
    from mi355attn import functional as F
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("assumes 256 CUs")
    torch.manual_seed(1)
    M, N, K = 128 * 392, 512, 2048
    x16 = torch.randn(M, K, device="cuda").half()
    w16 = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    resid = torch.randn(M, N, device="cuda")
    y_eager = F.linear16(x16, w16, None, resid=resid, precision=1)
    y_graph = torch.empty_like(y_eager)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        F.linear16(x16, w16, None, resid=resid, precision=1)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        y_graph.copy_(F.linear16(x16, w16, None, resid=resid, precision=1))
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(y_eager, y_graph)


# ---- LPI on 2 x 2 patches per lane (xcit.hip lpi_patch_kernel; option "lpi_patch") ---------------------------------------------------
@pytest.mark.parametrize("C,with_ln,with_tail", [(384, True, True), (96, False, True), (32, True, False), (64, False, False)])
def test_lpi_patch_kernel_against_the_general_kernel_and_the_oracle(C, with_ln, with_tail):
    """14 x 14 tokens, C % 32 == 0: the patch kernel (fused multiply-adds, taps in scalar registers) against the general kernel
    (option off: separately rounded products) to 1e-6, against the oracle, run-to-run bit identity, and a row's independence of the
    batch around it."""
    import mi355attn
    from mi355attn.modules import LPI
    torch.manual_seed(C + 7)
    B, H, W = 5, 14, 14
    m = LPI(C).eval()
    with torch.no_grad():
        m.bn.running_mean.normal_(0, 0.2)
        m.bn.running_var.uniform_(0.5, 1.5)
        m.bn.weight.uniform_(0.5, 1.5)
        m.bn.bias.normal_(0, 0.2)
    ln = torch.nn.LayerNorm(C)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.normal_(0, 0.2)
    x = torch.randn(B, H * W, C) * 1.5 + 0.3
    gamma = torch.rand(C) + 0.5
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m, ln = m.cuda(), ln.cuda()
    xd, gd = x.cuda(), gamma.cuda()
    kw = dict(gamma=gd, resid=xd) if with_tail else {}
    if with_ln:
        kw["ln"] = ln
    old = mi355attn.get_option("lpi_patch")
    try:
        with torch.no_grad():
            mi355attn.set_option("lpi_patch", 1)
            tags = []
            out = [None]

            def run():
                out[0] = m(xd, H, W, **kw)
            tags = [t for t, *_ in mi355attn.kernel_trace(run)]
            assert any("lpi_patch_kernel" in t for t in tags), tags
            y1 = out[0]
            y1b = m(xd, H, W, **kw)
            y_one = m(xd[3:4].contiguous(), H, W, **({**kw, "resid": xd[3:4].contiguous()} if with_tail else kw))
            mi355attn.set_option("lpi_patch", 0)
            y0 = m(xd, H, W, **kw)
    finally:
        mi355attn.set_option("lpi_patch", old)
    torch.cuda.synchronize()
    assert torch.equal(y1, y1b), "run-to-run difference"
    assert torch.equal(y1[3:4], y_one), "a row's bits depend on the batch"
    assert_parity(y1.cpu(), y0.cpu(), 1e-6, "patch kernel vs general kernel")
    u = torch.nn.functional.layer_norm(x, (C,), ln.weight.cpu(), ln.bias.cpu(), ln.eps) if with_ln else x
    ref = O.lpi_forward(u, sd, H, W)
    if with_tail:
        ref = x + gamma * ref
    assert_parity(y1.cpu(), ref, 2e-5, "patch kernel vs oracle")


# ---- MixerLayer token mixing in one kernel (csrc/mixer_fused.hip; option "mixer_fused") -----------------------------------------------
@pytest.mark.parametrize("prec,tol", [(1, 1e-3), (2, 8e-3)])
@pytest.mark.parametrize("B,C", [(1, 256), (5, 512), (3, 768)])
def test_mixer_token_mixing_in_one_kernel(B, C, prec, tol):
    """N = 196 tokens, T = C / 2 hidden token units (128 / 256 / 384): the fused half (row statistics + one kernel) against the three-launch path it replaces,
    against the oracle of the whole layer, run-to-run bit identity, and an image's independence of its batch."""
    import mi355attn
    from mi355attn.modules import MixerLayer
    torch.manual_seed(B * 1000 + C)
    m = MixerLayer(C, 196, precision=prec).eval()
    with torch.no_grad():
        for ln in (m.norm1, m.norm2):
            ln.weight.uniform_(0.5, 1.5)
            ln.bias.normal_(0, 0.2)
        m.token_mlp.fc1.bias.normal_(0, 0.3)
        m.token_mlp.fc2.bias.normal_(0, 0.3)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(B, 196, C) * 1.3 + 0.2
    ref = O.mixer_layer_forward(x, sd)
    m = m.cuda()
    xd = x.cuda()
    old = mi355attn.get_option("mixer_fused")
    try:
        with torch.no_grad():
            mi355attn.set_option("mixer_fused", 1)
            out = [None]

            def run():
                out[0] = m(xd)
            tags = [t for t, *_ in mi355attn.kernel_trace(run)]
            assert any("mixer_token_kernel" in t for t in tags), tags
            assert not any("layernorm16_t" in t for t in tags), tags
            y1 = out[0]
            y1b = m(xd)
            y_last = m(xd[B - 1:].contiguous())
            mi355attn.set_option("mixer_fused", 0)
            y0 = m(xd)
    finally:
        mi355attn.set_option("mixer_fused", old)
    torch.cuda.synchronize()
    assert torch.isfinite(y1).all()
    assert torch.equal(y1, y1b), "run-to-run difference"
    assert torch.equal(y1[B - 1:], y_last), "an image's bits depend on the batch"
    assert_parity(y1.cpu(), ref, tol, "fused token mixing vs oracle")
    assert_parity(y0.cpu(), ref, tol, "three-launch token mixing vs oracle")
    assert_parity(y1.cpu(), y0.cpu(), tol, "fused vs three launches")


def test_mixer_token_entry_refuses_other_geometries():
    import ctypes
    import mi355attn
    from mi355attn import _ffi
    L = _ffi.lib()
    x = torch.zeros(1, 49, 256, device="cuda")
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    p = lambda t_: ctypes.c_void_p(t_.data_ptr())
    rc = L.mi355_mixer_token_fwd(p(x), p(buf), p(buf), 1e-5, p(buf), p(buf), p(buf), p(buf), p(x), 1, 49, 256, 256, 1, p(buf), 1 << 16,
                                 _ffi.stream_ptr(None))
    assert rc == -2, rc          # MI355_EUNSUPPORTED (include/mi355attn.h)
