"""Round-6 kernels against the paths they replace (bit identity) and against an fp64 product ("bit-identity and
oracle tests as for gemm16_w4").

  * gemm16_wreg.hip: fp32 (+ residual) outputs of square short products (N = K = 256 / 384: XCiT proj xcit.py:263, CSWin stage-3 proj
    cswin.py:192) with the weights stationary in registers.
"""
import pytest
import torch

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _tags(fn):
    import mi355attn
    return [t for t, *_ in mi355attn.kernel_trace(fn)]


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("M,K,resid,bias", [(50176, 384, True, True), (50176, 256, True, True), (4096 + 37, 384, True, False),
                                            (8192 + 16, 256, False, True), (12544, 384, False, False), (392, 384, True, True), (40, 256, True, False)])
def test_weight_stationary_gemm_is_bit_identical_to_the_tile_kernels(M, K, resid, bias, prec):
    import mi355attn
    from mi355attn import functional as F
    torch.manual_seed(M + K + prec)
    x = torch.randn(M, K, device="cuda")
    w = (torch.randn(K, K, device="cuda") / K ** 0.5).contiguous()
    b = torch.randn(K, device="cuda") if bias else None
    r = torch.randn(M, K, device="cuda") if resid else None
    x16, w16 = F.cast16(x, prec), F.cast16(w, prec)
    outs, tags = {}, {}
    old = mi355attn.get_option("gemm_wreg")
    try:
        for v in (1, 0):
            mi355attn.set_option("gemm_wreg", v)

            def run():
                outs[v] = F.linear16(x16, w16, b, resid=r, precision=prec)
            tags[v] = _tags(run)
            torch.cuda.synchronize()
    finally:
        mi355attn.set_option("gemm_wreg", old)
    assert any("gemm16_wreg_kernel" in t for t in tags[1]), tags[1]
    assert not any("gemm16_wreg_kernel" in t for t in tags[0]), tags[0]
    assert torch.equal(outs[1], outs[0]), "the weight-stationary kernel and the tile kernel differ in some bit"
    ref = x16.double().cpu() @ w16.double().cpu().t()
    if bias:
        ref = ref + b.double().cpu()
    if resid:
        ref = ref + r.double().cpu()
    assert_parity(outs[1].cpu(), ref.float(), 2e-6, "gemm16_wreg vs fp64 product of the 16-bit operands")
    with torch.no_grad():                                              # run-to-run
        again = F.linear16(x16, w16, b, resid=r, precision=prec)
    assert torch.equal(again, outs[1])


def test_weight_stationary_gemm_output_may_alias_the_residual():
    """The block mirrors write x = x + proj(ctx) into a fresh tensor, but a C caller may pass y == resid: every lane reads its own 16 bytes
    of the residual before it writes them."""
    from mi355attn import functional as F
    from mi355attn import _ffi
    import mi355attn
    torch.manual_seed(7)
    M, K = 8192, 384
    x16 = F.cast16(torch.randn(M, K, device="cuda"), 1)
    w16 = F.cast16((torch.randn(K, K, device="cuda") / K ** 0.5).contiguous(), 1)
    r = torch.randn(M, K, device="cuda")
    want = F.linear16(x16, w16, None, resid=r, precision=1)
    y = r.clone()
    L = mi355attn.lib()
    assert L.mi355_linear16_fwd(_ffi.dptr(x16), _ffi.dptr(w16), None, None, _ffi.dptr(y), _ffi.dptr(y), M, K, K, K, K, 0, 0, 1,
                                _ffi.stream_ptr(y.device)) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, want)


def _xca_ref(qkv16, temperature, heads):
    """fp64 evaluation of xcit.py:249-262 on the 16-bit inputs (what both kernels are given)."""
    B, N, C3 = qkv16.shape
    C = C3 // 3
    d = C // heads
    q, k, v = (qkv16.double().cpu().reshape(B, N, 3, heads, d).permute(2, 0, 3, 4, 1))      # (3, B, h, d, N)
    q = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    k = k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    a = ((q @ k.transpose(-2, -1)) * temperature.double().cpu().reshape(1, heads, 1, 1)).softmax(dim=-1)
    return (a @ v).permute(0, 3, 1, 2).reshape(B, N, C)


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("B,N,heads,d", [(5, 196, 8, 48), (3, 197, 4, 32), (2, 224, 2, 64), (4, 50, 8, 48), (2, 16, 4, 32)])
def test_xca_core_on_the_16bit_pipe_matches_the_streaming_kernel(B, N, heads, d, prec):
    """xca_tr_kernel (round 6: covariance on the 16-bit matrix pipe from one transposed LDS image) against xca_kernel (exact-fp32 MFMAs on
    fp32 copies) on the same 16-bit q / k / v: both round the context to the operand type, so they agree to one unit of that rounding;
    both against the fp64 evaluation."""
    import mi355attn
    from mi355attn import functional as F
    torch.manual_seed(B * 1000 + N + d)
    C = heads * d
    qkv16 = F.cast16(torch.randn(B, N, 3 * C, device="cuda"), prec)
    temp = (0.5 + torch.rand(heads, device="cuda")) * 3.0
    outs, tags = {}, {}
    old = mi355attn.get_option("xca_tr")
    try:
        for v in (1, 0):
            mi355attn.set_option("xca_tr", v)

            def run():
                outs[v] = F.xca_core(qkv16, temp, heads, precision=prec, out16=True)
            tags[v] = _tags(run)
            torch.cuda.synchronize()
    finally:
        mi355attn.set_option("xca_tr", old)
    assert any("xca_tr_kernel" in t for t in tags[1]) and not any("xca_tr_kernel" in t for t in tags[0]), (tags[1], tags[0])
    ref = _xca_ref(qkv16, temp, heads).float()
    tol = 1.5e-3 if prec == 1 else 1.2e-2                               # the output itself is rounded to fp16 / bf16
    assert_parity(outs[1].float().cpu(), ref, tol, "xca_tr vs fp64")
    assert_parity(outs[0].float().cpu(), ref, tol, "xca (streaming) vs fp64")
    # one against the other: the same A within ~1e-7, then the same 16-bit P.V -> at most an ulp of the operand type apart
    ulp = 2.0 ** -10 if prec == 1 else 2.0 ** -7
    diff = (outs[1].float() - outs[0].float()).abs().max().item()
    assert diff <= 2 * ulp * outs[0].float().abs().max().item(), diff
    again = F.xca_core(qkv16, temp, heads, precision=prec, out16=True)
    assert torch.equal(again, outs[1]), "run-to-run"


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("C,M,use_gamma", [(256, 50176, False), (384, 50176, True), (384, 1000, False), (256, 97, True), (384, 196 * 3 + 5, True)])
def test_wide_fused_mlp_matches_the_reference_expression(C, M, use_gamma, prec):
    """mlp_wide.hip (LayerNorm + fc1 + GELU + fc2 + LayerScale + residual in one kernel at C = 256 / 384) against an fp64 evaluation of
    x + gamma * fc2(gelu(fc1(LN(x)))) (cswin.py:194-196 / xcit.py:294 with Mlp), on a ragged row count (steps that are not full, rows
    beyond M), and against the composition it replaces (LayerNorm launch + two GEMMs, option "mlp_wide" = 0)."""
    import torch.nn as nn
    import mi355attn
    from mi355attn import functional as F
    torch.manual_seed(C + M)
    ln = nn.LayerNorm(C)
    fc1, fc2 = nn.Linear(C, 4 * C), nn.Linear(4 * C, C)
    with torch.no_grad():
        ln.weight.add_(0.2 * torch.randn(C))
        ln.bias.add_(0.1 * torch.randn(C))
    gamma = (0.5 + torch.rand(C)) if use_gamma else None
    x = torch.randn(M, C) * 1.5 + 0.3
    xd = x.double()
    u = torch.nn.functional.layer_norm(xd, (C,), ln.weight.double(), ln.bias.double(), ln.eps)
    h = torch.nn.functional.gelu(u @ fc1.weight.double().t() + fc1.bias.double())
    br = h @ fc2.weight.double().t() + fc2.bias.double()
    ref = (xd + (br * gamma.double() if use_gamma else br)).float()
    ln, fc1, fc2 = ln.cuda(), fc1.cuda(), fc2.cuda()
    gd = gamma.cuda() if use_gamma else None
    xg = x.cuda()
    tol = 1e-3 if prec == 1 else 8e-3
    assert not F.mlp_fused_ok(C, 4 * C, prec), "opt-in: measured slower than the composition (profiles/r06_mlp_wide.md)"
    mi355attn.set_option("mlp_wide", 1)
    try:
        assert F.mlp_fused_ok(C, 4 * C, prec)
        got = {}

        def run():
            got["y"] = F.mlp_fused(xg.view(1, M, C), ln, fc1, fc2, gamma=gd, precision=prec)
        tags = _tags(run)
        assert any("mlp_wide_kernel" in t for t in tags), tags
        y = got["y"].view(M, C)
        again = F.mlp_fused(xg.view(1, M, C), ln, fc1, fc2, gamma=gd, precision=prec).view(M, C)
        sub = F.mlp_fused(xg[:200].contiguous().view(1, 200, C), ln, fc1, fc2, gamma=gd, precision=prec).view(200, C) if M > 300 else None
    finally:
        mi355attn.set_option("mlp_wide", 0)
    assert_parity(y.cpu(), ref, tol, "mlp_wide vs fp64")
    # the BRANCH alone (y - x): the residual must not hide an error of the MLP
    assert_parity((y - xg).cpu(), (ref.double() - xd).float(), 2 * tol, "mlp_wide branch vs fp64")
    assert torch.equal(again, y), "run-to-run"
    # batch independence: the first rows alone give the same bits (a row's arithmetic does not depend on its step)
    if sub is not None:
        assert torch.equal(sub, y[:200])
    # the composition it replaces
    u16 = F.layernorm16(xg, ln.weight, ln.bias, ln.eps, prec)
    h16 = F.linear16(u16, F.weight16(fc1.weight, prec), fc1.bias, act=F.ACT_GELU, out16=True, precision=prec)
    comp = F.linear16(h16, F.weight16(fc2.weight, prec), fc2.bias, gamma=gd, resid=xg, precision=prec)
    assert_parity(y.cpu(), comp.cpu(), tol, "mlp_wide vs LayerNorm + two GEMMs")


def test_wide_fused_mlp_reports_a_saturating_hidden_activation():
    """fc1 scaled so that gelu(H) passes 65504: the static proof fails on the host, the kernel tracks and reports (code 4), and module(x)
    of a CSWin stage-3 block falls back to strict mode like the reference (cswin.py:194-196 computes in fp32)."""
    import warnings
    import oracle as O
    import mi355attn
    from mi355attn.modules import CSWinBlock
    torch.manual_seed(1234)
    m = CSWinBlock(256, 14, 8, split_size=7, qkv_bias=True).eval()
    with torch.no_grad():
        m.mlp.fc1.weight.mul_(1e5)
        m.mlp.fc2.weight.mul_(1e-4)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.manual_seed(4321)
    x = torch.randn(4, 196, 256)
    ref = O.cswin_block_forward(x, sd, 14, 8, 7)
    md = m.cuda()
    mi355attn.range_status(wait=True)
    mi355attn.set_option("mlp_wide", 1)
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            with torch.no_grad():
                tags = _tags(lambda: md(x.cuda()))
                y = md(x.cuda())
            torch.cuda.synchronize()
    finally:
        mi355attn.set_option("mlp_wide", 0)
    assert any("mlp_wide_kernel" in t for t in tags), tags
    assert len([i for i in w if "strict mode" in str(i.message)]) == 2, [str(i.message) for i in w]      # two forwards, each falls back
    assert_parity(y.cpu(), ref, 2e-4, "CSWin s3 with a saturating hidden activation [strict re-run]")


@pytest.mark.parametrize("M,K", [(50176, 384), (4096 + 21, 256), (588, 384), (33, 256)])
def test_weight_stationary_gemm_writes_the_row_statistics_of_its_output(M, K):
    """mi355_linear16_stats_fwd: Y is bit-identical to the plain kernel and stats = (mean, 1 / sqrt(var + eps)) of every row of Y (two-pass,
    biased variance) -- what XCABlock's norm3 needs in front of LPI (xcit.py:292), without the statistics pass over Y."""
    from mi355attn import functional as F
    torch.manual_seed(M + K)
    x16 = F.cast16(torch.randn(M, K, device="cuda"), 1)
    w16 = F.cast16((torch.randn(K, K, device="cuda") / K ** 0.5).contiguous(), 1)
    b = torch.randn(K, device="cuda")
    r = torch.randn(M, K, device="cuda") * 2.0 + 0.7
    plain = F.linear16(x16, w16, b, resid=r, precision=1)
    got = F.linear16_stats(x16, w16, b, r, 1e-6, 1)
    assert got is not None
    y, st = got
    torch.cuda.synchronize()
    assert torch.equal(y, plain)
    yd = y.double()
    mean = yd.mean(dim=1)
    rstd = 1.0 / torch.sqrt(yd.var(dim=1, unbiased=False) + 1e-6)
    assert float((st[:, 0].double() - mean).abs().max()) <= 2e-6 * float(yd.abs().max())
    assert float(((st[:, 1].double() - rstd) / rstd).abs().max()) <= 5e-6
    y2, st2 = F.linear16_stats(x16, w16, b, r, 1e-6, 1)
    assert torch.equal(st2, st) and torch.equal(y2, y), "run-to-run"
    assert F.linear16_stats(x16[:, :K], F.cast16(torch.randn(128, K, device="cuda"), 1), None, torch.randn(M, 128, device="cuda"), 1e-6, 1) is None


def test_xcablock_uses_the_statistics_of_the_proj_gemm():
    """XCABlock at the XCiT-S width: no ln_stats_kernel launch any more, the result still matches the oracle and the launch without the
    fold (option gemm_wreg = 0) to rounding."""
    import oracle as O
    import mi355attn
    from mi355attn.modules import XCABlock
    torch.manual_seed(1234)
    m = XCABlock(384, 8, qkv_bias=True, eta=1.0).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.manual_seed(4321)
    x = torch.randn(24, 196, 384)
    ref = O.xca_block_forward(x, sd, 8, 14, 14)
    md, xd = m.cuda(), x.cuda()
    out = {}
    with torch.no_grad():
        tags = _tags(lambda: out.__setitem__("y", md(xd, 14, 14)))
        mi355attn.set_option("gemm_wreg", 0)
        try:
            tags0 = _tags(lambda: out.__setitem__("y0", md(xd, 14, 14)))
        finally:
            mi355attn.set_option("gemm_wreg", 1)
    assert any("resid+stats" in t for t in tags) and not any("ln_stats_kernel" in t for t in tags), tags
    assert any("ln_stats_kernel" in t for t in tags0), tags0
    assert_parity(out["y"].cpu(), ref, 1e-3, "XCABlock with the statistics fold")
    assert_parity(out["y"].cpu(), out["y0"].cpu(), 1e-4, "fold vs statistics pass")


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("M", [50176, 588, 40])
def test_weight_stationary_gemm_emits_the_next_layernorm(M, prec):
    """mi355_linear16_ln16_fwd (N = K = 256: CSWin stage 3, cswin.py:192-194): Y bit-identical to the plain kernel, U16 = LayerNorm(Y) in the
    operand format against an fp64 evaluation and against mi355_layernorm16_fwd on the same Y (one unit of the operand type apart at most)."""
    import torch.nn as nn
    from mi355attn import functional as F
    K = 256
    torch.manual_seed(M + prec)
    x16 = F.cast16(torch.randn(M, K, device="cuda"), prec)
    w16 = F.cast16((torch.randn(K, K, device="cuda") / K ** 0.5).contiguous(), prec)
    b = torch.randn(K, device="cuda")
    r = torch.randn(M, K, device="cuda") * 1.5 - 0.4
    ln = nn.LayerNorm(K).cuda()
    with torch.no_grad():
        ln.weight.add_(0.3 * torch.randn(K, device="cuda"))
        ln.bias.add_(0.2 * torch.randn(K, device="cuda"))
    plain = F.linear16(x16, w16, b, resid=r, precision=prec)
    got = {}
    tags = _tags(lambda: got.__setitem__("r", F.linear16_ln16(x16, w16, b, r, ln, prec)))
    assert any("resid+ln16" in t for t in tags), tags
    y, u = got["r"]
    torch.cuda.synchronize()
    assert torch.equal(y, plain)
    ref = torch.nn.functional.layer_norm(y.double(), (K,), ln.weight.double(), ln.bias.double(), ln.eps).float()
    tol = 1e-3 if prec == 1 else 8e-3
    assert_parity(u.float().cpu(), ref.cpu(), tol, "emitted LayerNorm vs fp64")
    u0 = F.layernorm16(y, ln.weight, ln.bias, ln.eps, prec)
    ulp = 2.0 ** -10 if prec == 1 else 2.0 ** -7
    assert float((u.float() - u0.float()).abs().max()) <= 2 * ulp * float(u0.float().abs().max())
    y2, u2 = F.linear16_ln16(x16, w16, b, r, ln, prec)
    assert torch.equal(u2, u) and torch.equal(y2, y), "run-to-run"
    if M > 300:                                                        # batch independence
        ys, us = F.linear16_ln16(x16[:200].contiguous(), w16, b, r[:200].contiguous(), ln, prec)
        assert torch.equal(us, u[:200]) and torch.equal(ys, y[:200])


def test_cswin_stage3_block_has_no_second_layernorm_launch():
    import oracle as O
    from mi355attn.modules import CSWinBlock
    torch.manual_seed(1234)
    m = CSWinBlock(256, 14, 8, split_size=7, qkv_bias=True).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.manual_seed(4321)
    x = torch.randn(8, 196, 256)
    ref = O.cswin_block_forward(x, sd, 14, 8, 7)
    md, xd = m.cuda(), x.cuda()
    out = {}
    with torch.no_grad():
        tags = _tags(lambda: out.__setitem__("y", md(xd)))
    assert sum("layernorm" in t for t in tags) == 1 and any("resid+ln16" in t for t in tags), tags
    assert_parity(out["y"].cpu(), ref, 1e-3, "CSWin s3 with norm2 emitted by the proj GEMM")


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("M,N,act", [(50432, 2304, 0), (8192 + 21, 3072, 1), (32768, 768, 0)])
def test_weight_stationary_k768_gemm_is_correct_when_opted_in(M, N, act, prec):
    """gemm16_wst.hip (opt-in, `gemm_wst`; measured slower: profiles/r06_gemm_wst.md): against an fp64 product of sampled rows, against the tile
    kernels (one unit of the 16-bit output at most for options 1 / 2, whose K halves are two chains; bit for bit for options 3 / 4), run to run,
    on half the rows, and for a ragged row count (rows beyond M clamped on the way in, never stored)."""
    import mi355attn
    from mi355attn import functional as F
    K = 768
    torch.manual_seed(M + N + prec)
    x16 = F.cast16(torch.randn(M, K, device="cuda"), prec)
    w16 = F.cast16((torch.randn(N, K, device="cuda") / K ** 0.5).contiguous(), prec)
    b = torch.randn(N, device="cuda")
    a = F.ACT_GELU if act else F.ACT_NONE
    outs, tags = {}, {}
    one_wave = N % 256 == 0 and M % 32 == 0                      # what options 3 / 4 take
    sub_m = (M // 2) // 32 * 32                                  # still above the launcher's "eight row tiles per workgroup" floor
    try:
        for v in (2, 4, 0) if one_wave else (2, 0):
            mi355attn.set_option("gemm_wst", v)
            tags[v] = _tags(lambda: outs.__setitem__(v, F.linear16(x16, w16, b, act=a, out16=True, precision=prec)))
        mi355attn.set_option("gemm_wst", 2)
        again = F.linear16(x16, w16, b, act=a, out16=True, precision=prec)
        sub_tags = _tags(lambda: outs.__setitem__("sub", F.linear16(x16[:sub_m].contiguous(), w16, b, act=a, out16=True, precision=prec)))
    finally:
        mi355attn.set_option("gemm_wst", 0)
    torch.cuda.synchronize()
    assert any("gemm16_wst_kernel" in t for t in tags[2]) and not any("gemm16_wst_kernel" in t for t in tags[0]), (tags[2], tags[0])
    rows = torch.tensor([0, 1, 31, 32, 33, M // 2, M - 33, M - 32, M - 2, M - 1], device="cuda")
    ref = x16[rows].double() @ w16.double().t() + b.double()
    if act:
        ref = torch.nn.functional.gelu(ref)
    tol = 1e-3 if prec == 1 else 8e-3
    assert_parity(outs[2][rows].float().cpu(), ref.float().cpu(), tol, "gemm16_wst vs fp64")
    ulp = 2.0 ** -10 if prec == 1 else 2.0 ** -7
    scale = float(outs[0].float().abs().max())
    assert float((outs[2].float() - outs[0].float()).abs().max()) <= 2 * ulp * scale
    assert torch.equal(again, outs[2]), "run-to-run"
    if any("gemm16_wst_kernel" in t for t in sub_tags):           # the same schedule on fewer rows: a row's bits do not depend on its neighbours
        assert torch.equal(outs["sub"], outs[2][:sub_m])
    else:
        assert float((outs["sub"].float() - outs[2][:sub_m].float()).abs().max()) <= 2 * ulp * scale
    if one_wave:                                                  # one chain per row, like the tile kernels: the same bits
        assert any("gemm16_wst_kernel" in t for t in tags[4]), tags[4]
        assert torch.equal(outs[4], outs[0]), "the one-wave-per-SIMD kernel adds a row's K steps in the tile kernels' order"


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("M,N,K,act", [(50176, 1536, 384, 1), (50176, 1152, 384, 0), (50176 - 45, 1152, 384, 0), (50176, 1024, 256, 1),
                                       (12544, 2048, 512, 1), (12544 + 7, 1536, 512, 0), (50176, 768, 256, 0), (1000, 768, 256, 1)])
def test_slab_stationary_short_k_gemm_has_the_tile_kernels_bits(M, N, K, act, prec):
    """gemm16_wslab.hip (round 6, default for GELU epilogues and row counts off the 256-row grid): the same bits as the tile kernels for every row,
    ragged row counts included; an fp64 product of sampled rows; run to run; the default policy takes exactly the products it measured faster on;
    too few rows fall back to the tile kernels (nothing unsupported is launched)."""
    import mi355attn
    from mi355attn import functional as F
    torch.manual_seed(M + N + K + prec)
    x16 = F.cast16(torch.randn(M, K, device="cuda"), prec)
    w16 = F.cast16((torch.randn(N, K, device="cuda") / K ** 0.5).contiguous(), prec)
    b = torch.randn(N, device="cuda")
    a = F.ACT_GELU if act else F.ACT_NONE
    outs, tags = {}, {}
    old = mi355attn.get_option("gemm_wslab")
    try:
        for v in (2, 1, 0):
            mi355attn.set_option("gemm_wslab", v)
            tags[v] = _tags(lambda: outs.__setitem__(v, F.linear16(x16, w16, b, act=a, out16=True, precision=prec)))
        mi355attn.set_option("gemm_wslab", 2)
        again = F.linear16(x16, w16, b, act=a, out16=True, precision=prec)
        nob = F.linear16(x16, w16, None, act=a, out16=True, precision=prec)
        mi355attn.set_option("gemm_wslab", 0)
        nob0 = F.linear16(x16, w16, None, act=a, out16=True, precision=prec)
    finally:
        mi355attn.set_option("gemm_wslab", old)
    torch.cuda.synchronize()
    on = {v: any("gemm16_wslab_kernel" in t for t in tags[v]) for v in tags}
    enough_rows = M >= 4 * 32 * (512 if K != 512 else 256) // (N // (192 if K == 384 else 256))
    assert not on[0]
    assert on[2] == enough_rows, (tags[2], enough_rows)
    assert on[1] == (enough_rows and (bool(act) or M % 256 != 0)), tags[1]
    assert torch.equal(outs[2], outs[0]) and torch.equal(outs[1], outs[0]), "one ascending chain per row and the same epilogue: the same bits"
    assert torch.equal(again, outs[2]), "run-to-run"
    assert torch.equal(nob, nob0), "bias == None"
    rows = torch.tensor([0, 1, 15, 16, 31, 32, M // 2, M - 33, M - 32, M - 2, M - 1], device="cuda")
    ref = x16[rows].double() @ w16.double().t() + b.double()
    if act:
        ref = torch.nn.functional.gelu(ref)
    assert_parity(outs[2][rows].float().cpu(), ref.float().cpu(), 1e-3 if prec == 1 else 8e-3, "gemm16_wslab vs fp64")


def test_slab_stationary_gemm_reports_fp16_saturation():
    """The 16-bit epilogue of gemm16_wslab feeds the range word like the tile kernels' (the `range_fallback` contract of the modules rests on it)."""
    import mi355attn
    from mi355attn import functional as F
    M, N, K = 50176, 1536, 384
    x16 = F.cast16(torch.full((M, K), 8.0, device="cuda"), 1)
    w16 = F.cast16(torch.full((N, K), 32.0, device="cuda"), 1)          # 8 * 32 * 384 = 98 304 > 65 504
    mi355attn.range_status(wait=True)
    old = mi355attn.get_option("gemm_wslab")
    try:
        mi355attn.set_option("gemm_wslab", 2)
        tags = _tags(lambda: F.linear16(x16, w16, None, act=F.ACT_GELU, out16=True, precision=1))
        assert any("gemm16_wslab_kernel" in t for t in tags), tags
        with pytest.raises(mi355attn.Mi355RangeError):
            mi355attn.range_status(wait=True)
    finally:
        mi355attn.set_option("gemm_wslab", old)
        try:
            mi355attn.range_status(wait=True)
        except mi355attn.Mi355RangeError:
            pass


def test_slab_stationary_gemm_under_graph_capture_and_on_a_side_stream():
    """gemm16_wslab has no inter-workgroup exchange, no workspace and no host-side state per launch: a captured launch replays on changed
    inputs bit for bit like an eager one, and a launch on a non-default stream gives the same bits."""
    import mi355attn
    from mi355attn import functional as F
    M, N, K = 50176, 1536, 384
    torch.manual_seed(6)
    x16 = torch.randn(M, K, device="cuda").half()
    w16 = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda")
    run = lambda: F.linear16(x16, w16, b, act=F.ACT_GELU, out16=True, precision=1)      # noqa: E731
    tags = _tags(run)
    assert any("gemm16_wslab_kernel" in t for t in tags), tags                # the default policy sends a GELU epilogue at K = 384 here
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            y_cap = run()
    torch.cuda.current_stream().wait_stream(s)
    for rep in range(3):
        x16.copy_(torch.randn(M, K, device="cuda").half())
        g.replay()
        torch.cuda.synchronize()
        y_ref = run()
        torch.cuda.synchronize()
        assert torch.equal(y_cap, y_ref), "replay %d" % rep
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        y_side = run()
    side.synchronize()
    assert torch.equal(y_side, y_ref)
    mi355attn.range_status(wait=True)


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("M,N,K,act", [(50176, 1152, 384, 0), (50176 - 45, 1536, 384, 1), (50176, 768, 256, 0), (12544, 1536, 512, 1),
                                       (50176, 1000, 384, 0), (100, 1152, 384, 0)])
def test_cast_rides_in_the_gemm_staging_with_the_same_bits(M, N, K, act, prec):
    """mi355_linear16_x32_fwd (round 6): act(T(x32) W16^T + b) in ONE launch equals cast16 + linear16 bit for bit; shapes the slab-stationary
    kernel does not take come back as MI355_EUNSUPPORTED with nothing launched and the host helper composes the two calls."""
    import mi355attn
    from mi355attn import functional as F
    torch.manual_seed(M + N + K + prec)
    x = torch.randn(M, K, device="cuda")
    w16 = F.cast16((torch.randn(N, K, device="cuda") / K ** 0.5).contiguous(), prec)
    b = torch.randn(N, device="cuda")
    a = F.ACT_GELU if act else F.ACT_NONE
    out = {}
    tags = _tags(lambda: out.__setitem__("fused", F.cast_linear16(x, w16, b, act=a, precision=prec)))
    two = F.linear16(F.cast16(x, prec), w16, b, act=a, out16=True, precision=prec)
    torch.cuda.synchronize()
    slabw = {256: 256, 384: 384, 512: 256}[K]                    # fp32 rows: eight waves at K = 384 / 512, four at K = 256
    slots = 256 * (2 if K == 256 else 1)
    takes = N % slabw == 0 and M >= 128 and (M + 31) // 32 >= 4 * (slots // (N // slabw))
    assert any(",x32>" in t for t in tags) == takes, (tags, takes)
    assert (len(tags) == 1) == takes, tags
    assert torch.equal(out["fused"], two)


def test_cast_in_the_gemm_staging_reports_input_saturation():
    """The fused cast keeps mi355_cast16_fwd's range report: a finite |x| >= 65520 is an fp16 inf the fp32 reference does not have."""
    import mi355attn
    from mi355attn import functional as F
    x = torch.randn(50176, 384, device="cuda")
    x[12345, 17] = 7.0e4
    w16 = F.cast16(torch.randn(1152, 384, device="cuda") / 20, 1)
    mi355attn.range_status(wait=True)
    try:
        tags = _tags(lambda: F.cast_linear16(x, w16, None, precision=1))
        assert any(",x32>" in t for t in tags), tags
        with pytest.raises(mi355attn.Mi355RangeError):
            mi355attn.range_status(wait=True)
    finally:
        try:
            mi355attn.range_status(wait=True)
        except mi355attn.Mi355RangeError:
            pass


def test_block_entries_use_the_fused_cast_and_keep_their_bits():
    """mi355_mhsa_fwd and XCA.forward on an fp32 input at C = 384: no cast16 launch (the qkv GEMM takes the fp32 rows), the result equals the
    composition cast16 -> linear16 -> core -> linear16 bit for bit and the oracle within 1e-3."""
    import oracle as O
    import mi355attn
    from mi355attn import functional as F
    from mi355attn.modules import Attention
    B, N, C, heads = 64, 196, 384, 6
    torch.manual_seed(1234)
    m = Attention(C, heads, qkv_bias=True).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.manual_seed(4321)
    x = torch.randn(B, N, C)
    ref = O.vit_attention_forward(x[:2], sd, heads)
    m, xd = m.cuda(), x.cuda()
    p = F._prec(None)
    out = {}
    with torch.no_grad():
        m(xd)                                                       # first call converts the weights (cached 16-bit copies)
        tags = _tags(lambda: out.__setitem__("y", m(xd)))
        qkv16 = F.linear16(F.cast16(xd, p), F.weight16(m.qkv.weight, p), m.qkv.bias, out16=True, precision=p)
        yc = F.linear16(m._core(qkv16, True), F.weight16(m.proj.weight, p), m.proj.bias, precision=p)
    assert any(",x32>" in t for t in tags) and not any("cast16" in t for t in tags), tags
    assert torch.equal(out["y"], yc)
    assert_parity(out["y"][:2].cpu(), ref, 1e-3, "mhsa block (fused cast) vs oracle")
    from mi355attn.modules import XCA
    torch.manual_seed(7)
    xca = XCA(C, 8).eval().cuda()
    with torch.no_grad():
        xca(xd)
        tags = _tags(lambda: out.__setitem__("x", xca(xd)))
        old = mi355attn.get_option("gemm_wslab")
        try:
            mi355attn.set_option("gemm_wslab", 0)
            tags0 = _tags(lambda: out.__setitem__("x0", xca(xd)))
        finally:
            mi355attn.set_option("gemm_wslab", old)
    assert any(",x32>" in t for t in tags) and not any("cast16" in t for t in tags), tags
    assert any("cast16" in t for t in tags0) and len(tags0) == len(tags) + 1, (tags, tags0)
    assert torch.equal(out["x"], out["x0"])
