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
                                            (8192 + 16, 256, False, True), (12544, 384, False, False)])
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
