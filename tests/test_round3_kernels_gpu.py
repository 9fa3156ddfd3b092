"""Op-level checks of the kernels and fixes added in round 3:

  * two-accumulator persistent GEMM (gemm16_pa.hip, gemm_variant 16): bit identity with the round-1 tile kernel (variant 7) for every
    epilogue it takes, on shapes that exercise one tile per workgroup (serial drain only), several tiles (epilogue pieces riding in the
    next tile's main loop), a partial last round, bias / no bias, residual / none, GELU / none, fp16 and bf16; the fp64 product;
    the default dispatch picks it for fp32 + residual outputs; it refuses what it does not take;
  * persistent GEMM's split last round (gemm16_p8.hip): correct under hipGraph capture + replay with changing inputs (ADVICE round 2:
    a captured launch replays its per-launch tag), and a many-launch stress run with varying inputs against the unsplit result.
"""
import pytest
import torch

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _ref_linear(x16, w16, b, act, resid):
    y = x16.double() @ w16.double().t()
    if b is not None:
        y = y + b.double()
    if act:
        y = torch.nn.functional.gelu(y)
    if resid is not None:
        y = y + resid.double()
    return y


PA_SHAPES = [  # (M, N, K): M % 128 == 0, N % 256 == 0, K >= 640
    (128, 256, 640),                 # one tile, one workgroup: serial drain only
    (128 * 5, 512, 768),             # 10 tiles on 256 CUs: one tile each
    (128 * 300, 256, 640),           # 300 tiles: one full round + a partial one; tiles_n = 1
    (128 * 394, 768, 768),           # ViT-Base proj: 4.6 rounds
    (128 * 200, 768, 3072),          # long reduction (ViT-Base fc2 geometry)
    (128 * 37, 1024, 1152),
]


@pytest.mark.parametrize("prec,dt", [(1, torch.float16), (2, torch.bfloat16)])
@pytest.mark.parametrize("M,N,K", PA_SHAPES)
def test_two_accumulator_gemm_bit_identical_to_tile_kernel(M, N, K, prec, dt):
    import mi355attn
    from mi355attn import functional as F
    torch.manual_seed(M + N + K)
    x16 = torch.randn(M, K, device="cuda").to(dt)
    w16 = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
    b = torch.randn(N, device="cuda")
    resid = torch.randn(M, N, device="cuda")
    combos = [dict(bias=b, out16=True), dict(bias=b, act=F.ACT_GELU, out16=True), dict(bias=None, out16=True),
              dict(bias=b, resid=resid), dict(bias=b, resid=resid, act=F.ACT_GELU), dict(bias=None, resid=resid), dict(bias=b), dict(bias=None)]
    try:
        for kw in combos:
            mi355attn.set_option("gemm_variant", 7)
            y7 = F.linear16(x16, w16, precision=prec, **kw)
            mi355attn.set_option("gemm_variant", 16)
            y16 = F.linear16(x16, w16, precision=prec, **kw)
            y16b = F.linear16(x16, w16, precision=prec, **kw)
            assert torch.equal(y16, y16b), f"run-to-run difference with {sorted(kw)}"
            assert torch.equal(y7, y16), f"two-accumulator kernel differs from variant 7 with {sorted(kw)}"
        if M * N <= 3_000_000:
            ref = _ref_linear(x16.cpu(), w16.cpu(), b.cpu(), True, resid.cpu())
            assert_parity(F.linear16(x16, w16, b, act=F.ACT_GELU, resid=resid, precision=prec).cpu(), ref.float(), 2e-5, "fp64 product")
    finally:
        mi355attn.set_option("gemm_variant", 0)


def test_two_accumulator_gemm_is_the_default_for_fp32_residual_outputs_and_batch_independent():
    """Default dispatch at the ViT-Base proj / fc2 shapes: same bits as the forced variant 16 (i.e. that kernel ran), and a row's bits
    do not depend on the batch it sits in (no split-K anywhere on this path)."""
    import mi355attn
    from mi355attn import functional as F
    torch.manual_seed(3)
    for N, K in ((768, 768), (768, 3072)):
        M = 256 * 197
        x16 = torch.randn(M, K, device="cuda").half()
        w16 = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        b = torch.randn(N, device="cuda")
        resid = torch.randn(M, N, device="cuda")
        y0 = F.linear16(x16, w16, b, resid=resid, precision=1)
        try:
            mi355attn.set_option("gemm_variant", 16)
            y16 = F.linear16(x16, w16, b, resid=resid, precision=1)
        finally:
            mi355attn.set_option("gemm_variant", 0)
        assert torch.equal(y0, y16)
        Ms = 128 * 197                                        # half the batch: different tile -> workgroup assignment
        ys = F.linear16(x16[:Ms].contiguous(), w16, b, resid=resid[:Ms].contiguous(), precision=1)
        assert torch.equal(ys, y0[:Ms]), "rows depend on the batch they are computed in"


def test_two_accumulator_gemm_refuses_what_it_does_not_take():
    import mi355attn
    from mi355attn import functional as F
    x16 = torch.randn(130, 768, device="cuda").half()          # M % 128 != 0
    w16 = torch.randn(256, 768, device="cuda").half()
    try:
        mi355attn.set_option("gemm_variant", 16)
        with pytest.raises(RuntimeError, match="two-accumulator"):
            F.linear16(x16, w16, precision=1)
        with pytest.raises(RuntimeError, match="two-accumulator"):      # K too short for the nine epilogue K-tiles
            F.linear16(torch.randn(128, 512, device="cuda").half(), torch.randn(256, 512, device="cuda").half(), precision=1)
    finally:
        mi355attn.set_option("gemm_variant", 0)
    y = F.linear16(x16, w16, precision=1)                      # default dispatch falls through to the other kernels
    assert_parity(y.cpu(), (x16.double() @ w16.double().t()).float().cpu(), 2e-5, "fallback")


def test_split_k_gemm_under_graph_capture_and_replay():
    """A captured launch replays with the same kernel arguments.  The split last round keys its flags on a per-launch tag, so the
    launcher must not split under capture: replay twice with CHANGED inputs and compare with eager results."""
    import mi355attn
    from mi355attn import functional as F
    M, N, K = 50432, 768, 3072
    torch.manual_seed(11)
    x16 = torch.randn(M, K, device="cuda").half()
    w16 = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda")
    resid = torch.randn(M, N, device="cuda")
    try:
        mi355attn.set_option("gemm_pa", 0)                     # the persistent 256 x 256 kernel with its split last round is the default then
        mi355attn.set_option("gemm_splitk", 1)
        y_eager = F.linear16(x16, w16, b, resid=resid, precision=1)          # also warms the workspace cache outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                y_cap = F.linear16(x16, w16, b, resid=resid, precision=1)
        torch.cuda.current_stream().wait_stream(s)
        for rep in range(3):
            x16.copy_(torch.randn(M, K, device="cuda").half())
            resid.copy_(torch.randn(M, N, device="cuda"))
            g.replay()
            torch.cuda.synchronize()
            mi355attn.set_option("gemm_splitk", 0)
            y_ref = F.linear16(x16, w16, b, resid=resid, precision=1)
            mi355attn.set_option("gemm_splitk", 1)
            assert_parity(y_cap.cpu(), y_ref.cpu(), 2e-6, f"replay {rep}")
        # eager launches after the replays still work (the workspace history was not corrupted by the capture)
        y2 = F.linear16(x16, w16, b, resid=resid, precision=1)
        mi355attn.set_option("gemm_splitk", 0)
        y3 = F.linear16(x16, w16, b, resid=resid, precision=1)
        assert_parity(y2.cpu(), y3.cpu(), 2e-6, "eager after replay")
        assert mi355attn.lib().mi355_sync_status() == 0
    finally:
        mi355attn.set_option("gemm_pa", 1)
        mi355attn.set_option("gemm_splitk", 1)
    del y_eager


def test_split_k_gemm_stress_varying_inputs():
    """Forty back-to-back split launches on fresh inputs (the flag area is never zeroed: only the per-launch tag separates them), each
    compared with the unsplit kernel; every element, not a sample."""
    import mi355attn
    from mi355attn import functional as F
    M, N, K = 50432, 768, 3072
    torch.manual_seed(5)
    w16 = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda")
    try:
        mi355attn.set_option("gemm_pa", 0)
        worst = 0.0
        for it in range(40):
            x16 = torch.randn(M, K, device="cuda").half() * (1.0 + 0.1 * it)
            mi355attn.set_option("gemm_splitk", 1)
            ys = F.linear16(x16, w16, b, precision=1)
            mi355attn.set_option("gemm_splitk", 0)
            yu = F.linear16(x16, w16, b, precision=1)
            d = float((ys - yu).abs().max() / yu.abs().max())
            worst = max(worst, d)
            assert d < 2e-6, f"launch {it}: split result off by {d:.3e} (stale or half-written partial slab?)"
        assert mi355attn.lib().mi355_sync_status() == 0
    finally:
        mi355attn.set_option("gemm_pa", 1)
        mi355attn.set_option("gemm_splitk", 1)
