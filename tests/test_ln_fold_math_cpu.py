"""The algebra of the LayerNorm fold (csrc/ln_fold.hip) restated in torch on the CPU: what the three GPU pieces compute,
step by step, must reproduce act(LayerNorm(y) W^T + b) -- independent of the kernels, so a wrong identity cannot hide behind a matching
implementation.  Also the host-side weight preparations (`functional.lnfold_weights`, `functional.weight16_scaled`).

  producer   a = fp16(y - c)            c = the row's mean before the update;   per 32-column group (mean_g, M2_g)
  finalize   mu = mean_g(mean_g);  M2 = sum_g M2_g + 32 sum_g (mean_g - mu)^2;  r = rsqrt(M2 / C + eps);  rowtau = {r, r (c - mu)}
  consumer   y' = r * (a W'^T) + r (c - mu) * colsum + b',   W' = fp16(gamma * W),  colsum = sum_k W',  b' = b + W beta
"""
import torch

from conftest import rel_fro


def _emulate(y, c, ln, lin, F):
    C = y.shape[-1]
    a = (y - c[:, None]).half()
    g = y.view(y.shape[0], C // 32, 32)
    mean_g = g.mean(-1)
    m2_g = ((g - mean_g[..., None]) ** 2).sum(-1)
    mu = mean_g.mean(-1)
    m2 = m2_g.sum(-1) + 32.0 * ((mean_g - mu[:, None]) ** 2).sum(-1)
    r = torch.rsqrt(m2 / C + ln.eps)
    w16, colsum, bias = F.lnfold_weights(ln, lin, 1)
    acc = a.double() @ w16.double().t()
    out = r.double()[:, None] * acc + (r * (c - mu)).double()[:, None] * colsum.double()[None, :] + bias.double()[None, :]
    return out, mu, r


def test_fold_identity_matches_layernorm_then_linear():
    from mi355attn import functional as F
    torch.manual_seed(0)
    M, C, N = 96, 768, 320
    ln = torch.nn.LayerNorm(C)
    lin = torch.nn.Linear(C, N)
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.4 * torch.randn(C))
        ln.bias.copy_(0.3 * torch.randn(C))
    y_old = torch.randn(M, C) * 2.0 + torch.randn(M, 1) * 5.0              # row means of a few std: what fp16(y) would choke on
    y = y_old + 0.7 * torch.randn(M, C) + 0.2                              # one residual update
    c = y_old.mean(-1)
    out, mu, r = _emulate(y, c, ln, lin, F)
    yd = y.double()
    mu_ref = yd.mean(-1)
    var_ref = yd.var(-1, unbiased=False)
    assert torch.allclose(mu.double(), mu_ref, atol=1e-5)
    assert torch.allclose(r.double(), 1.0 / torch.sqrt(var_ref + ln.eps), rtol=1e-5)
    lnd = (yd - mu_ref[:, None]) / torch.sqrt(var_ref + ln.eps)[:, None] * ln.weight.double() + ln.bias.double()
    ref = lnd @ lin.weight.double().t() + lin.bias.double()
    # the unfolded 16-bit path: fp16(LayerNorm(y)) times fp16(W)
    unf = lnd.half().double() @ lin.weight.half().double().t() + lin.bias.double()
    e_fold, e_unf = rel_fro(out, ref), rel_fro(unf, ref)
    assert e_fold < 1e-3 and e_fold < 1.6 * e_unf + 1e-6, (e_fold, e_unf)
    # centring by 0 instead of the previous mean is what round 3 rejected: |mean| / std times the error
    a0 = y.half().double()
    w16, colsum, bias = F.lnfold_weights(ln, lin, 1)
    out0 = r.double()[:, None] * (a0 @ w16.double().t()) + (r * (0 - mu)).double()[:, None] * colsum.double()[None, :] + bias.double()[None, :]
    assert rel_fro(out0, ref) > 1.5 * e_fold, "the test rows no longer separate x - c from x: strengthen the row means"


def test_lnfold_weights_and_layerscale_fold():
    from mi355attn import functional as F
    torch.manual_seed(1)
    ln = torch.nn.LayerNorm(64)
    lin = torch.nn.Linear(64, 48, bias=False)                 # ViT's qkv has no bias: b' is W beta alone
    with torch.no_grad():
        ln.bias.copy_(torch.randn(64))
        ln.weight.copy_(torch.rand(64) + 0.5)
    w16, colsum, bias = F.lnfold_weights(ln, lin, 1)
    assert w16.dtype == torch.float16 and tuple(w16.shape) == (48, 64)
    assert torch.equal(w16, (lin.weight.detach() * ln.weight.detach()[None, :]).half())
    assert torch.allclose(colsum.double(), w16.double().sum(1), rtol=0, atol=1e-6)
    assert torch.allclose(bias.double(), lin.weight.detach().double() @ ln.bias.detach().double(), atol=1e-6)
    assert F.lnfold_weights(ln, lin, 1)[0] is w16, "cached per parameter version"
    with torch.no_grad():
        ln.weight.mul_(1.0e6)                                  # gamma * W leaves the fp16 range: the fold must decline
    assert F.lnfold_weights(ln, lin, 1) is None
    # LayerScale folded into a projection: y = resid + gamma * (x W^T + b) = resid + x (gamma W)^T + gamma b
    proj = torch.nn.Linear(32, 24)
    gamma = torch.rand(24) + 0.5
    gamma = torch.nn.Parameter(gamma)
    ws, bs = F.weight16_scaled(proj.weight, proj.bias, gamma, 2)
    assert ws.dtype == torch.bfloat16
    x = torch.randn(5, 32)
    want = gamma.detach() * (x @ proj.weight.detach().t() + proj.bias.detach())
    got = x @ ws.float().t() + bs
    assert rel_fro(got, want) < 8e-3


def test_layerscale_fold_declines_outside_the_fp16_normal_range():
    """XCiT's published deep initialisation (eta = 1e-5) puts gamma * W (~2e-7) into the fp16 subnormals; the fold must
    decline in fp16 (gamma then stays in the GEMM's fp32 epilogue) and still be taken in bf16 (fp32 exponent range) and for O(1) gammas."""
    from mi355attn import functional as F
    torch.manual_seed(3)
    proj = torch.nn.Linear(384, 384)
    for eta, want_fold in ((1.0, True), (0.1, True), (1e-3, False), (1e-5, False)):
        gamma = torch.nn.Parameter(eta * torch.ones(384))
        got = F.weight16_scaled(proj.weight, proj.bias, gamma, 1)
        assert (got is not None) == want_fold, eta
        assert F.weight16_scaled(proj.weight, proj.bias, gamma, 2) is not None, "bf16 always folds"
        if got is not None:                                    # the folded weights keep fp16's relative precision
            w = (gamma.detach()[:, None] * proj.weight.detach()).double()
            normal = w.abs() >= 2.0 ** -13                    # fp16 normal numbers: half an ulp = 2^-11 relative
            assert float(normal.double().mean()) > 0.97, "nearly all folded weights are fp16 normals when the fold is taken"
            assert float(((got[0].double() - w).abs() / w.abs())[normal].max()) <= 2.0 ** -11
    big = torch.nn.Parameter(1e7 * torch.ones(384))
    assert F.weight16_scaled(proj.weight, proj.bias, big, 1) is None, "gamma * W beyond the fp16 maximum"
    zero = torch.nn.Parameter(torch.zeros(384))
    ws = F.weight16_scaled(proj.weight, proj.bias, zero, 1)
    assert ws is not None and float(ws[0].abs().max()) == 0.0, "an all-zero LayerScale folds exactly"
