"""GPU tests (-m gpu): every C-ABI op against an fp64 torch-CPU restatement, across shapes and the three precisions.

Tolerances are relative Frobenius + max-abs (conftest.assert_parity):
    strict (bf16x3 split)  5e-5      fp16 operands  1e-3      bf16 operands  1.2e-2 (reported, out of the parity tolerance)
Shape/index paths are checked BIT-EXACTLY with a one-hot attention construction (see test_window_index_bit_exact).
"""
import math

import pytest
import torch

import oracle as O
from conftest import assert_parity

pytestmark = pytest.mark.gpu

TOL = {0: 5e-5, 1: 1e-3, 2: 1.2e-2}


def F():
    from mi355attn import functional
    return functional


def gelu64(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


# ---------------------------------------------------------------------------------------------- linear / GEMM engine
LINEAR_SHAPES = [(1, 1, 4), (5, 7, 12), (130, 70, 36), (257, 129, 100), (64, 1000, 768), (1000, 768, 768), (300, 2304, 768),
                 (50, 196, 256), (129, 130, 4)]


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", LINEAR_SHAPES)
def test_linear_plain(M, N, K, prec):
    torch.manual_seed(M * 31 + N * 7 + K)
    x, w, b = torch.randn(M, K), torch.randn(N, K) / math.sqrt(K), torch.randn(N)
    y = F().linear(x.cuda(), w.cuda(), b.cuda(), precision=prec).cpu()
    ref = (x.double() @ w.double().t() + b.double()).float()
    assert_parity(y, ref, TOL[prec], f"linear{(M, N, K)} p{prec}")


@pytest.mark.parametrize("prec", [0, 1])
def test_linear_epilogues(prec):
    torch.manual_seed(3)
    M, N, K = 200, 136, 72
    x, w, b = torch.randn(2, 100, K), torch.randn(N, K) / math.sqrt(K), torch.randn(N)
    gamma, resid = torch.rand(N) + 0.5, torch.randn(2, 100, N)
    f = F()
    z = x.double() @ w.double().t() + b.double()
    got = f.linear(x.cuda(), w.cuda(), b.cuda(), act=f.ACT_GELU, precision=prec).cpu()
    assert_parity(got, gelu64(z).float(), TOL[prec], "gelu")
    got = f.linear(x.cuda(), w.cuda(), None, resid=resid.cuda(), precision=prec).cpu()
    assert_parity(got, (x.double() @ w.double().t() + resid.double()).float(), TOL[prec], "resid, no bias")
    got = f.linear(x.cuda(), w.cuda(), b.cuda(), act=f.ACT_GELU, gamma=gamma.cuda(), resid=resid.cuda(), precision=prec).cpu()
    assert_parity(got, (resid.double() + gamma.double() * gelu64(z)).float(), TOL[prec], "gelu+gamma+resid")


def test_linear_row_strided_input():
    torch.manual_seed(4)
    tok = torch.randn(6, 5, 64).cuda()
    w, b = torch.randn(10, 64).cuda() / 8, torch.randn(10).cuda()
    view = tok[:, 0]                                   # stride (320, 1): consumed in place
    assert not view.is_contiguous()
    y = F().linear(view, w, b, precision=0).cpu()
    ref = (tok[:, 0].double().cpu() @ w.double().cpu().t() + b.double().cpu()).float()
    assert_parity(y, ref, TOL[0], "strided rows")


def test_linear_rejects_bad_shapes():
    from mi355attn import Mi355Error
    x, w = torch.randn(4, 6).cuda(), torch.randn(3, 6).cuda()          # K = 6 is not a multiple of 4
    with pytest.raises(Mi355Error):
        F().linear(x, w)
    with pytest.raises(ValueError):
        F().linear(torch.randn(4, 8).cuda(), w)


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("B,T,N,C", [(2, 8, 12, 16), (3, 256, 196, 512), (2, 196, 256, 512), (1, 130, 36, 132)])
def test_token_mix(B, T, N, C, prec):
    torch.manual_seed(B + T + N)
    w, x, b, r = torch.randn(T, N) / math.sqrt(N), torch.randn(B, N, C), torch.randn(T), torch.randn(B, T, C)
    f = F()
    got = f.token_mix(w.cuda(), x.cuda(), b.cuda(), act=f.ACT_GELU, precision=prec).cpu()
    z = torch.einsum("tn,bnc->btc", w.double(), x.double()) + b.double()[None, :, None]
    assert_parity(got, gelu64(z).float(), TOL[prec], "token_mix gelu")
    got = f.token_mix(w.cuda(), x.cuda(), b.cuda(), resid=r.cuda(), precision=prec).cpu()
    assert_parity(got, (z + r.double()).float(), TOL[prec], "token_mix resid")


@pytest.mark.parametrize("rows,cols", [(1, 4), (7, 64), (300, 384), (1000, 768), (33, 1000), (17, 1536), (5, 2052), (9, 50)])
def test_layernorm(rows, cols):
    torch.manual_seed(rows + cols)
    x, w, b = torch.randn(rows, cols) * 3 + 1, torch.randn(cols), torch.randn(cols)
    y = F().layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-5).cpu()
    ref = torch.nn.functional.layer_norm(x.double(), (cols,), w.double(), b.double(), 1e-5).float()
    assert_parity(y, ref, 2e-6, f"layernorm{(rows, cols)}")


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("B,Cin,HW,ps,E", [(2, 3, 32, 16, 64), (1, 3, 224, 16, 768), (3, 4, 24, 8, 132)])
def test_patch_embed(B, Cin, HW, ps, E, prec):
    torch.manual_seed(E)
    img = torch.randn(B, Cin, HW, HW)
    w, b = torch.randn(E, Cin, ps, ps) / math.sqrt(Cin * ps * ps), torch.randn(E)
    P = (HW // ps) ** 2
    cls, pos = torch.randn(E), torch.randn(P + 1, E)
    tok = F().patch_embed(img.cuda(), w.cuda(), b.cuda(), cls.cuda(), pos.cuda(), ps, precision=prec).cpu()
    ref = O.vit_patch_embed_forward(img, w, b, torch.float64)
    ref = torch.cat([ref, cls.double().expand(B, 1, E)], dim=1) + pos.double()
    assert_parity(tok, ref.float(), TOL[prec], "patch_embed")


@pytest.mark.parametrize("prec", [0, 1])
def test_patch_embed_without_cls_and_pos(prec):
    """MLP-Mixer's PatchEmbedding (mlps/mlp_mixer.py:11-23): same gather GEMM, no cls row, no position add."""
    torch.manual_seed(5)
    img = torch.randn(2, 3, 64, 64)
    w, b = torch.randn(96, 3, 16, 16) / math.sqrt(768), torch.randn(96)
    tok = F().patch_embed(img.cuda(), w.cuda(), b.cuda(), None, None, 16, precision=prec).cpu()
    assert tok.shape == (2, 16, 96)
    assert_parity(tok, O.vit_patch_embed_forward(img, w, b, torch.float64).float(), TOL[prec], "patch_embed[no cls]")


CONVS = [  # B, Cin, H, W, Cout, k, stride, pad
    (2, 3, 224, 224, 64, 7, 4, 2),        # CSWin stem (cswin.py:247): K = 147, padded to 148 weight columns
    (2, 64, 56, 56, 128, 3, 2, 1),        # Merge_Block 1 (cswin.py:224-233)
    (3, 256, 14, 14, 512, 3, 2, 1),       # Merge_Block 3: 14 -> 7
    (1, 5, 9, 11, 20, 3, 1, 1),           # ragged: odd sizes, stride 1, Cin not a multiple of 4
    (2, 8, 6, 6, 12, 3, 2, 0),            # no padding
    # the image layer of a narrow stem runs on the direct fp32 kernel (stem_conv.hip) when the input is NCHW
    (2, 3, 224, 224, 16, 3, 2, 1),        # XCiT ConvPatchEmbed layer 1 (xcit.py:97)
    (3, 3, 17, 23, 24, 3, 1, 1),          # odd output width (second pixel of the last pair missing), COUT = 32 template
    (1, 4, 9, 9, 20, 5, 2, 2),            # 4 input channels, 5x5
    (2, 1, 12, 10, 64, 3, 1, 0),          # single channel, 64 outputs, no padding
]


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("B,Cin,H,W,Cout,k,stride,pad", CONVS)
def test_conv2d_tokens(B, Cin, H, W, Cout, k, stride, pad, layout, prec):
    """Implicit-GEMM convolution against ATen conv2d in fp64; input NCHW (layout 0) or token-major (layout 1)."""
    torch.manual_seed(Cin * 7 + k)
    img = torch.randn(B, Cin, H, W)
    w, b = torch.randn(Cout, Cin, k, k) / math.sqrt(Cin * k * k), torch.randn(Cout)
    ref = torch.nn.functional.conv2d(img.double(), w.double(), b.double(), stride=stride, padding=pad)
    OH, OW = ref.shape[-2:]
    ref = ref.flatten(2).transpose(1, 2)
    wp = torch.nn.Parameter(w.cuda())
    if layout == 0:
        y, hw = F().conv2d_tokens(img.cuda(), wp, b.cuda(), k, stride, pad, 0, precision=prec)
    else:
        tokens = img.flatten(2).transpose(1, 2).contiguous().cuda()
        if Cin % 4:                                   # token-major gathers are 16-byte channel runs: documented restriction
            from mi355attn import Mi355Error
            with pytest.raises(Mi355Error, match="Cin % 4"):
                F().conv2d_tokens(tokens, wp, b.cuda(), k, stride, pad, 1, hw=(H, W), precision=prec)
            return
        y, hw = F().conv2d_tokens(tokens, wp, b.cuda(), k, stride, pad, 1, hw=(H, W), precision=prec)
    assert hw == (OH, OW) and y.shape == (B, OH * OW, Cout)
    assert_parity(y.cpu(), ref.float(), TOL[prec], f"conv2d_tokens[layout {layout}]")


def test_conv2d_tokens_border_is_zero_padding():
    """All-ones input and weights, no bias: every output equals the count of in-bounds taps (exact small integers)."""
    x = torch.ones(1, 4, 6, 6)
    w = torch.nn.Parameter(torch.ones(8, 4, 3, 3).cuda())
    ref = torch.nn.functional.conv2d(x, torch.ones(8, 4, 3, 3), None, stride=2, padding=1).flatten(2).transpose(1, 2)
    for layout, inp, hw in ((0, x.cuda(), None), (1, x.flatten(2).transpose(1, 2).contiguous().cuda(), (6, 6))):
        y, _ = F().conv2d_tokens(inp, w, None, 3, 2, 1, layout, hw=hw, precision=1)
        assert torch.equal(y.cpu(), ref)


@pytest.mark.parametrize("B,N,C,skip", [(2, 196, 512, 0), (3, 197, 768, 1), (1, 49, 512, 0), (2, 5, 36, 2), (1, 1, 4, 0), (2, 3136, 64, 0)])
def test_token_mean(B, N, C, skip):
    torch.manual_seed(N)
    x = torch.randn(B, N, C)
    y = F().token_mean(x.cuda(), skip_first=skip).cpu()
    assert_parity(y, x[:, skip:].double().mean(dim=1).float(), 1e-6, "token_mean")


@pytest.mark.parametrize("act", [0, 1])
def test_conv2d_tokens_epilogue_pos_and_gelu(act):
    """act(conv + bias + pos): the position rows are indexed by output token, shared by all images (XCiT patch embedding)."""
    torch.manual_seed(3)
    B, Cin, H, W, Cout = 3, 8, 12, 10, 24
    img = torch.randn(B, Cin, H, W)
    w, b = torch.randn(Cout, Cin, 3, 3) / math.sqrt(Cin * 9), torch.randn(Cout)
    ref = torch.nn.functional.conv2d(img.double(), w.double(), b.double(), stride=2, padding=1)
    OH, OW = ref.shape[-2:]
    pos = torch.randn(OH * OW, Cout)
    ref = ref.flatten(2).transpose(1, 2) + pos.double()
    if act:
        ref = gelu64(ref)
    tokens = img.flatten(2).transpose(1, 2).contiguous().cuda()
    y, _ = F().conv2d_tokens(tokens, torch.nn.Parameter(w.cuda()), b.cuda(), 3, 2, 1, 1, hw=(H, W), precision=0, act=act, pos=pos.cuda())
    assert_parity(y.cpu(), ref.float(), TOL[0], "conv2d_tokens[pos, act]")


def test_conv_bn_folding_matches_eval_batchnorm():
    torch.manual_seed(4)
    conv = torch.nn.Conv2d(6, 16, 3, stride=2, padding=1, bias=False)
    bn = torch.nn.BatchNorm2d(16).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.normal_(1, 0.3); bn.bias.normal_(0, 0.3)
    x = torch.randn(2, 6, 9, 9)
    with torch.no_grad():
        ref = bn(conv(x)).flatten(2).transpose(1, 2)
    conv, bn = conv.cuda(), bn.cuda()
    wrows, bias = F().conv_bn_rows(conv.weight, bn, 0)
    y, _ = F().conv2d_tokens(x.cuda(), None, bias, 3, 2, 1, 0, precision=0, wrows=wrows)
    assert_parity(y.cpu(), ref, TOL[0], "conv + folded BatchNorm")


@pytest.mark.parametrize("B,N,h,d", [(3, 197, 4, 32), (2, 50, 8, 16), (1, 1, 2, 64), (2, 300, 3, 48), (5, 64, 4, 32)])
def test_class_attention_core(B, N, h, d):
    """One query per (image, head) against fp64; q/k/v are consumed in place from the fused (B,N,3C) projection."""
    torch.manual_seed(N + d)
    C = h * d
    qkv = torch.randn(B, N, 3 * C)
    scale = d ** -0.5
    q = qkv[:, 0, :C].double().reshape(B, h, 1, d)
    k = qkv[:, :, C:2 * C].double().reshape(B, N, h, d).permute(0, 2, 1, 3)
    v = qkv[:, :, 2 * C:].double().reshape(B, N, h, d).permute(0, 2, 1, 3)
    a = torch.softmax((q * k).sum(-1) * scale, dim=-1)
    ref = (a.unsqueeze(2) @ v).transpose(1, 2).reshape(B, C)
    dev = qkv.cuda()
    out = F().class_attention(dev[:, 0, :C], dev[:, :, C:2 * C], dev[:, :, 2 * C:], h, scale, N, N * 3 * C, 3 * C)
    assert_parity(out.cpu(), ref.float(), 2e-6, "class attention")


@pytest.mark.parametrize("B,H,W,C,sr,with_bias", [(2, 56, 56, 64, 8, True), (3, 14, 14, 320, 2, True), (1, 8, 12, 20, 4, False), (2, 6, 6, 8, 1, True)])
def test_dwconv_patch_tokens(B, H, W, C, sr, with_bias):
    """Depth-wise kernel == stride conv (+ bias) + eval BatchNorm on the token grid against ATen in fp64."""
    torch.manual_seed(C + sr)
    conv = torch.nn.Conv2d(C, C, sr, stride=sr, groups=C, bias=with_bias)
    bn = torch.nn.BatchNorm2d(C).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.normal_(1, 0.3); bn.bias.normal_(0, 0.3)
    x = torch.randn(B, H * W, C)
    with torch.no_grad():
        ref = bn.double()(conv.double()(x.double().transpose(1, 2).reshape(B, C, H, W))).reshape(B, C, -1).transpose(1, 2)
    conv, bn = conv.float().cuda(), bn.float().cuda()
    y = F().dwconv_patch_tokens(x.cuda(), conv.weight, conv.bias, bn, H, W, sr)
    assert_parity(y.cpu(), ref.float(), 2e-6, "dwconv_patch_tokens")


def test_axpby_strided_rows():
    """y = alpha*x + gamma*u on strided row views: cls-row gather / scatter, broadcast row, token-block copy, scalar tail."""
    torch.manual_seed(9)
    B, N, C = 3, 7, 20
    x, u, g = torch.randn(B, N, C), torch.randn(B, C), torch.randn(C)
    xd, ud, gd = x.cuda(), u.cuda(), g.cuda()
    c1 = torch.empty(B, C, device="cuda")
    F().axpby(xd, c1, B, C, N * C, C, alpha=0.5, u=ud, ldu=C, gamma=gd)                  # gather cls rows + scaled residual
    assert_parity(c1.cpu(), 0.5 * x[:, 0] + g * u, 1e-6, "axpby gather")
    out = torch.zeros(B, N + 1, C, device="cuda")
    F().axpby(gd, out, B, C, 0, (N + 1) * C)                                               # broadcast one row into every image
    F().axpby(xd, out[:, 1:], B, N * C, N * C, (N + 1) * C)                                # token blocks behind it
    assert torch.equal(out.cpu(), torch.cat([g.expand(B, 1, C), x], dim=1))
    odd = torch.randn(5, 9)
    y = torch.empty(5, 9, device="cuda")
    F().axpby(odd.cuda(), y, 5, 9, 9, 9, alpha=2.0)                                        # cols % 4 != 0: scalar path
    assert torch.equal(y.cpu(), 2.0 * odd)


# ---------------------------------------------------------------------------------------------- attention cores
def _sdpa_ref(qkv, h, scale):
    B, N, C3 = qkv.shape
    C = C3 // 3
    d = C // h
    q, k, v = (qkv.double().reshape(B, N, 3, h, d).permute(2, 0, 3, 1, 4)[i] for i in range(3))
    return O.sdpa_core(q, k, v, scale).transpose(1, 2).reshape(B, N, C)


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("B,N,h,d", [(2, 197, 12, 64), (1, 5, 2, 32), (3, 64, 4, 32), (2, 100, 3, 64), (1, 224, 2, 64),
                                    (2, 17, 1, 64), (2, 1, 2, 32), (1, 129, 2, 32)])
def test_sdpa(B, N, h, d, prec):
    torch.manual_seed(N * 3 + h)
    qkv = torch.randn(B, N, 3 * h * d)
    out = F().sdpa(qkv.cuda(), h, d ** -0.5, precision=prec).cpu()
    assert_parity(out, _sdpa_ref(qkv, h, d ** -0.5).float(), TOL[prec], f"sdpa{(B, N, h, d)} p{prec}")


GENERAL = [  # B, Nq, Nkv, heads, d, bias
    (2, 197, 197, 3, 64, False),      # ViT-like
    (2, 784, 49, 2, 64, False),       # PVT / SegFormer: keys from a spatially reduced grid
    (1, 1024, 1024, 2, 64, False),    # SETR at 512x512: 16 key tiles
    (2, 50, 300, 4, 32, True),        # ragged: partial query block, partial last key tile, additive bias
    (3, 1, 77, 2, 32, False),         # single query
    (2, 196, 49, 4, 32, True),        # CMT: relative-position term, N_kv % 4 != 0 -> scalar bias loads
    (1, 65, 64, 1, 64, True),         # exactly one key tile, N_kv % 4 == 0 -> vector bias loads
]


def _general_ref(q, k, v, h, scale, bias):
    B, Nq, C = q.shape
    d = C // h
    qh = q.double().reshape(B, Nq, h, d).permute(0, 2, 1, 3)
    kh = k.double().reshape(B, -1, h, d).permute(0, 2, 1, 3)
    vh = v.double().reshape(B, -1, h, d).permute(0, 2, 1, 3)
    att = qh @ kh.transpose(-1, -2) * scale
    if bias is not None:
        att = att + bias.double()
    return (torch.softmax(att, dim=-1) @ vh).transpose(1, 2).reshape(B, Nq, C)


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("B,Nq,Nkv,h,d,with_bias", GENERAL)
def test_sdpa_general(B, Nq, Nkv, h, d, with_bias, prec):
    """Streaming attention core vs fp64: N_q != N_kv, partial tiles, optional additive bias; q sliced out of a fused (B,N,3C)
    projection when N_q == N_kv (strided rows), separate tensors otherwise."""
    torch.manual_seed(Nq * 7 + Nkv)
    C = h * d
    scale = d ** -0.5
    if Nq == Nkv:
        qkv = torch.randn(B, Nq, 3 * C)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        dev = qkv.cuda()
        qd, kd, vd = dev[..., :C], dev[..., C:2 * C], dev[..., 2 * C:]
    else:
        q, k, v = torch.randn(B, Nq, C), torch.randn(B, Nkv, C), torch.randn(B, Nkv, C)
        qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    bias = torch.randn(h, Nq, Nkv) if with_bias else None
    out = F().sdpa_general(qd, kd, vd, h, scale, bias=None if bias is None else bias.cuda(), precision=prec).cpu()
    assert_parity(out, _general_ref(q, k, v, h, scale, bias).float(), TOL[prec], "sdpa_general")


@pytest.mark.parametrize("prec,dt", [(1, torch.float16), (2, torch.bfloat16)])
def test_sdpa_general_16bit_io_and_batched_bias(prec, dt):
    torch.manual_seed(5)
    B, Nq, Nkv, h, d = 2, 130, 200, 2, 64
    C = h * d
    q, k, v = (torch.randn(B, n, C).to(dt) for n in (Nq, Nkv, Nkv))
    bias = torch.randn(B, h, Nq, Nkv)
    out = F().sdpa_general(q.cuda(), k.cuda(), v.cuda(), h, 0.125, bias=bias.cuda(), precision=prec)
    assert out.dtype == dt
    ref = _general_ref(q.float(), k.float(), v.float(), h, 0.125, bias)
    assert_parity(out.float().cpu(), ref.float(), TOL[prec], "sdpa_general[16-bit io]")


def test_sdpa_general_matches_short_sequence_kernel():
    """Same problem through the all-keys-in-LDS kernel (mi355_sdpa_fwd) and the streaming kernel: both within tolerance of each other."""
    torch.manual_seed(6)
    qkv = torch.randn(2, 197, 3 * 768).cuda()
    a = F().sdpa(qkv, 12, 0.125, precision=0)
    b = F().sdpa_general(qkv[..., :768], qkv[..., 768:1536], qkv[..., 1536:], 12, 0.125, precision=0)
    assert_parity(b.cpu(), a.cpu(), 2e-5, "streaming vs resident")


def test_sdpa_sharp_softmax_and_large_logits():
    """Rows whose max dwarfs the rest (one spiked key) and logits ~ +-60: the masked/shifted softmax must stay finite."""
    torch.manual_seed(1)
    B, N, h, d = 1, 197, 2, 64
    qkv = torch.randn(B, N, 3 * h * d)
    qkv[0, 7, 0:d] *= 25.0
    qkv[0, 100, h * d:h * d + d] *= 25.0
    out = F().sdpa(qkv.cuda(), h, d ** -0.5, precision=0).cpu()
    assert_parity(out, _sdpa_ref(qkv, h, d ** -0.5).float(), 2e-4, "sdpa spiked")


def _onehot_codes(T, d, scale=48.0):
    """q_t = scale * code(t), k_s = code(s) with +-1 bit codes: q_t.k_s is maximal only at s == t, by >= 2*scale."""
    bits = max(1, (T - 1).bit_length())
    assert bits <= d
    idx = torch.arange(T)
    code = torch.zeros(T, d)
    for b in range(bits):
        code[:, b] = ((idx >> b) & 1).float() * 2 - 1
    return code * scale, code


WINDOWS = [(56, 0, 1, 32, 1), (56, 1, 1, 32, 1), (28, 0, 2, 64, 2), (28, 1, 2, 64, 2), (14, 0, 7, 128, 4), (14, 1, 7, 128, 4),
           (7, -1, 7, 512, 16), (8, 0, 2, 64, 2), (8, 1, 4, 32, 1), (6, -1, 6, 64, 2)]


@pytest.mark.parametrize("reso,idx,split,dim,heads", WINDOWS)
def test_window_index_bit_exact(reso, idx, split, dim, heads):
    """Stripe-window gather / head split / scatter are pure index math: with a one-hot attention (q_t.k_s peaks only at
    s == t inside every window) and LePE switched off, out must equal v BIT FOR BIT (values are integers < 2^16, exact
    in the strict operand format)."""
    from mi355attn.modules import LePEAttention
    torch.manual_seed(0)
    m = LePEAttention(dim, reso, idx, split_size=split, num_heads=heads, precision=0).eval()
    torch.nn.init.zeros_(m.get_v.weight)
    torch.nn.init.zeros_(m.get_v.bias)
    B, L, d = 2, reso * reso, dim // heads
    T = m.H_sp * m.W_sp
    tab = O.window_token_index(reso, m.H_sp, m.W_sp)                      # (nWin, T) token ids
    qc, kc = _onehot_codes(T, d)
    q = torch.zeros(B, L, dim)
    k = torch.zeros(B, L, dim)
    for w in range(tab.shape[0]):
        for hh in range(heads):
            q[:, tab[w], hh * d:(hh + 1) * d] = qc / m.scale               # kernel pre-scales q by m.scale
            k[:, tab[w], hh * d:(hh + 1) * d] = kc
    lidx, cidx, bidx = torch.arange(L).float()[None, :, None], torch.arange(dim).float()[None, None, :], \
        torch.arange(B).float()[:, None, None]
    patterns = [lidx * 16 + cidx % 16 + bidx * 7,                 # distinguishes every token
                cidx + (lidx % 64) * 512 + bidx * 0]              # distinguishes every channel (head split / merge)
    m = m.cuda()
    for v in patterns:
        v = v.expand(B, L, dim).contiguous()
        assert float(v.max()) < 65536
        qkv = torch.stack([q, k, v], dim=0)
        with torch.no_grad():
            out = m(qkv.cuda()).cpu()
        assert torch.equal(out, v), "window / head index math is not bit-exact"


def test_vit_head_index_bit_exact():
    B, N, h, d = 2, 197, 12, 64
    qc, kc = _onehot_codes(N, d)
    scale = d ** -0.5
    q = (qc / scale).repeat(1, h)
    k = kc.repeat(1, h)
    v = torch.arange(N).float()[:, None] * 64 + (torch.arange(h * d).float()[None, :] % 64)
    qkv = torch.cat([q, k, v], dim=1)[None].repeat(B, 1, 1).contiguous()
    qkv[1, :, 2 * h * d:] += 3
    out = F().sdpa(qkv.cuda(), h, scale, precision=0).cpu()
    assert torch.equal(out, qkv[:, :, 2 * h * d:]), "ViT head split / merge is not bit-exact"


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("reso,idx,split,dim,heads", WINDOWS)
def test_lepe_attention_vs_oracle(reso, idx, split, dim, heads, prec):
    from mi355attn.modules import LePEAttention
    torch.manual_seed(reso + dim)
    m = LePEAttention(dim, reso, idx, split_size=split, num_heads=heads, precision=prec).eval()
    qkv = torch.randn(3, 2, reso * reso, dim)
    ref = O.lepe_attention_forward(qkv, m.get_v.weight, m.get_v.bias, reso, idx, split, heads, torch.float64)
    with torch.no_grad():
        out = m.cuda()(qkv.cuda()).cpu()
    assert_parity(out, ref.float(), TOL[prec], f"lepe{(reso, idx, split, dim, heads)} p{prec}")


def test_lepe_accepts_permuted_view_without_copy():
    from mi355attn.modules import LePEAttention
    torch.manual_seed(5)
    m = LePEAttention(32, 8, 0, split_size=2, num_heads=1, precision=0).eval().cuda()
    buf = torch.randn(2, 64, 3, 32).cuda()
    view = buf.permute(2, 0, 1, 3)                                        # the reference's (3,B,L,C) view of (B,L,3,C)
    with torch.no_grad():
        a = m(view)
        b = m(view.contiguous())
    assert torch.equal(a, b)


@pytest.mark.parametrize("B,N,h,d", [(2, 196, 8, 48), (1, 49, 4, 32), (2, 10, 2, 64), (1, 224, 2, 48), (3, 196, 4, 32), (1, 784, 2, 48),
                                    (2, 64, 3, 64), (1, 65, 1, 32), (1, 1, 2, 48)])
def test_xca_core(B, N, h, d):
    torch.manual_seed(N + d)
    C = h * d
    qkv = torch.randn(B, N, 3 * C)
    temp = torch.rand(h) + 0.5
    out = F().xca_core(qkv.cuda(), temp.cuda(), h).cpu()
    ref = _xca_core_ref(qkv.double(), temp.double(), h)
    assert_parity(out, ref.float(), 2e-5, f"xca{(B, N, h, d)}")


def _xca_core_ref(qkv, temp, h):
    B, N, C3 = qkv.shape
    C = C3 // 3
    d = C // h
    q, k, v = (qkv.reshape(B, N, 3, h, d).permute(2, 0, 3, 4, 1)[i] for i in range(3))    # (B,h,d,N)
    qn = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    kn = k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    a = torch.softmax((qn @ kn.transpose(-1, -2)) * temp.reshape(1, h, 1, 1), dim=-1)
    return (a @ v).permute(0, 3, 1, 2).reshape(B, N, C)


@pytest.mark.parametrize("B,H,W,C", [(2, 14, 14, 384), (1, 7, 7, 40), (3, 4, 9, 100), (1, 1, 1, 32), (2, 5, 3, 30), (1, 14, 14, 130)])
def test_lpi(B, H, W, C):
    torch.manual_seed(H * W + C)
    x = torch.randn(B, H * W, C)
    p = {"conv1.weight": torch.randn(C, 1, 3, 3) / 3, "conv1.bias": torch.randn(C), "bn.weight": torch.rand(C) + 0.5,
         "bn.bias": torch.randn(C), "bn.running_mean": torch.randn(C) * 0.1, "bn.running_var": torch.rand(C) + 0.5,
         "conv2.weight": torch.randn(C, 1, 3, 3) / 3, "conv2.bias": torch.randn(C)}
    gamma, resid = torch.rand(C), torch.randn(B, H * W, C)
    ref = O.lpi_forward(x, p, H, W, torch.float64)
    d = {k: v.cuda() for k, v in p.items()}
    got = F().lpi(x.cuda(), d["conv1.weight"], d["conv1.bias"], d["bn.weight"], d["bn.bias"], d["bn.running_mean"],
                  d["bn.running_var"], 1e-5, d["conv2.weight"], d["conv2.bias"], H, W).cpu()
    assert_parity(got, ref.float(), 5e-6, "lpi")
    got = F().lpi(x.cuda(), d["conv1.weight"], d["conv1.bias"], d["bn.weight"], d["bn.bias"], d["bn.running_mean"],
                  d["bn.running_var"], 1e-5, d["conv2.weight"], d["conv2.bias"], H, W, gamma=gamma.cuda(),
                  resid=resid.cuda()).cpu()
    assert_parity(got, (resid.double() + gamma.double() * ref).float(), 5e-6, "lpi gamma+resid")


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("B,C,cm,cn,H,W", [(2, 64, 32, 32, 32, 32), (2, 32, 16, 8, 8, 8), (1, 256, 128, 128, 14, 14), (3, 16, 4, 12, 6, 6),
                                          (3, 64, 32, 32, 8, 8), (1, 64, 32, 32, 16, 24), (5, 64, 32, 32, 4, 8), (2, 64, 32, 32, 8, 12)])   # one-kernel path (double_attn_small.hip): 2 .. 12 pixel groups, fewer groups than waves
def test_double_attention(B, C, cm, cn, H, W, prec):
    from mi355attn.modules import DoubleAttention
    torch.manual_seed(C + cm)
    m = DoubleAttention(C, cm, cn, precision=prec).eval()
    x = torch.randn(B, C, H, W)
    sd = m.state_dict()
    ref = O.double_attention_forward(x, sd["convA.weight"], sd["convA.bias"], sd["convB.weight"], sd["convB.bias"],
                                     sd["convV.weight"], sd["convV.bias"], sd["proj.weight"], sd["proj.bias"], torch.float64)
    with torch.no_grad():
        y = m.cuda()(x.cuda()).cpu()
    assert_parity(y, ref.float(), TOL[prec], f"double_attention p{prec}")


# ---------------------------------------------------------------------------------------------- 16-bit activation dataflow
@pytest.mark.parametrize("prec,dt", [(1, torch.float16), (2, torch.bfloat16)])
def test_cast16_is_round_to_nearest_even(prec, dt):
    torch.manual_seed(0)
    x = torch.cat([torch.randn(100003) * 3, torch.tensor([0.0, -0.0, 1e-8, 65504.0, 1e-5, 3.0e38 if prec == 2 else 6e4])]).cuda()
    got = F().cast16(x, prec)
    assert got.dtype == dt and torch.equal(got, x.to(dt))


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("M,N,K", [(1000, 768, 768), (130, 72, 64), (257, 132, 192), (64, 2304, 768), (50, 64, 256),
                                   # short-K weight-stationary kernel (M >= 2048, K in {64,128}), incl. ragged M and N
                                   (4096, 192, 64), (3000, 64, 64), (2049, 256, 64), (5000, 384, 128), (2500, 128, 128), (2100, 68, 128),
                                   (2304, 520, 64)])
def test_linear16_bit_identical_to_fp32_entry(M, N, K, prec):
    """Same rounding point, same accumulation order: the 16-bit-operand GEMM must reproduce mi355_linear_fwd exactly."""
    torch.manual_seed(M + N + K)
    f = F()
    x, w, b = torch.randn(M, K).cuda(), (torch.randn(N, K) / math.sqrt(K)).cuda(), torch.randn(N).cuda()
    gamma, resid = (torch.rand(N) + 0.5).cuda(), torch.randn(M, N).cuda()
    x16, w16 = f.cast16(x, prec), f.cast16(w, prec)
    ref = f.linear(x, w, b, precision=prec)
    assert torch.equal(f.linear16(x16, w16, b, precision=prec), ref)
    ref2 = f.linear(x, w, b, act=f.ACT_GELU, gamma=gamma, resid=resid, precision=prec)
    assert torch.equal(f.linear16(x16, w16, b, act=f.ACT_GELU, gamma=gamma, resid=resid, precision=prec), ref2)
    # 16-bit outputs without an activation round the same fp32 number; with GELU the 16-bit epilogues use the short form of the
    # function (csrc/common.h gelu16_fast: |error| <= 9e-7 before the rounding), so at most a rounding boundary may be crossed
    out16 = f.linear16(x16, w16, b, out16=True, precision=prec)
    assert torch.equal(out16, f.linear(x, w, b, precision=prec).to(out16.dtype))
    out16 = f.linear16(x16, w16, b, act=f.ACT_GELU, out16=True, precision=prec)
    ref3 = f.linear(x, w, b, act=f.ACT_GELU, precision=prec)
    ulp = 2.0 ** -10 if prec == 1 else 2.0 ** -7
    assert torch.all((out16.float() - ref3).abs() <= ulp * ref3.abs() + 2e-6)
    assert (out16 != ref3.to(out16.dtype)).float().mean() < 0.02


def test_linear16_rejects_shapes_outside_the_envelope():
    from mi355attn import Mi355Error
    f = F()
    x16, w16 = torch.randn(8, 96).cuda().half(), torch.randn(16, 96).cuda().half()       # K = 96 is not a multiple of 64
    with pytest.raises(Mi355Error):
        f.linear16(x16, w16, precision=1)


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("B,N,C,NP", [(3, 196, 512, 256), (2, 50, 132, 64), (1, 64, 64, 64), (2, 33, 1024, 96), (5, 7, 4, 32)])
def test_layernorm16_transposed(B, N, C, NP, prec):
    """LayerNorm written channel-major per image in 16 bit, zero-filled up to NP tokens (the Mixer token-mixing operand)."""
    torch.manual_seed(B * N + C)
    x, w, b = (torch.randn(B, N, C) * 2 + 0.5).cuda(), torch.randn(C).cuda(), torch.randn(C).cuda()
    ut = F().layernorm16_t(x, w, b, 1e-5, NP, prec)
    assert ut.shape == (B, C, NP) and torch.count_nonzero(ut[:, :, N:]) == 0
    ref = torch.nn.functional.layer_norm(x.double().cpu(), (C,), w.double().cpu(), b.double().cpu(), 1e-5).transpose(1, 2)
    assert_parity(ut[:, :, :N].float().cpu(), ref.float(), 6e-4 if prec == 1 else 5e-3, f"ln16_t{(B, N, C)} p{prec}")


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("B,C,N,K", [(3, 512, 196, 256), (2, 132, 50, 64), (1, 64, 7, 128), (5, 4, 300, 64), (2, 260, 129, 192)])
def test_linear16_transposed_output(B, C, N, K, prec):
    """Same products as mi355_linear16_fwd, written (B, N, C) instead of (B, C, N), bias per n, residual in the output layout."""
    torch.manual_seed(B + C + N + K)
    f = F()
    xt16 = f.cast16(torch.randn(B, C, K).cuda(), prec)
    w16 = f.cast16((torch.randn(N, K) / math.sqrt(K)).cuda(), prec)
    bias, resid = torch.randn(N).cuda(), torch.randn(B, N, C).cuda()
    got = f.linear16_tr(xt16, w16, bias, resid, prec)
    ref = xt16.double().cpu() @ w16.double().cpu().t() + bias.double().cpu()
    assert_parity(got.cpu(), (ref.transpose(1, 2) + resid.double().cpu()).float(), 1e-5, f"linear16_tr{(B, C, N, K)} p{prec}")
    if N % 4 == 0:
        plain = f.linear16(xt16, w16, bias, precision=prec)
        assert torch.equal(got, plain.transpose(1, 2) + resid)
    got = f.linear16_tr(xt16, w16, None, None, prec)
    assert_parity(got.cpu(), (xt16.double().cpu() @ w16.double().cpu().t()).transpose(1, 2).float(), 1e-5, "no bias / resid")


def test_linear16_transposed_rejects_bad_rows():
    from mi355attn import Mi355Error
    f = F()
    with pytest.raises(Mi355Error):
        f.linear16_tr(torch.randn(2, 6, 64).cuda().half(), torch.randn(8, 64).cuda().half(), None, None, 1)     # C % 4 != 0


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("rows,cols", [(300, 384), (1000, 768), (802, 64), (33, 128), (9, 52)])
def test_layernorm16(rows, cols, prec):
    torch.manual_seed(rows)
    x, w, b = (torch.randn(rows, cols) * 2 + 0.5).cuda(), torch.randn(cols).cuda(), torch.randn(cols).cuda()
    y32 = F().layernorm(x, w, b, 1e-5)
    y16 = F().layernorm16(x, w, b, 1e-5, prec)
    assert torch.equal(y16, y32.to(y16.dtype))


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("B,N,h,d", [(2, 197, 12, 64), (3, 64, 4, 32), (1, 130, 2, 64)])
def test_sdpa16_matches_fp32_io_kernel(B, N, h, d, prec):
    torch.manual_seed(N)
    f = F()
    qkv16 = f.cast16(torch.randn(B, N, 3 * h * d).cuda(), prec)
    ref = f.sdpa(qkv16.float(), h, d ** -0.5, precision=prec)
    got = f.sdpa16(qkv16, h, d ** -0.5, precision=prec)
    assert torch.equal(got, f.sdpa16(qkv16, h, d ** -0.5, precision=prec)), "run-to-run results differ"
    # same math in both kernels; the only admissible difference is the final fp32 -> 16-bit rounding of values that sit on a
    # rounding tie (the two template instantiations may differ in the last fp32 bit): <= 1 ulp of the 16-bit type, and rare
    ref16 = ref.to(got.dtype)
    diff = (got.float() - ref16.float()).abs()
    ulp = 2.0 ** (-10 if prec == 1 else -7)
    assert float((diff / ref16.float().abs().clamp_min(1e-3)).max()) <= ulp * 1.01
    assert float((diff > 0).float().mean()) < 2e-3


def test_weight16_cache_tracks_inplace_updates():
    f = F()
    w = torch.nn.Parameter(torch.randn(8, 64).cuda())
    a = f.weight16(w, 1)
    assert f.weight16(w, 1) is a                         # cached
    with torch.no_grad():
        w.mul_(2.0)                                      # version bump -> re-converted
    b = f.weight16(w, 1)
    assert b is not a and torch.equal(b, w.detach().half())
    assert f.weight16(w, 2).dtype == torch.bfloat16


def test_derived_weight_cache_is_tied_to_the_live_parameter():
    """A deleted model must not leave 16-bit / re-laid-out weights behind for a new parameter that happens to get the same id, address
    and version counter: the cache entry dies with its parameter (weak reference) and the copies follow the new values."""
    import gc
    from mi355attn import functional as Fm
    torch.manual_seed(0)
    seen = 0
    for trial in range(20):
        w = torch.nn.Parameter(torch.randn(64, 64, device="cuda") * (trial + 1))
        w16 = Fm.weight16(w, 1)
        assert torch.equal(w16.float(), w.detach().half().float())
        seen = max(seen, len(Fm._derived))
        del w, w16
        gc.collect()
    assert len(Fm._derived) <= seen and all(all(r() is not None for r in e[0]) for e in Fm._derived.values())


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("C", [64, 128])
@pytest.mark.parametrize("M,with_ln,with_gamma", [(802, True, False), (33, True, True), (4096, False, True), (1, True, False), (70000, True, True)])
def test_mlp_fused(M, with_ln, with_gamma, prec, C):
    """LayerNorm + fc1 + GELU + fc2 + residual in one kernel (C = 64: LDS-resident weights; C = 128: weights streamed in 32-unit
    slices) against fp64; ragged token counts exercise partial chunks / idle waves, gamma the LayerScale variant."""
    torch.manual_seed(M)
    Hd = 4 * C
    ln = torch.nn.LayerNorm(C)
    fc1, fc2 = torch.nn.Linear(C, Hd), torch.nn.Linear(Hd, C)
    with torch.no_grad():
        ln.weight.normal_(1, 0.2); ln.bias.normal_(0, 0.2)
    gamma = torch.randn(C) if with_gamma else None
    x = torch.randn(M, C)
    xd = x.double()
    u = torch.nn.functional.layer_norm(xd, (C,), ln.weight.double(), ln.bias.double(), ln.eps) if with_ln else xd
    h = gelu64(u @ fc1.weight.double().t() + fc1.bias.double())
    z = h @ fc2.weight.double().t() + fc2.bias.double()
    ref = (xd + (z * gamma.double() if with_gamma else z)).detach()
    y = F().mlp_fused(x.cuda(), ln.cuda() if with_ln else None, fc1.cuda(), fc2.cuda(), gamma=None if gamma is None else gamma.cuda(),
                      precision=prec)
    assert_parity(y.cpu(), ref.float(), TOL[prec], "mlp_fused")


def test_mlp_fused_rejects_other_shapes():
    from mi355attn import Mi355Error
    fc1, fc2 = torch.nn.Linear(256, 1024).cuda(), torch.nn.Linear(1024, 256).cuda()
    with pytest.raises(Mi355Error, match="C = 64"):
        F().mlp_fused(torch.randn(8, 256).cuda(), None, fc1, fc2, precision=1)


# ---------------------------------------------------------------------------------------------- glue of the f1 / f2 rows
@pytest.mark.parametrize("rows,N,k", [(5, 7, 1), (5, 7, 7), (33, 64, 10), (12, 197, 100), (9, 300, 150), (3, 1500, 20), (2, 4096, 4000)])
def test_topk_mask_matches_torch_topk(rows, N, k):
    """0 at the k largest entries of every row, -1e30 elsewhere (kvt.py:85-88 as an additive bias)."""
    torch.manual_seed(rows * N + k)
    x = torch.randn(rows, N) * 3
    x[0, : min(N, 4)] = torch.tensor([-0.0, 0.0, -1e-30, 1e-30])[: min(N, 4)]           # sign / denormal ordering
    want = torch.full_like(x, -1e30)
    want.scatter_(-1, torch.topk(x, k, dim=-1)[1], 0.0)
    got = F().topk_mask_(x.clone().cuda(), k).cpu()
    if not torch.equal(got, want):                                     # -0.0 == 0.0 ties may resolve either way in torch.topk
        assert int((got == 0).sum(-1).min()) >= k and torch.equal(got[1:], want[1:])


@pytest.mark.parametrize("B,H,W,C,oh,ow", [(2, 56, 56, 64, 5, 4), (3, 11, 15, 80, 6, 8), (1, 7, 7, 4, 7, 7), (2, 9, 5, 12, 1, 1), (2, 5, 9, 8, 2, 3)])
def test_adaptive_pool_and_dwconv_residual_on_tokens(B, H, W, C, oh, ow):
    torch.manual_seed(H * W + C)
    x = torch.randn(B, H * W, C)
    conv = torch.nn.Conv2d(C, C, 3, 1, 1, groups=C).eval()
    grid = x.double().permute(0, 2, 1).reshape(B, C, H, W)
    pool = torch.nn.functional.adaptive_avg_pool2d(grid, (oh, ow))
    ref = pool + torch.nn.functional.conv2d(pool, conv.weight.double(), conv.bias.double(), padding=1, groups=C)
    ref = ref.reshape(B, C, -1).permute(0, 2, 1).float()
    got = F().pooled_pyramid_tokens(x.cuda(), H, W, [(oh, ow), (oh, ow)], [conv.cuda(), conv.cuda()]).cpu()
    assert got.shape == (B, 2 * oh * ow, C)
    assert_parity(got[:, : oh * ow], ref, 1e-5, "pyramid level 0")
    assert torch.equal(got[:, : oh * ow], got[:, oh * ow:])


@pytest.mark.parametrize("B,C,H,W,ks", [(2, 64, 28, 28, 3), (2, 96, 9, 13, 5), (1, 5, 4, 3, 3), (3, 33, 7, 40, 7)])
def test_dwconv_bn_nchw_to_tokens_and_back(B, C, H, W, ks):
    torch.manual_seed(C * H + W)
    x = torch.randn(B, C, H, W)
    conv = torch.nn.Conv2d(C, C, ks, 1, (ks - 1) // 2, groups=C)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 1.5); bn.weight.normal_(1, 0.2); bn.bias.normal_(0, 0.2)
    conv.eval(); bn.eval()
    with torch.no_grad():
        ref = bn.double()(conv.double()(x.double())).float()
    conv.float(); bn.float()
    f = F()
    tok = f.dwconv_bn_nchw_tokens(x.cuda(), conv.weight.cuda(), conv.bias.cuda(), bn.cuda())
    assert tok.shape == (B, H * W, C)
    assert_parity(tok.cpu(), ref.reshape(B, C, -1).permute(0, 2, 1), 1e-5, "dwconv + bn -> tokens")
    back = f.tokens_to_nchw(tok, H, W)
    assert torch.equal(back.cpu(), tok.cpu().permute(0, 2, 1).reshape(B, C, H, W))
    alpha = torch.tensor([0.37]).cuda()
    assert_parity(f.tokens_to_nchw_axpy(tok, x.cuda(), alpha).cpu(), 0.37 * back.cpu() + x, 1e-6, "alpha * tokens^T + x")


@pytest.mark.parametrize("B,h,Nq,Nkv,d", [(2, 4, 197, 197, 64), (3, 2, 50, 13, 32), (1, 3, 7, 9, 24)])
def test_qk_logits(B, h, Nq, Nkv, d):
    torch.manual_seed(Nq + Nkv)
    C = h * d
    qkv = torch.randn(B, Nq, 3 * C)
    kk = torch.randn(B, Nkv, 2 * C)
    got = F().qk_logits(qkv.cuda()[..., :C], kk.cuda()[..., C:], h).cpu()
    q = qkv[..., :C].double().reshape(B, Nq, h, d).permute(0, 2, 1, 3)
    k = kk[..., C:].double().reshape(B, Nkv, h, d).permute(0, 2, 1, 3)
    assert_parity(got, (q @ k.transpose(-1, -2)).float(), 5e-5, "unscaled logits (split-bf16)")


def test_axis_gates_reject_training_mode_and_bad_shapes():
    from mi355attn.modules import BAM, CoordinateAttention, TripletAttention
    x = torch.randn(2, 32, 8, 8).cuda()
    for m in (BAM(32), TripletAttention(), CoordinateAttention(32, 32)):
        with pytest.raises(RuntimeError):
            m.cuda().train()(x)
    with pytest.raises(ValueError):
        CoordinateAttention(32, 16).cuda().eval()(x)


def test_stem_direct_option_switches_between_two_agreeing_paths():
    import mi355attn
    torch.manual_seed(5)
    x = torch.randn(2, 3, 40, 36).cuda()
    w = torch.nn.Parameter((torch.randn(16, 3, 3, 3) / 5).cuda())
    b, pos = torch.randn(16).cuda(), torch.randn(20 * 18, 16).cuda()
    f = F()
    assert mi355attn.get_option("stem_direct") == 1
    direct, _ = f.conv2d_tokens(x, w, b, 3, 2, 1, 0, precision=0, act=f.ACT_GELU, pos=pos)
    mi355attn.set_option("stem_direct", 0)
    try:
        gemm, _ = f.conv2d_tokens(x, w, b, 3, 2, 1, 0, precision=0, act=f.ACT_GELU, pos=pos)
    finally:
        mi355attn.set_option("stem_direct", 1)
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), stride=2, padding=1).flatten(2).transpose(1, 2)
    ref = gelu64(ref + pos.double().cpu())
    assert_parity(direct.cpu(), ref.float(), 5e-6, "direct stem conv")
    assert_parity(gemm.cpu(), ref.float(), TOL[0], "implicit-GEMM stem conv")


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("M,N,K,out16,act", [(3136, 192, 64, True, 0), (5000, 384, 128, True, 0), (130, 64, 64, False, 1), (2049, 264, 128, False, 0),
                                             (1, 8, 64, True, 1)])
def test_ln_linear16(M, N, K, out16, act, prec):
    """LayerNorm applied on the way into the GEMM == layernorm16 followed by linear16 up to the rounding of the folded weights."""
    torch.manual_seed(M + N + K)
    f = F()
    x = (torch.randn(M, K) * 1.7 + 0.3).cuda()
    ln = torch.nn.LayerNorm(K).cuda()
    lin = torch.nn.Linear(K, N).cuda()
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.2); ln.bias.normal_(0, 0.2)
    got = f.ln_linear16(x, ln, lin, act=f.ACT_GELU if act else f.ACT_NONE, out16=out16, precision=prec)
    z = torch.nn.functional.layer_norm(x.double().cpu(), (K,), ln.weight.double().cpu(), ln.bias.double().cpu(), ln.eps)
    z = z @ lin.weight.double().cpu().t() + lin.bias.double().cpu()
    ref = gelu64(z) if act else z
    assert got.dtype == (f.dtype16(prec) if out16 else torch.float32)
    assert_parity(got.float().cpu(), ref.float(), 1.5e-3 if prec == 1 else 1.5e-2, f"ln_linear16{(M, N, K)} p{prec}")


def test_ln_linear16_rejects_other_widths():
    from mi355attn import Mi355Error
    f = F()
    with pytest.raises(Mi355Error):
        f.ln_linear16(torch.randn(16, 96).cuda(), torch.nn.LayerNorm(96).cuda(), torch.nn.Linear(96, 64).cuda(), precision=1)
