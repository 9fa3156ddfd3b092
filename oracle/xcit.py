"""Oracle (test infrastructure): XCiT cross-covariance attention (XCA), LPI, XCABlock, the convolutional patch embedding, the
Fourier position encoding, the class-attention stage and the full XCiT forward (eval mode)."""
import math

import torch
import torch.nn.functional as TF
from .transformer import layernorm, gelu, linear, _t, _sub


def xca_forward(x, p, num_heads, dtype=torch.float32):
    """XCA.forward -- vision_transformers/xcit.py:245-265.

    Per head: q,k,v as (d x N); q,k L2-normalised along N (F.normalize, eps 1e-12 on the norm);
    A = softmax_rows((q k^T) * temperature_h) (d x d); O = A v (d x N); y[b,n,i*d+j] = O_i[j,n]; proj.
    """
    x = _t(x, dtype)
    B, N, C = x.shape
    d = C // num_heads
    qkv = linear(x, _t(p["qkv.weight"], dtype), _t(p["qkv.bias"], dtype) if "qkv.bias" in p else None)
    temp = _t(p["temperature"], dtype).reshape(num_heads)
    out = torch.empty(B, N, C, dtype=dtype)
    for i in range(num_heads):
        q = qkv[:, :, 0 * C + i * d: 0 * C + (i + 1) * d].transpose(1, 2)      # (B,d,N)
        k = qkv[:, :, 1 * C + i * d: 1 * C + (i + 1) * d].transpose(1, 2)
        v = qkv[:, :, 2 * C + i * d: 2 * C + (i + 1) * d].transpose(1, 2)
        qn = q / torch.clamp_min(torch.sqrt((q * q).sum(dim=2, keepdim=True)), 1e-12)
        kn = k / torch.clamp_min(torch.sqrt((k * k).sum(dim=2, keepdim=True)), 1e-12)
        a = (qn @ kn.transpose(1, 2)) * temp[i]                                # (B,d,d)
        a = torch.softmax(a, dim=2)
        out[:, :, i * d:(i + 1) * d] = (a @ v).transpose(1, 2)
    return linear(out, _t(p["proj.weight"], dtype), _t(p["proj.bias"], dtype))


def _dw3x3(img, w, b):
    """Depth-wise 3x3 cross-correlation, zero pad 1.  img (B,C,H,W), w (C,1,3,3), b (C,)."""
    B, C, H, W = img.shape
    halo = torch.zeros(B, C, H + 2, W + 2, dtype=img.dtype)
    halo[:, :, 1:1 + H, 1:1 + W] = img
    acc = torch.zeros_like(img) + b[None, :, None, None]
    for dy in range(3):
        for dx in range(3):
            acc = acc + halo[:, :, dy:dy + H, dx:dx + W] * w[None, :, 0, dy, dx, None, None]
    return acc


def lpi_forward(x, p, H, W, dtype=torch.float32, bn_eps=1e-5):
    """LPI.forward -- vision_transformers/xcit.py:149-157, BatchNorm2d in eval mode (running stats).

    tokens (B,N,C) -> image (B,C,H,W) -> dw3x3 -> GELU -> BN(eval) -> dw3x3 -> tokens.
    """
    x = _t(x, dtype)
    B, N, C = x.shape
    img = x.transpose(1, 2).reshape(B, C, H, W)
    y = gelu(_dw3x3(img, _t(p["conv1.weight"], dtype), _t(p["conv1.bias"], dtype)))
    mean, var = _t(p["bn.running_mean"], dtype), _t(p["bn.running_var"], dtype)
    g, bt = _t(p["bn.weight"], dtype), _t(p["bn.bias"], dtype)
    y = (y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + bn_eps) \
        * g[None, :, None, None] + bt[None, :, None, None]
    y = _dw3x3(y, _t(p["conv2.weight"], dtype), _t(p["conv2.bias"], dtype))
    return y.reshape(B, C, N).transpose(1, 2)


def xca_block_forward(x, p, num_heads, H, W, dtype=torch.float32):
    """XCABlock.forward -- vision_transformers/xcit.py:290-294.

    x += gamma1 * XCA(LN1 x); x += gamma3 * LPI(LN3 x); x += gamma2 * Mlp(LN2 x)   (Mlp :32-38, one GELU).
    """
    x = _t(x, dtype)
    u = layernorm(x, _t(p["norm1.weight"], dtype), _t(p["norm1.bias"], dtype))
    x = x + _t(p["gamma1"], dtype) * xca_forward(u, _sub(p, "attn."), num_heads, dtype)
    u = layernorm(x, _t(p["norm3.weight"], dtype), _t(p["norm3.bias"], dtype))
    x = x + _t(p["gamma3"], dtype) * lpi_forward(u, _sub(p, "local_mp."), H, W, dtype)
    u = layernorm(x, _t(p["norm2.weight"], dtype), _t(p["norm2.bias"], dtype))
    m = _sub(p, "mlp.")
    h = gelu(linear(u, _t(m["fc1.weight"], dtype), _t(m["fc1.bias"], dtype)))
    return x + _t(p["gamma2"], dtype) * linear(h, _t(m["fc2.weight"], dtype), _t(m["fc2.bias"], dtype))


def fourier_position_rows(H, W, w, b, hidden_dim=32, temperature=10000, dtype=torch.float32):
    """PositionalEncodingFourier.forward -- vision_transformers/xcit.py:55-77, for one image, as token rows (H*W, dim).

    y/x_embed = cumulative index (1-based) / (last + 1e-6) * 2*pi; channel i of each half is embed / T^(2*(i//2)/hidden) with sin on
    even and cos on odd channels; the halves are ordered [y, x]; token_projection is a 1x1 conv = a matmul over the 2*hidden features.
    """
    ys = torch.arange(1, H + 1, dtype=torch.float32)[:, None].expand(H, W)
    xs = torch.arange(1, W + 1, dtype=torch.float32)[None, :].expand(H, W)
    ys = ys / (float(H) + 1e-6) * (2 * math.pi)
    xs = xs / (float(W) + 1e-6) * (2 * math.pi)
    i = torch.arange(hidden_dim, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / hidden_dim)
    feats = []
    for emb in (ys, xs):
        ang = emb[:, :, None] / dim_t                                  # (H, W, hidden)
        f = torch.empty(H, W, hidden_dim, dtype=torch.float32)
        f[:, :, 0::2] = ang[:, :, 0::2].sin()
        f[:, :, 1::2] = ang[:, :, 1::2].cos()
        feats.append(f)
    feat = torch.cat(feats, dim=2).reshape(H * W, 2 * hidden_dim).to(dtype)
    w = _t(w, dtype).reshape(w.shape[0], -1)
    return feat @ w.t() + _t(b, dtype)


def conv_patch_embed_forward(img, p, dtype=torch.float32, eps=1e-5):
    """ConvPatchEmbed.forward -- xcit.py:119-125 (ctor :91-117): [conv3x3 s2 p1 (no bias) -> BatchNorm2d(eval) -> GELU] x (n-1), then
    conv + BN; tokens row-major over the final grid.  BatchNorm in eval mode: (z - running_mean) / sqrt(running_var + eps) * w + b."""
    x = _t(img, dtype)
    stages = sorted({int(k.split(".")[1]) for k in p if k.startswith("proj.")})
    for n, si in enumerate(stages):
        q = _sub(p, f"proj.{si}.")
        x = TF.conv2d(x, _t(q["0.weight"], dtype), None, stride=2, padding=1)
        mean, var = _t(q["1.running_mean"], dtype), _t(q["1.running_var"], dtype)
        x = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
        x = x * _t(q["1.weight"], dtype)[None, :, None, None] + _t(q["1.bias"], dtype)[None, :, None, None]
        if n + 1 < len(stages):
            x = gelu(x)
    Hp, Wp = x.shape[2], x.shape[3]
    return x.flatten(2).transpose(1, 2), (Hp, Wp)


def class_attention_block_forward(x, p, num_heads, dtype=torch.float32, tokens_norm=False):
    """ClassAttentionBlock.forward -- xcit.py:218-231 with ClassAttention.forward :174-188.  tokens_norm=True (xcit.py:221-222, the
    XCiT-S/M/L configurations): norm2 is applied to EVERY token after the first residual, not to the cls token only.

    The attention returns cat(proj(cls attention), NORMED patch tokens) (:187), so the first residual gives patch tokens
    x + gamma1*LN1(x); norm2 is applied to the cls token only (:223); the second residual adds x_res to cat(gamma2*mlp(cls), x[1:])
    (:226-230), i.e. the patch tokens are doubled.
    """
    x = _t(x, dtype)
    B, N, C = x.shape
    d = C // num_heads
    u = layernorm(x, _t(p["norm1.weight"], dtype), _t(p["norm1.bias"], dtype))
    qkv = linear(u, _t(p["attn.qkv.weight"], dtype), _t(p["attn.qkv.bias"], dtype) if "attn.qkv.bias" in p else None)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    cls = torch.empty(B, C, dtype=dtype)
    for i in range(num_heads):
        sl = slice(i * d, (i + 1) * d)
        s = (q[:, 0:1, sl] * k[:, :, sl]).sum(dim=-1) * (d ** -0.5)                 # (B, N)
        s = s - s.amax(dim=-1, keepdim=True)
        e = torch.exp(s)
        a = e / e.sum(dim=-1, keepdim=True)
        cls[:, sl] = torch.einsum("bn,bnd->bd", a, v[:, :, sl])
    cls = linear(cls, _t(p["attn.proj.weight"], dtype), _t(p["attn.proj.bias"], dtype))
    att = torch.cat([cls[:, None, :], u[:, 1:]], dim=1)
    x = x + _t(p["gamma1"], dtype) * att
    if tokens_norm:
        x = layernorm(x, _t(p["norm2.weight"], dtype), _t(p["norm2.bias"], dtype))
    else:
        c = layernorm(x[:, 0:1], _t(p["norm2.weight"], dtype), _t(p["norm2.bias"], dtype))
        x = torch.cat([c, x[:, 1:]], dim=1)
    m = _sub(p, "mlp.")
    h = gelu(linear(x[:, 0:1], _t(m["fc1.weight"], dtype), _t(m["fc1.bias"], dtype)))
    c = _t(p["gamma2"], dtype) * linear(h, _t(m["fc2.weight"], dtype), _t(m["fc2.bias"], dtype))
    return x + torch.cat([c, x[:, 1:]], dim=1)


def xcit_forward(img, p, num_heads=4, depth=12, cls_layers=2, dtype=torch.float32, tokens_norm=False):
    """XCiT.forward -- xcit.py:392-414 (xcit_nano_12_p16 :416-420): ConvPatchEmbed, + Fourier position rows, `depth` XCABlocks,
    cls token prepended, `cls_layers` ClassAttentionBlocks, LayerNorm, cls row, head."""
    x, (Hp, Wp) = conv_patch_embed_forward(img, _sub(p, "patch_embed."), dtype)
    x = x + fourier_position_rows(Hp, Wp, p["pos_embeder.token_projection.weight"], p["pos_embeder.token_projection.bias"], dtype=dtype)
    for i in range(depth):
        x = xca_block_forward(x, _sub(p, f"blocks.{i}."), num_heads, Hp, Wp, dtype)
    B = x.shape[0]
    x = torch.cat([_t(p["cls_token"], dtype).expand(B, -1, -1), x], dim=1)
    for i in range(cls_layers):
        x = class_attention_block_forward(x, _sub(p, f"cls_attn_blocks.{i}."), num_heads, dtype, tokens_norm)
    x = layernorm(x, _t(p["norm.weight"], dtype), _t(p["norm.bias"], dtype))[:, 0]
    return linear(x, _t(p["head.weight"], dtype), _t(p["head.bias"], dtype))
