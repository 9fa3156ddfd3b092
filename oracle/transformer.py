"""Oracle (test infrastructure): ViT attention / encoder / full model and the MLP-Mixer layer.

Parameters are passed as a dict ``p`` whose keys are the reference's state_dict names relative to the
module (e.g. ``qkv.weight``).  Missing optional keys (``qkv.bias``) mean "no bias", as in the reference.
"""
import math
import torch


def _t(v, dtype):
    return v.detach().to("cpu", dtype)


def layernorm(x, weight, bias, eps=1e-5):
    """nn.LayerNorm over the last axis (biased variance, eps inside the sqrt), as used at ViT.py:111-114."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * weight + bias


def gelu(x):
    """nn.GELU() default = exact erf form (ViT.py:53, cswin.py:35 act_layer=nn.GELU)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def sdpa_core(q, k, v, scale, pre_scale=False):
    """softmax(q k^T * scale) v over (..., N, d) tensors.

    ``pre_scale=False``: scale multiplies the product (ViT.py:83).  ``pre_scale=True``: q is scaled
    before the product (cswin.py:116-117).  Kept separate because fp32 rounding differs.
    """
    if pre_scale:
        s = (q * scale) @ k.transpose(-1, -2)
    else:
        s = (q @ k.transpose(-1, -2)) * scale
    s = s - s.amax(dim=-1, keepdim=True)
    e = torch.exp(s)
    return (e / e.sum(dim=-1, keepdim=True)) @ v


def vit_attention_forward(x, p, num_heads, dtype=torch.float32):
    """Attention.forward -- vision_transformers/ViT.py:79-89 (ctor :68-77).

    Channel index of q/k/v for head i lane j in the qkv GEMM output is s*C + i*d + j (s=0,1,2).
    """
    x = _t(x, dtype)
    B, N, C = x.shape
    d = C // num_heads
    qkv = linear(x, _t(p["qkv.weight"], dtype), _t(p["qkv.bias"], dtype) if "qkv.bias" in p else None)
    out = torch.empty(B, N, C, dtype=dtype)
    scale = d ** -0.5
    for i in range(num_heads):
        q = qkv[:, :, 0 * C + i * d: 0 * C + (i + 1) * d]
        k = qkv[:, :, 1 * C + i * d: 1 * C + (i + 1) * d]
        v = qkv[:, :, 2 * C + i * d: 2 * C + (i + 1) * d]
        out[:, :, i * d:(i + 1) * d] = sdpa_core(q, k, v, scale)
    return linear(out, _t(p["proj.weight"], dtype), _t(p["proj.bias"], dtype))


def vit_mlp_forward(x, p, dtype=torch.float32):
    """Mlp.forward -- vision_transformers/ViT.py:58-65: GELU after fc1 AND after fc2 (reference quirk)."""
    x = _t(x, dtype)
    h = gelu(linear(x, _t(p["fc1.weight"], dtype), _t(p["fc1.bias"], dtype)))
    return gelu(linear(h, _t(p["fc2.weight"], dtype), _t(p["fc2.bias"], dtype)))


def _sub(p, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix)}


def vit_encoder_forward(x, p, num_heads, dtype=torch.float32):
    """TransformerEncoder.forward -- vision_transformers/ViT.py:116-119 (pre-LN residual block)."""
    x = _t(x, dtype)
    u = layernorm(x, _t(p["layernorm1.weight"], dtype), _t(p["layernorm1.bias"], dtype))
    x = x + vit_attention_forward(u, _sub(p, "attn."), num_heads, dtype)
    u = layernorm(x, _t(p["layernorm2.weight"], dtype), _t(p["layernorm2.bias"], dtype))
    return x + vit_mlp_forward(u, _sub(p, "mlp."), dtype)


def vit_patch_embed_forward(img, w, b, dtype=torch.float32):
    """PatchEmbedding.forward -- vision_transformers/ViT.py:101-105.

    Conv2d(k=P,s=P) == per-patch GEMM with K = Cin*P*P in (c, ky, kx) order; tokens row-major over the
    patch grid.
    """
    img, w, b = _t(img, dtype), _t(w, dtype), _t(b, dtype)
    B, Cin, H, W = img.shape
    E, _, P, _ = w.shape
    gh, gw = H // P, W // P
    patches = img.reshape(B, Cin, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, Cin * P * P)
    return patches @ w.reshape(E, Cin * P * P).t() + b


def _cubic_coeffs(t):
    """Cubic-convolution weights (A = -0.75) of the four taps around a sample at fractional offset t in [0, 1)."""
    A = -0.75
    x0, x1, x2, x3 = t + 1.0, t, 1.0 - t, 2.0 - t
    return (((A * x0 - 5.0 * A) * x0 + 8.0 * A) * x0 - 4.0 * A, ((A + 2.0) * x1 - (A + 3.0)) * x1 * x1 + 1.0,
            ((A + 2.0) * x2 - (A + 3.0)) * x2 * x2 + 1.0, ((A * x3 - 5.0 * A) * x3 + 8.0 * A) * x3 - 4.0 * A)


def bicubic_rows(table, n0h, n0w, oh, ow, scale_h, scale_w, dtype=torch.float32):
    """Restatement of F.interpolate(mode="bicubic", align_corners=False, scale_factor=(scale_h, scale_w)) on a token-major table
    (n0h*n0w, dim) -> (oh*ow, dim): output (oy, ox) samples the source at (o + 0.5) / scale - 0.5 (coordinate NOT clamped for the
    cubic filter), four taps per axis, tap indices clamped to the grid.  The call site is ViT.py:171-175."""
    t = _t(table, dtype).reshape(n0h, n0w, -1)

    def axis(n_out, n_in, scale):
        r = (torch.arange(n_out, dtype=dtype) + 0.5) / scale - 0.5
        f = torch.floor(r)
        c = _cubic_coeffs(r - f)
        idx = [torch.clamp(f.long() - 1 + i, 0, n_in - 1) for i in range(4)]
        return idx, c

    iy, cy = axis(oh, n0h, scale_h)
    ix, cx = axis(ow, n0w, scale_w)
    out = torch.zeros(oh, ow, t.shape[-1], dtype=dtype)
    for i in range(4):
        rows = t[iy[i]]                                            # (oh, n0w, dim)
        acc = torch.zeros(oh, ow, t.shape[-1], dtype=dtype)
        for j in range(4):
            acc = acc + cx[j][None, :, None] * rows[:, ix[j]]
        out = out + cy[i][:, None, None] * acc
    return out.reshape(oh * ow, -1)


def vit_position_rows(pe, patch, H, W, dtype=torch.float32):
    """interpolate_pos_encoding -- ViT.py:160-178: the parameter at the native resolution; otherwise row 0 (`class_pos_embed`, :165)
    followed by the other rows resized from their (n0, n0) grid to (W // patch, H // patch) with scale factors
    ((W // patch + 0.1) / n0, (H // patch + 0.1) / n0) (:169-175: note that the WIDTH count scales the first grid axis)."""
    import math
    pe = _t(pe, dtype)
    N = pe.shape[1] - 1
    if (H // patch) * (W // patch) == N and W == H:
        return pe
    n0 = int(math.sqrt(N))
    w0, h0 = W // patch, H // patch
    body = bicubic_rows(pe[0, 1:], n0, n0, w0, h0, (w0 + 0.1) / math.sqrt(N), (h0 + 0.1) / math.sqrt(N), dtype)
    return torch.cat([pe[:, 0:1], body[None]], dim=1)


def vit_forward(img, p, num_heads, depth, dtype=torch.float32):
    """VisionTransformer.forward -- vision_transformers/ViT.py:180-192.

    Tokens = [patch_0..patch_{n-1}, cls] (cls appended LAST, :183), + position rows (interpolated off the native resolution,
    :160-178), `depth` encoder blocks, logits = head(token 0) (global_pool="token", :187-188); no final LayerNorm.
    """
    w = p["patch_embedding.proj.weight"]
    x = vit_patch_embed_forward(img, w, p["patch_embedding.proj.bias"], dtype)
    B = x.shape[0]
    cls = _t(p["cls_token"], dtype).expand(B, -1, -1)
    x = torch.cat([x, cls], dim=1) + vit_position_rows(p["position_embedding"], w.shape[-1], img.shape[-2], img.shape[-1], dtype)
    for i in range(depth):
        x = vit_encoder_forward(x, _sub(p, f"blocks.{i}."), num_heads, dtype)
    return linear(x[:, 0], _t(p["head.weight"], dtype), _t(p["head.bias"], dtype))


def mixer_layer_forward(x, p, dtype=torch.float32):
    """MixerLayer.forward -- mlps/mlp_mixer.py:45-50 (Mlp :26-33, single GELU).

    Token mixing is a LEFT multiplication of the (N x C) matrix: x_b += W2 gelu(W1 LN1(x_b) + b1) + b2.
    """
    x = _t(x, dtype)
    u = layernorm(x, _t(p["norm1.weight"], dtype), _t(p["norm1.bias"], dtype))
    w1, b1 = _t(p["token_mlp.fc1.weight"], dtype), _t(p["token_mlp.fc1.bias"], dtype)
    w2, b2 = _t(p["token_mlp.fc2.weight"], dtype), _t(p["token_mlp.fc2.bias"], dtype)
    hid = gelu(torch.einsum("tn,bnc->btc", w1, u) + b1[None, :, None])
    x = x + torch.einsum("nt,btc->bnc", w2, hid) + b2[None, :, None]
    u = layernorm(x, _t(p["norm2.weight"], dtype), _t(p["norm2.bias"], dtype))
    h = gelu(linear(u, _t(p["channel_mlp.fc1.weight"], dtype), _t(p["channel_mlp.fc1.bias"], dtype)))
    return x + linear(h, _t(p["channel_mlp.fc2.weight"], dtype), _t(p["channel_mlp.fc2.bias"], dtype))


def mixer_forward(img, p, depth=12, dtype=torch.float32):
    """MLP_Mixer.forward -- mlps/mlp_mixer.py:74-79: patch embedding (Conv k=s=patch == per-patch GEMM), `depth` MixerLayers,
    mean over tokens, head."""
    x = vit_patch_embed_forward(img, p["patch_embedding.proj.weight"], p["patch_embedding.proj.bias"], dtype)
    for i in range(depth):
        x = mixer_layer_forward(x, _sub(p, f"blocks.{i}."), dtype)
    return linear(x.mean(dim=1), _t(p["head.weight"], dtype), _t(p["head.bias"], dtype))


def mhsa_forward(x, p, num_heads, H=None, W=None, sr_ratio=1, relative_pos=None, layout="qkv", dtype=torch.float32, bn_eps=1e-5):
    """The plain multi-head attention of the reference's other ViT files.

    layout "qkv"    : fused qkv Linear                                   -- setr.py:62-72, moat.py:74-84
    layout "q,k,v"  : separate Linears; K/V source reduced by a depth-wise conv (k = s = sr_ratio, bias) + BatchNorm2d(eval)
                      when sr_ratio > 1; optional additive relative_pos    -- pvt.py:73-91, cmt.py:93-111
    layout "q,kv"   : q Linear + fused kv Linear; K/V source reduced by a dense conv (k = s = sr_ratio, bias)   -- segformer.py:33-50
    """
    import torch.nn.functional as TF
    x = _t(x, dtype)
    B, N, C = x.shape
    d = C // num_heads

    def opt(k):
        return _t(p[k], dtype) if k in p else None

    src = x
    if sr_ratio > 1:
        grid = x.transpose(1, 2).reshape(B, C, H, W)
        if layout == "q,k,v":
            grid = TF.conv2d(grid, _t(p["sr.0.weight"], dtype), opt("sr.0.bias"), stride=sr_ratio, groups=C)
            mean, var = _t(p["sr.1.running_mean"], dtype), _t(p["sr.1.running_var"], dtype)
            grid = (grid - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + bn_eps)
            grid = grid * _t(p["sr.1.weight"], dtype)[None, :, None, None] + _t(p["sr.1.bias"], dtype)[None, :, None, None]
        else:
            grid = TF.conv2d(grid, _t(p["sr.weight"], dtype), opt("sr.bias"), stride=sr_ratio)
        src = grid.reshape(B, C, -1).transpose(1, 2)
    if layout == "qkv":
        qkv = linear(x, _t(p["qkv.weight"], dtype), opt("qkv.bias"))
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    elif layout == "q,k,v":
        q = linear(x, _t(p["q.weight"], dtype), opt("q.bias"))
        k = linear(src, _t(p["k.weight"], dtype), opt("k.bias"))
        v = linear(src, _t(p["v.weight"], dtype), opt("v.bias"))
    else:
        q = linear(x, _t(p["q.weight"], dtype), opt("q.bias"))
        kv = linear(src, _t(p["kv.weight"], dtype), opt("kv.bias"))
        k, v = kv[..., :C], kv[..., C:]
    out = torch.empty(B, N, C, dtype=dtype)
    for i in range(num_heads):
        sl = slice(i * d, (i + 1) * d)
        s = (q[..., sl] @ k[..., sl].transpose(-1, -2)) * d ** -0.5
        if relative_pos is not None:
            s = s + _t(relative_pos, dtype)[i]
        s = s - s.amax(dim=-1, keepdim=True)
        e = torch.exp(s)
        out[..., sl] = (e / e.sum(dim=-1, keepdim=True)) @ v[..., sl]
    return linear(out, _t(p["proj.weight"], dtype), _t(p["proj.bias"], dtype))


# ---- the remaining copies of the multi-head pattern: dilateformer, bvit, efficientformer, kvt, cvt, p2t ---------------
def _heads_attention(q, k, v, scale, topk=None):
    """softmax(q k^T * scale) v per head; q (B,h,Nq,dq), k (B,h,Nk,dq), v (B,h,Nk,dv).  topk: keep the k largest logits of every row
    (the rest -> -inf) as kvt.py:84-88 does with topk + scatter + where."""
    attn = (q @ k.transpose(-1, -2)) * scale
    if topk is not None:
        index = torch.topk(attn, k=topk, dim=-1)[1]
        mask = torch.zeros_like(attn).scatter_(-1, index, 1.0)
        attn = torch.where(mask > 0, attn, torch.full_like(attn, float("-inf")))
    return torch.softmax(attn, dim=-1) @ v


def global_attention_forward(x, p, num_heads, dtype=torch.float32):
    """GlobalAttention.forward -- vision_transformers/dilateformer.py:152-164: plain MHSA on a channels-last (B,H,W,C) grid."""
    x = _t(x, dtype)
    B, H, W, C = x.shape
    d = C // num_heads
    qkv = linear(x.reshape(B, H * W, C), _t(p["qkv.weight"], dtype), _t(p["qkv.bias"], dtype) if "qkv.bias" in p else None)
    qkv = qkv.reshape(B, H * W, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    o = _heads_attention(qkv[0], qkv[1], qkv[2], d ** -0.5).transpose(1, 2).reshape(B, H, W, C)
    return linear(o, _t(p["proj.weight"], dtype), _t(p["proj.bias"], dtype))


def broad_attention_forward(x, p, heads, dim_head, dtype=torch.float32):
    """Broad_Attention.forward -- vision_transformers/bvit.py:66-76: returns (to_out(out), q, k, v) with q, k, v as (b, h, n, d)."""
    x = _t(x, dtype)
    B, N, _ = x.shape
    qkv = linear(x, _t(p["to_qkv.weight"], dtype), None).chunk(3, dim=-1)
    q, k, v = [t.reshape(B, N, heads, dim_head).permute(0, 2, 1, 3) for t in qkv]
    o = _heads_attention(q, k, v, dim_head ** -0.5).permute(0, 2, 1, 3).reshape(B, N, heads * dim_head)
    if "to_out.0.weight" in p:
        o = linear(o, _t(p["to_out.0.weight"], dtype), _t(p["to_out.0.bias"], dtype))
    return o, q, k, v


def qk_v_attention_forward(x, p, query_dim, num_heads, dtype=torch.float32):
    """Attention.forward -- vision_transformers/efficientformer.py:70-81: q / k of width query_dim / heads, v of width dim / heads."""
    x = _t(x, dtype)
    B, N, C = x.shape
    opt = lambda k: _t(p[k], dtype) if k in p else None
    qk = linear(x, _t(p["qk.weight"], dtype), opt("qk.bias")).reshape(B, N, 2, num_heads, query_dim // num_heads).permute(2, 0, 3, 1, 4)
    v = linear(x, _t(p["v.weight"], dtype), opt("v.bias")).reshape(B, N, num_heads, C // num_heads).permute(0, 2, 1, 3)
    o = _heads_attention(qk[0], qk[1], v, (query_dim // num_heads) ** -0.5).transpose(1, 2).reshape(B, N, C)
    return linear(o, _t(p["proj.weight"], dtype), _t(p["proj.bias"], dtype))


def knn_attention_forward(x, p, num_heads, topk, dtype=torch.float32):
    """KNNAttention.forward -- vision_transformers/kvt.py:80-94: every query keeps its `topk` largest scaled logits."""
    x = _t(x, dtype)
    B, N, C = x.shape
    d = C // num_heads
    qkv = linear(x, _t(p["qkv.weight"], dtype), _t(p["qkv.bias"], dtype) if "qkv.bias" in p else None)
    qkv = qkv.reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    o = _heads_attention(qkv[0], qkv[1], qkv[2], d ** -0.5, topk=topk).transpose(1, 2).reshape(B, N, C)
    return linear(o, _t(p["proj.weight"], dtype), _t(p["proj.bias"], dtype))


def conv_attention_forward(x, p, num_heads, dtype=torch.float32, bn_eps=1e-5):
    """Attention.forward -- vision_transformers/cvt.py:64-76: qkv from depth-wise conv -> BatchNorm2d (eval) -> 1x1 conv on an NCHW map,
    attention over the H*W positions, 1x1 conv projection, NCHW result."""
    import torch.nn.functional as TF
    x = _t(x, dtype)
    B, C, H, W = x.shape
    d = C // num_heads
    g = lambda k: _t(p[k], dtype)
    ks = p["conv_proj_qkv.0.weight"].shape[-1]
    y = TF.conv2d(x, g("conv_proj_qkv.0.weight"), g("conv_proj_qkv.0.bias"), padding=(ks - 1) // 2, groups=C)
    y = (y - g("conv_proj_qkv.1.running_mean")[None, :, None, None]) / torch.sqrt(g("conv_proj_qkv.1.running_var")[None, :, None, None] + bn_eps)
    y = y * g("conv_proj_qkv.1.weight")[None, :, None, None] + g("conv_proj_qkv.1.bias")[None, :, None, None]
    qkv = TF.conv2d(y, g("conv_proj_qkv.2.weight"), g("conv_proj_qkv.2.bias")).reshape(B, 3, num_heads, d, H * W).permute(1, 0, 2, 4, 3)
    o = _heads_attention(qkv[0], qkv[1], qkv[2], d ** -0.5)                 # (B, h, N, d)
    o = o.transpose(-1, -2).reshape(B, C, H, W)
    return TF.conv2d(o, g("proj.weight"), g("proj.bias"))


def pooling_attention_forward(x, p, H, W, d_convs, num_heads, pool_ratios, dtype=torch.float32):
    """PoolingAttention.forward -- vision_transformers/p2t.py:71-95.  `d_convs`: the depth-wise 3x3 Conv2d modules the enclosing block
    passes in (p2t.py:127-128), one per pool ratio."""
    import torch.nn.functional as TF
    x = _t(x, dtype)
    B, N, C = x.shape
    d = C // num_heads
    opt = lambda k: _t(p[k], dtype) if k in p else None
    q = linear(x, _t(p["q.0.weight"], dtype), opt("q.0.bias")).reshape(B, N, num_heads, d).permute(0, 2, 1, 3)
    grid = x.permute(0, 2, 1).reshape(B, C, H, W)
    pools = []
    for ratio, conv in zip(pool_ratios, d_convs):
        pool = TF.adaptive_avg_pool2d(grid, (round(H / ratio), round(W / ratio)))
        pool = pool + TF.conv2d(pool, _t(conv.weight, dtype), _t(conv.bias, dtype), padding=1, groups=C)
        pools.append(pool.reshape(B, C, -1))
    pools = torch.cat(pools, dim=2).permute(0, 2, 1)
    pools = layernorm(pools, _t(p["norm.weight"], dtype), _t(p["norm.bias"], dtype))
    kv = linear(pools, _t(p["kv.0.weight"], dtype), opt("kv.0.bias")).reshape(B, -1, 2, num_heads, d).permute(2, 0, 3, 1, 4)
    o = _heads_attention(q, kv[0], kv[1], d ** -0.5).transpose(1, 2).reshape(B, N, C)
    return linear(o, _t(p["proj.weight"], dtype), _t(p["proj.bias"], dtype))
