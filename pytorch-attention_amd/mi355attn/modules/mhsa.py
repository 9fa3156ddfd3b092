"""Drop-in attention modules for the plain multi-head pattern the reference repeats outside ViT.py, on the streaming
attention core (csrc/sdpa_general.hip).  Module level only: the surrounding models of those files are not mirrored.

  Attention            setr.py:50-72, moat.py:62-84      fused qkv Linear -> softmax(QK^T s)V -> proj
  SRAttention          pvt.py:55-91                      separate q / k / v Linears; K, V from a token grid reduced by a depth-wise
                                                         conv (kernel == stride == sr_ratio) + BatchNorm2d
  SRAttentionRelPos    cmt.py:75-111                     the same with an additive relative-position term (heads, N, N')
  SRConvAttention      segformer.py:17-50                q Linear, fused kv Linear; K, V from a dense conv (kernel == stride == sr_ratio)
"""
import torch
from torch import nn

from .. import functional as F


def _no_dropout(*rates):
    """Dropout rates are accepted as the reference accepts them: in eval mode nn.Dropout is the identity, so a model built with its
    training configuration runs unchanged.  What the forward-only engine cannot do is the stochastic training-mode forward."""
    for r in rates:
        if not 0 <= float(r) <= 1:
            raise ValueError(f"dropout probability has to be between 0 and 1, but got {r}")


def _dropout_is_identity(m):
    if m.training and any(isinstance(c, nn.Dropout) and c.p > 0 for c in m.modules()):
        raise RuntimeError("inference engine: a non-zero dropout rate is only the identity in eval mode; call .eval()")


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0, proj_drop=0, precision=None):
        super().__init__()
        assert dim % num_heads == 0
        _no_dropout(attn_drop, proj_drop)
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.precision = precision

    def forward(self, x):
        _dropout_is_identity(self)
        C = x.shape[-1]
        qkv = F.linear(x, self.qkv.weight, self.qkv.bias, precision=self.precision)
        ctx = F.sdpa_general(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], self.num_heads, self.scale, precision=self.precision)
        return F.linear(ctx, self.proj.weight, self.proj.bias, precision=self.precision)


class SRAttention(nn.Module):
    """pvt.py:53-54 / cmt.py:73-74: `sr_ratio` is the THIRD positional parameter (both files call `Attention(dim, num_heads,
    sr_ratio, ...)` positionally, pvt.py:98, cmt.py:119); segformer.py:18 has it last (SRConvAttention below)."""

    def __init__(self, dim, num_heads=8, sr_ratio=1, qkv_bias=False, attn_drop=0, proj_drop=0, precision=None):
        super().__init__()
        assert dim % num_heads == 0
        _no_dropout(attn_drop, proj_drop)
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(dim, dim, bias=qkv_bias)
        self.v = nn.Linear(dim, dim, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.sr_ratio = sr_ratio
        if self.sr_ratio > 1:
            self.sr = nn.Sequential(nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio, groups=dim), nn.BatchNorm2d(dim))
        self.precision = precision

    def _kv_source(self, x, H, W):
        if self.sr_ratio <= 1:
            return x
        conv, bn = self.sr[0], self.sr[1]
        return F.dwconv_patch_tokens(x, conv.weight, conv.bias, bn, H, W, self.sr_ratio)

    def forward(self, x, H, W, relative_pos=None):
        _dropout_is_identity(self)
        p = self.precision
        q = F.linear(x, self.q.weight, self.q.bias, precision=p)
        src = self._kv_source(x, H, W)
        k = F.linear(src, self.k.weight, self.k.bias, precision=p)
        v = F.linear(src, self.v.weight, self.v.bias, precision=p)
        ctx = F.sdpa_general(q, k, v, self.num_heads, self.scale, bias=relative_pos, precision=p)
        return F.linear(ctx, self.proj.weight, self.proj.bias, precision=p)


class SRAttentionRelPos(SRAttention):
    """cmt.py:75-111: `relative_pos` (heads, N, N') is a required forward argument, added to the scaled logits."""

    def forward(self, x, H, W, relative_pos):
        _dropout_is_identity(self)
        return super().forward(x, H, W, relative_pos)


class SRConvAttention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0, proj_drop=0, sr_ratio=1, precision=None):
        super().__init__()
        assert dim % num_heads == 0
        _no_dropout(attn_drop, proj_drop)
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, 2 * dim, bias=qkv_bias)
        self.sr_ratio = sr_ratio
        if self.sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.precision = precision

    def forward(self, x, H, W):
        _dropout_is_identity(self)
        p = self.precision
        C = x.shape[-1]
        q = F.linear(x, self.q.weight, self.q.bias, precision=p)
        src = x
        if self.sr_ratio > 1:
            src, _ = F.conv2d_tokens(x, self.sr.weight, self.sr.bias, self.sr_ratio, self.sr_ratio, 0, 1, hw=(H, W), precision=p)
        kv = F.linear(src, self.kv.weight, self.kv.bias, precision=p)
        ctx = F.sdpa_general(q, kv[..., :C], kv[..., C:], self.num_heads, self.scale, precision=p)
        return F.linear(ctx, self.proj.weight, self.proj.bias, precision=p)


# ---- the remaining copies (dilateformer, bvit, efficientformer, kvt, cvt) -------------------------------------------------------------
def _fused_qkv_attention(x, qkv, proj, heads, d, scale, precision, bias_fn=None, qkv_precision=None):
    """x (B,N,C) -> proj(softmax(q k^T scale [+ bias]) v) for a fused qkv Linear laid out [3][heads][d]; head widths other than 32 / 64
    run zero padded (functional.head_padded).  Returns (out, qkv tensor, padded width) so that callers can expose q / k / v."""
    dp = F.attn_head_width(d)
    if dp == d:
        wq, bq, wp = qkv.weight, qkv.bias, (proj.weight if proj is not None else None)
    else:
        wq, bq = F.head_padded(qkv.weight, qkv.bias, 3 * heads, d, dp, 0)
        wp = F.head_padded(proj.weight, None, heads, d, dp, 1)[0] if proj is not None else None
    Cp = heads * dp
    t = F.linear(x, wq, bq, precision=precision if qkv_precision is None else qkv_precision)
    q, k, v = t[..., :Cp], t[..., Cp:2 * Cp], t[..., 2 * Cp:]
    bias = bias_fn(q, k) if bias_fn is not None else None
    ctx = F.sdpa_general(q, k, v, heads, scale, bias=bias, precision=precision)
    out = F.linear(ctx, wp, proj.bias, precision=precision) if proj is not None else ctx
    return out, t, dp


class GlobalAttention(nn.Module):
    """dilateformer.py:137-164: the multi-head pattern on a channels-last (B, H, W, C) grid."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., precision=None):
        super().__init__()
        _no_dropout(attn_drop, proj_drop)
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.precision = precision

    def forward(self, x):
        _dropout_is_identity(self)
        B, H, W, C = x.shape
        out, _, _ = _fused_qkv_attention(x.reshape(B, H * W, C), self.qkv, self.proj, self.num_heads, C // self.num_heads, self.scale,
                                         self.precision)
        return out.reshape(B, H, W, C)


class Broad_Attention(nn.Module):
    """bvit.py:49-76: returns (to_out(out), q, k, v) with q, k, v as (b, heads, n, dim_head) views of the fp32 projection."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0., precision=None):
        super().__init__()
        _no_dropout(dropout)
        inner_dim = dim_head * heads
        project_out = not (heads == 1 and dim_head == dim)
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        self.attend = nn.Softmax(dim=-1)
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout)) if project_out else nn.Identity()
        self.precision = precision

    def forward(self, x):
        _dropout_is_identity(self)
        B, N, _ = x.shape
        h, d = self.heads, self.dim_head
        proj = self.to_out[0] if isinstance(self.to_out, nn.Sequential) else None
        out, t, dp = _fused_qkv_attention(x, self.to_qkv, proj, h, d, self.scale, self.precision)
        if proj is None and dp != d:
            out = out.reshape(B, N, h, dp)[..., :d].reshape(B, N, h * d)
        q, k, v = (t[..., i * h * dp:(i + 1) * h * dp].reshape(B, N, h, dp)[..., :d].permute(0, 2, 1, 3) for i in range(3))
        return out, q, k, v


class QKVSplitAttention(nn.Module):
    """efficientformer.py:56-81 (class Attention there): q / k of width query_dim / heads from one Linear, v of width dim / heads from
    another.  Both widths are padded to one kernel head width; the padded q|k|v projection is a single GEMM."""

    def __init__(self, dim, query_dim, num_heads, qkv_bias=False, attn_drop=0, proj_drop=0, precision=None):
        super().__init__()
        assert dim % num_heads == 0
        _no_dropout(attn_drop, proj_drop)
        self.num_heads = num_heads
        self.query_dim = query_dim
        self.scale = (query_dim // num_heads) ** -0.5
        self.qk = nn.Linear(dim, query_dim * 2, bias=qkv_bias)
        self.v = nn.Linear(dim, dim, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.precision = precision

    def _fused(self, dq, dv, dp):
        h = self.num_heads
        ps = [t for t in (self.qk.weight, self.qk.bias, self.v.weight, self.v.bias) if t is not None]
        tag = tuple((t._version, t.data_ptr()) for t in ps)

        def build():
            wqk, bqk = F.head_padded(self.qk.weight, self.qk.bias, 2 * h, dq, dp, 0)
            wv, bv = F.head_padded(self.v.weight, self.v.bias, h, dv, dp, 0)
            w = torch.cat([wqk, wv], dim=0).contiguous()
            b = torch.cat([bqk, bv]).contiguous() if bqk is not None else None
            return w, b

        return F._derived_get(tuple(ps), ("qk_v_fused", dp), tag, build)

    def forward(self, x):
        _dropout_is_identity(self)
        B, N, C = x.shape
        h = self.num_heads
        dq, dv = self.query_dim // h, C // h
        dp = max(F.attn_head_width(dq), F.attn_head_width(dv))
        w, b = self._fused(dq, dv, dp)
        Cp = h * dp
        t = F.linear(x, w, b, precision=self.precision)
        ctx = F.sdpa_general(t[..., :Cp], t[..., Cp:2 * Cp], t[..., 2 * Cp:], h, self.scale, precision=self.precision)
        wp = self.proj.weight if dp == dv else F.head_padded(self.proj.weight, None, h, dv, dp, 1)[0]
        return F.linear(ctx, wp, self.proj.bias, precision=self.precision)


class KNNAttention(nn.Module):
    """kvt.py:67-94: every query attends to its `topk` highest-scoring keys.  The selection is discontinuous -- a key that swaps places
    with its neighbour at the cut changes the row by ~1/topk -- so everything that feeds it runs in the fp32-class split-bf16 mode:
    the qkv projection and the logits (mi355_qk_logits_fwd).  The mask enters the attention kernel as an additive 0 / -1e30 bias;
    the softmax-V part runs in the module's precision mode."""

    def __init__(self, dim, num_heads=4, qkv_bias=False, attn_drop=0, proj_drop=0, topk=100, precision=None):
        super().__init__()
        assert dim % num_heads == 0
        _no_dropout(attn_drop, proj_drop)
        self.topk = topk
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.precision = precision

    def forward(self, x):
        _dropout_is_identity(self)
        B, N, C = x.shape
        if self.topk > N:
            raise RuntimeError(f"KNNAttention: topk {self.topk} > sequence length {N} (torch.topk raises as well)")
        mask = lambda q, k: F.topk_mask_(F.qk_logits(q, k, self.num_heads), self.topk)
        out, _, _ = _fused_qkv_attention(x, self.qkv, self.proj, self.num_heads, C // self.num_heads, self.scale, self.precision, mask,
                                         qkv_precision=F.PREC_STRICT)
        return out


class ConvAttention(nn.Module):
    """cvt.py:46-76 (class Attention there): qkv from depth-wise conv -> BatchNorm2d -> 1x1 conv on an NCHW map, attention over the
    positions, 1x1 conv projection.  Runs token-major in between: the depth-wise conv writes tokens, the 1x1 convs are Linears."""

    def __init__(self, dim, num_heads=8, ks=3, attn_drop=0, proj_drop=0, precision=None):
        super().__init__()
        assert dim % num_heads == 0
        _no_dropout(attn_drop, proj_drop)
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.conv_proj_qkv = nn.Sequential(
            nn.Conv2d(dim, dim, kernel_size=ks, stride=1, padding=(ks - 1) // 2, groups=dim),
            nn.BatchNorm2d(dim),
            nn.Conv2d(dim, 3 * dim, 1),
        )
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Conv2d(dim, dim, 1)
        self.proj_drop = nn.Dropout(proj_drop)
        self.precision = precision

    class _AsLinear:
        """A 1x1 Conv2d seen as the Linear it is (weight (out, in, 1, 1) -> (out, in))."""

        def __init__(self, conv):
            self.weight = conv.weight.view(conv.weight.shape[0], conv.weight.shape[1])
            self.bias = conv.bias
            self.src = conv.weight.data_ptr()

    def _lin(self, name, conv):
        """The view object is kept (derived-weight caches are tied to object identity) until the parameter moves."""
        cur = self.__dict__.get(name)
        if cur is None or cur.src != conv.weight.data_ptr():
            cur = self._AsLinear(conv)
            self.__dict__[name] = cur
        return cur

    def forward(self, x):
        _dropout_is_identity(self)
        if self.training:
            raise RuntimeError("inference engine: BatchNorm runs with its running statistics; call .eval()")
        B, C, H, W = x.shape
        dw, bn, pw = self.conv_proj_qkv[0], self.conv_proj_qkv[1], self.conv_proj_qkv[2]
        tokens = F.dwconv_bn_nchw_tokens(x, dw.weight, dw.bias, bn)
        out, _, _ = _fused_qkv_attention(tokens, self._lin("_pw_lin", pw), self._lin("_proj_lin", self.proj), self.num_heads,
                                         C // self.num_heads, self.scale, self.precision)
        return F.tokens_to_nchw(out, H, W)


class PoolingAttention(nn.Module):
    """p2t.py:46-95: queries from every token, keys / values from a pyramid of adaptively pooled token grids (each refined by a
    depth-wise 3x3 conv handed in by the enclosing block), LayerNorm, fused kv Linear."""

    def __init__(self, dim, num_heads=2, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., pool_ratios=[1, 2, 3, 6], precision=None):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        _no_dropout(attn_drop, proj_drop)
        self.dim = dim
        self.num_heads = num_heads
        self.num_elements = sum(t * t for t in pool_ratios)
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.q = nn.Sequential(nn.Linear(dim, dim, bias=qkv_bias))
        self.kv = nn.Sequential(nn.Linear(dim, dim * 2, bias=qkv_bias))
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.pool_ratios = pool_ratios
        self.pools = nn.ModuleList()
        self.norm = nn.LayerNorm(dim)
        self.precision = precision

    def forward(self, x, H, W, d_convs=None):
        _dropout_is_identity(self)
        B, N, C = x.shape
        p, h = self.precision, self.num_heads
        d = C // h
        dp = F.attn_head_width(d)
        sizes = [(round(H / r), round(W / r)) for r in self.pool_ratios]
        pools = F.pooled_pyramid_tokens(x, H, W, sizes, d_convs)
        pools = F.layernorm(pools, self.norm.weight, self.norm.bias, self.norm.eps)
        ql, kvl = self.q[0], self.kv[0]
        if dp == d:
            wq, bq, wkv, bkv, wp = ql.weight, ql.bias, kvl.weight, kvl.bias, self.proj.weight
        else:
            wq, bq = F.head_padded(ql.weight, ql.bias, h, d, dp, 0)
            wkv, bkv = F.head_padded(kvl.weight, kvl.bias, 2 * h, d, dp, 0)
            wp = F.head_padded(self.proj.weight, None, h, d, dp, 1)[0]
        Cp = h * dp
        q = F.linear(x, wq, bq, precision=p)
        kv = F.linear(pools, wkv, bkv, precision=p)
        ctx = F.sdpa_general(q, kv[..., :Cp], kv[..., Cp:], h, self.scale, precision=p)
        return F.linear(ctx, wp, self.proj.bias, precision=p)
