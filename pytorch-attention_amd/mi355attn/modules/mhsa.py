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
    if any(rates):
        raise NotImplementedError("inference engine: dropout rates must be 0")


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
        C = x.shape[-1]
        qkv = F.linear(x, self.qkv.weight, self.qkv.bias, precision=self.precision)
        ctx = F.sdpa_general(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], self.num_heads, self.scale, precision=self.precision)
        return F.linear(ctx, self.proj.weight, self.proj.bias, precision=self.precision)


class SRAttention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0, proj_drop=0, sr_ratio=1, precision=None):
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
        p = self.precision
        C = x.shape[-1]
        q = F.linear(x, self.q.weight, self.q.bias, precision=p)
        src = x
        if self.sr_ratio > 1:
            src, _ = F.conv2d_tokens(x, self.sr.weight, self.sr.bias, self.sr_ratio, self.sr_ratio, 0, 1, hw=(H, W), precision=p)
        kv = F.linear(src, self.kv.weight, self.kv.bias, precision=p)
        ctx = F.sdpa_general(q, kv[..., :C], kv[..., C:], self.num_heads, self.scale, precision=p)
        return F.linear(ctx, self.proj.weight, self.proj.bias, precision=p)
