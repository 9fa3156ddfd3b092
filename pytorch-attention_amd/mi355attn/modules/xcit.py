"""Drop-in XCiT modules (reference: vision_transformers/xcit.py), forward routed to libmi355attn.

  XCA       xcit.py:233-265  cross-covariance attention: (d x d) attention over channels, L2-normalised q/k
  LPI       xcit.py:128-157  dw3x3 -> GELU -> BatchNorm2d(eval) -> dw3x3 on the token grid
  XCABlock  xcit.py:267-294  x += g1*XCA(LN1 x); x += g3*LPI(LN3 x); x += g2*Mlp(LN2 x)

BatchNorm runs with its running statistics (inference engine); compare against the reference in .eval().
"""
import torch
from torch import nn

from .. import functional as F
from .vit import _fast, _mlp16


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, bias=True, drop=0.):
        super().__init__()
        if act_layer is not nn.GELU:
            raise NotImplementedError("only the exact-erf GELU epilogue is built")
        self.fc1 = nn.Linear(in_features, hidden_features or in_features, bias=bias)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features, bias=bias)
        self.precision = None

    def forward(self, x, gamma=None, resid=None):
        if self.fc1.bias is not None and _fast(self.precision, self.fc1, self.fc2):
            return _mlp16(x, self.fc1, self.fc2, self.precision, second_gelu=False, gamma=gamma, resid=resid)
        h = F.linear(x, self.fc1.weight, self.fc1.bias, act=F.ACT_GELU, precision=self.precision)
        return F.linear(h, self.fc2.weight, self.fc2.bias, gamma=gamma, resid=resid, precision=self.precision)


class LPI(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0., kernel_size=3):
        super().__init__()
        out_features = out_features or in_features
        if kernel_size != 3 or out_features != in_features or act_layer is not nn.GELU:
            raise NotImplementedError("LPI kernel is built for depth-wise 3x3 + GELU only")
        self.conv1 = nn.Conv2d(in_features, out_features, kernel_size=3, padding=1, groups=out_features)
        self.bn = nn.BatchNorm2d(in_features)
        self.conv2 = nn.Conv2d(in_features, out_features, kernel_size=3, padding=1, groups=out_features)

    def forward(self, x, H, W, gamma=None, resid=None):
        bn = self.bn
        return F.lpi(x, self.conv1.weight, self.conv1.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                     self.conv2.weight, self.conv2.bias, H, W, gamma=gamma, resid=resid)


class XCA(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., precision=None):
        super().__init__()
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.precision = precision

    def forward(self, x, gamma=None, resid=None):
        if _fast(self.precision, self.qkv, self.proj):
            p = F._prec(self.precision)      # GEMMs on 16-bit operands; the d x d covariance core itself is exact fp32
            x16 = x if x.dtype != torch.float32 else F.cast16(x, p)
            qkv = F.linear16(x16, F.weight16(self.qkv.weight, p), self.qkv.bias, precision=p)
            ctx = F.xca_core(qkv, self.temperature, self.num_heads, precision=p)
            return F.linear16(F.cast16(ctx, p), F.weight16(self.proj.weight, p), self.proj.bias, gamma=gamma, resid=resid,
                              precision=p)
        qkv = F.linear(x, self.qkv.weight, self.qkv.bias, precision=self.precision)
        ctx = F.xca_core(qkv, self.temperature, self.num_heads, precision=self.precision)
        return F.linear(ctx, self.proj.weight, self.proj.bias, gamma=gamma, resid=resid, precision=self.precision)


class XCABlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, num_tokens=196, eta=None, precision=None):
        super().__init__()
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("only nn.LayerNorm is built")
        self.norm1 = norm_layer(dim)
        self.attn = XCA(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                        proj_drop=drop, precision=precision)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.mlp.precision = precision
        self.norm3 = norm_layer(dim)
        self.local_mp = LPI(in_features=dim, act_layer=act_layer)
        # eta=None multiplies None by a tensor in the reference (xcit.py:286): same TypeError here
        self.gamma1 = nn.Parameter(eta * torch.ones(dim), requires_grad=True)
        self.gamma2 = nn.Parameter(eta * torch.ones(dim), requires_grad=True)
        self.gamma3 = nn.Parameter(eta * torch.ones(dim), requires_grad=True)

    def forward(self, x, H, W):
        p = F._prec(self.attn.precision)
        fast = _fast(p, self.attn.qkv, self.attn.proj, self.mlp.fc1, self.mlp.fc2)

        def norm(ln, t, to16):
            if to16:
                return F.layernorm16(t, ln.weight, ln.bias, ln.eps, p)
            return F.layernorm(t, ln.weight, ln.bias, ln.eps)

        x = self.attn(norm(self.norm1, x, fast), gamma=self.gamma1, resid=x)
        x = self.local_mp(norm(self.norm3, x, False), H, W, gamma=self.gamma3, resid=x)
        return self.mlp(norm(self.norm2, x, fast), gamma=self.gamma2, resid=x)
