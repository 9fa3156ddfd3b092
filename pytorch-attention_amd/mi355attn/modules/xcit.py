"""Drop-in XCiT modules (reference: vision_transformers/xcit.py), forward routed to libmi355attn.

  XCA       xcit.py:233-265  cross-covariance attention: (d x d) attention over channels, L2-normalised q/k
  LPI       xcit.py:128-157  dw3x3 -> GELU -> BatchNorm2d(eval) -> dw3x3 on the token grid
  XCABlock  xcit.py:267-294  x += g1*XCA(LN1 x); x += g3*LPI(LN3 x); x += g2*Mlp(LN2 x)

  PositionalEncodingFourier  xcit.py:42-77   input-independent sin/cos table -> 1x1 conv (one small GEMM, cached)
  ConvPatchEmbed             xcit.py:79-126  3x3 stride-2 conv + BatchNorm (+GELU) chain as implicit GEMMs, BN folded
  ClassAttention(Block)      xcit.py:159-231 cls-token attention stage, including the reference's token-path quirks
  XCiT, xcit_nano_12_p16     xcit.py:296-414

BatchNorm runs with its running statistics (inference engine); compare against the reference in .eval().
"""
import math

import torch
from torch import nn

from .. import functional as F
from .vit import _fast, _mlp16
from .mhsa import _dropout_is_identity, _no_dropout


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, bias=True, drop=0.):
        super().__init__()
        if act_layer is not nn.GELU:
            raise NotImplementedError("only the exact-erf GELU epilogue is built")
        _no_dropout(drop)
        self.fc1 = nn.Linear(in_features, hidden_features or in_features, bias=bias)
        self.drop1 = nn.Dropout(drop)                      # xcit.py:28,30: the identity in eval mode, refused in train mode
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features, bias=bias)
        self.drop2 = nn.Dropout(drop)
        self.precision = None

    def forward(self, x, gamma=None, resid=None):
        _dropout_is_identity(self)
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

    def forward(self, x, H, W, gamma=None, resid=None, ln=None, stats=None):
        """`ln`: the LayerNorm in front of the block (XCABlock passes norm3 with the un-normalised x): fused into the kernel.  `stats`: that
        LayerNorm's (mean, rstd) per token when the producer of x has already written them (XCA's proj GEMM)."""
        bn = self.bn
        return F.lpi(x, self.conv1.weight, self.conv1.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                     self.conv2.weight, self.conv2.bias, H, W, gamma=gamma, resid=resid, ln=ln, stats=stats)


class XCA(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., precision=None):
        super().__init__()
        self.num_heads = num_heads
        _no_dropout(attn_drop, proj_drop)
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)             # xcit.py:241,243: eval-mode identities
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.precision = precision

    def forward(self, x, gamma=None, resid=None, ln=None, stats_eps=None):
        """`ln`: the LayerNorm in front of the block (XCABlock passes norm1 with the un-normalised x) -- applied on the way into the qkv
        GEMM when the width allows (functional.ln_linear16), else by the caller.  `stats_eps` (round 6): the caller wants the LayerNorm
        statistics of the OUTPUT rows (eps of the LayerNorm that follows: XCABlock's norm3) -- the return value is then (y, stats) with
        stats None where the projection kernel does not own whole rows."""
        _dropout_is_identity(self)
        want_stats = stats_eps is not None
        if _fast(self.precision, self.qkv, self.proj):
            p = F._prec(self.precision)      # GEMMs on 16-bit operands; the d x d covariance core itself is exact fp32
            if ln is not None:
                qkv = F.ln_linear16(x, ln, self.qkv, out16=True, precision=p)
            else:
                qkv = F.cast_linear16(x, F.weight16(self.qkv.weight, p), self.qkv.bias, precision=p)    # fp32 x: the cast rides in the GEMM
            ctx16 = F.xca_core(qkv, self.temperature, self.num_heads, precision=p, out16=True)
            folded = F.weight16_scaled(self.proj.weight, self.proj.bias, gamma, p) if gamma is not None else None
            if folded is not None:                            # LayerScale folded into the projection (no activation in between)
                w16, b = folded
                if want_stats and resid is not None:
                    got = F.linear16_stats(ctx16, w16, b, resid, stats_eps, p)
                    if got is not None:
                        return got
                y = F.linear16(ctx16, w16, b, resid=resid, precision=p)
                return (y, None) if want_stats else y
            # no LayerScale, or gamma * W would leave the fp16 normal range (eta = 1e-5 initialisations): gamma in the fp32 epilogue
            y = F.linear16(ctx16, F.weight16(self.proj.weight, p), self.proj.bias, gamma=gamma, resid=resid, precision=p)
            return (y, None) if want_stats else y
        qkv = F.linear(x, self.qkv.weight, self.qkv.bias, precision=self.precision)
        ctx = F.xca_core(qkv, self.temperature, self.num_heads, precision=self.precision)
        y = F.linear(ctx, self.proj.weight, self.proj.bias, gamma=gamma, resid=resid, precision=self.precision)
        return (y, None) if want_stats else y


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
        _dropout_is_identity(self)
        p = F._prec(self.attn.precision)
        fast = _fast(p, self.attn.qkv, self.attn.proj, self.mlp.fc1, self.mlp.fc2)

        def norm(ln, t, to16):
            if to16:
                return F.layernorm16(t, ln.weight, ln.bias, ln.eps, p)
            return F.layernorm(t, ln.weight, ln.bias, ln.eps)

        # the proj GEMM of the attention branch writes norm3's statistics of x beside x where it owns whole rows (gemm16_wreg, round 6)
        if fast and F.ln_linear16_ok(x.shape[-1], 3 * x.shape[-1], p):
            x, st = self.attn(x, gamma=self.gamma1, resid=x, ln=self.norm1, stats_eps=self.norm3.eps)   # LayerNorm fused into the qkv GEMM
        else:
            x, st = self.attn(norm(self.norm1, x, fast), gamma=self.gamma1, resid=x, stats_eps=self.norm3.eps)
        x = self.local_mp(x, H, W, gamma=self.gamma3, resid=x, ln=self.norm3, stats=st)   # norm3 fused into the LPI kernel
        if fast and F.mlp_fused_ok(x.shape[-1], self.mlp.fc1.weight.shape[0], p) and self.mlp.fc1.bias is not None:
            return F.mlp_fused(x, self.norm2, self.mlp.fc1, self.mlp.fc2, gamma=self.gamma2, precision=p)   # LN2 + MLP + LayerScale + residual
        return self.mlp(norm(self.norm2, x, fast), gamma=self.gamma2, resid=x)


class PositionalEncodingFourier(nn.Module):
    """xcit.py:42-77.  The encoding depends on (H, W) and on token_projection only, never on the input: the sin/cos feature
    table is built once per grid with the reference's own fp32 formula (host torch ops on index ranges -- constants, no
    activations involved) and the 1x1 conv runs as a GEMM on the device; the result (H*W, dim) is cached per weight version."""

    def __init__(self, hidden_dim=32, dim=768, temperature=10000):
        super().__init__()
        self.token_projection = nn.Conv2d(hidden_dim * 2, dim, kernel_size=1)
        self.scale = 2 * math.pi
        self.temperature = temperature
        self.hidden_dim = hidden_dim
        self.dim = dim
        self._cache = {}

    def features(self, H, W):
        """(H*W, 2*hidden_dim) rows [pos_y | pos_x] of xcit.py:57-74, batch-independent.  Closed form of the reference's cumsum-of-mask
        construction: pixel (i, j) has the normalised coordinates (i+1) / (H + 1e-6) * 2 pi and (j+1) / (W + 1e-6) * 2 pi; feature pair p
        of an axis holds sin / cos of coordinate / temperature^(2p / hidden_dim).  Every step is the same fp32 operation the reference
        performs (the sums of ones are exact integers), so the table matches it to the bit."""
        f32 = torch.float32
        half = self.hidden_dim // 2

        def axis(n):                                              # (n, hidden_dim): [sin f0, cos f0, sin f1, cos f1, ...]
            coord = torch.arange(1, n + 1, dtype=f32) / (torch.tensor(float(n), dtype=f32) + 1e-6) * self.scale
            freq = self.temperature ** (2 * torch.arange(half, dtype=f32) / self.hidden_dim)
            ang = coord[:, None] / freq[None, :]
            out = torch.empty(n, 2 * half, dtype=f32)
            out[:, 0::2], out[:, 1::2] = ang.sin(), ang.cos()
            return out

        if self.hidden_dim % 2:
            raise ValueError("PositionalEncodingFourier: hidden_dim must be even (sin / cos pairs)")
        fy, fx = axis(H), axis(W)
        feat = torch.empty(H, W, 2 * self.hidden_dim, dtype=f32)
        feat[:, :, :self.hidden_dim] = fy[:, None, :]
        feat[:, :, self.hidden_dim:] = fx[None, :, :]
        return feat.reshape(H * W, 2 * self.hidden_dim)

    def tokens(self, H, W):
        """Position rows (H*W, dim) = token_projection(features), the layout forward_features adds to the patch tokens."""
        w, b = self.token_projection.weight, self.token_projection.bias
        tag = (w._version, w.data_ptr(), b._version, b.data_ptr())
        hit = self._cache.get((H, W))
        if hit is None or hit[0] != tag:
            feat = self.features(H, W).to(w.device)
            pos = F.linear(feat, w.reshape(self.dim, -1), b, precision=F.PREC_STRICT)
            hit = (tag, pos)
            self._cache[(H, W)] = hit
        return hit[1]

    def forward(self, B, H, W):
        """xcit.py:56-77: the encoding as a (B, dim, H, W) map.  (XCiT itself uses tokens(H, W): the rows are added inside the last
        patch-embedding conv.)  The batch axis is a broadcast view, as every image gets the same table."""
        pos = self.tokens(H, W)                                            # (H*W, dim) on the device
        return F.tokens_to_nchw(pos.reshape(1, H * W, self.dim), H, W).expand(B, -1, -1, -1)


def conv3x3(in_channels, out_channels, stride=1):
    return nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=False),
                         nn.BatchNorm2d(out_channels))


class ConvPatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, precision=None):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.precision = precision
        if patch_size == 16:
            self.proj = nn.Sequential(conv3x3(3, embed_dim // 8, 2), nn.GELU(), conv3x3(embed_dim // 8, embed_dim // 4, 2), nn.GELU(),
                                      conv3x3(embed_dim // 4, embed_dim // 2, 2), nn.GELU(), conv3x3(embed_dim // 2, embed_dim, 2))
        elif patch_size == 8:
            self.proj = nn.Sequential(conv3x3(3, embed_dim // 4, 2), nn.GELU(), conv3x3(embed_dim // 4, embed_dim // 2, 2), nn.GELU(),
                                      conv3x3(embed_dim // 2, embed_dim, 2))
        else:
            raise ValueError("For convolutional projection, patch size has to be in [8, 16]")

    def forward(self, x, padding_size=None, pos=None):
        """Returns (tokens (B, Hp*Wp, E), (Hp, Wp)).  Layer 1 gathers from NCHW, the rest from the token-major output of the
        previous layer; BatchNorm is folded into weights/bias, GELU and (last layer) the position rows ride in the epilogue."""
        stages = [m for m in self.proj if isinstance(m, nn.Sequential)]
        hw = None
        for li, st in enumerate(stages):
            layout = 0 if li == 0 else 1
            wrows, bias = F.conv_bn_rows(st[0].weight, st[1], layout)
            last = li == len(stages) - 1
            x, hw = F.conv2d_tokens(x, None, bias, 3, 2, 1, layout, hw=hw, precision=self.precision,
                                    act=F.ACT_NONE if last else F.ACT_GELU, pos=pos if last else None, wrows=wrows)
        return x, hw


class ClassAttention(nn.Module):
    """xcit.py:159-188.  Only the cls query is ever used (:180), so q is projected for the cls rows alone; k, v for all tokens."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., precision=None):
        super().__init__()
        self.num_heads = num_heads
        _no_dropout(attn_drop, proj_drop)
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)             # xcit.py:170,172: eval-mode identities
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.precision = precision

    def cls_out(self, u):
        """proj(attention of the cls query over all tokens) for normed tokens u (B,N,C) -> (B,C)."""
        B, N, C = u.shape
        w, b = self.qkv.weight, self.qkv.bias
        kv = F.linear(u, w[C:], None if b is None else b[C:], precision=self.precision)                  # (B,N,2C)
        qc = F.linear(u[:, 0], w[:C], None if b is None else b[:C], precision=self.precision)          # (B,C), cls rows in place
        att = F.class_attention(qc, kv, kv[:, :, C:], self.num_heads, self.scale, N, C, 2 * C)
        return F.linear(att, self.proj.weight, self.proj.bias, precision=self.precision)

    def forward(self, x):
        _dropout_is_identity(self)
        B, N, C = x.shape
        out = torch.empty_like(x)
        F.axpby(x[:, 1:], out[:, 1:], B, (N - 1) * C, N * C, N * C)                  # tokens pass through (xcit.py:187)
        F.axpby(self.cls_out(x), out, B, C, C, N * C)
        return out


class ClassAttentionBlock(nn.Module):
    """xcit.py:190-231, reproduced with its token-path behaviour: the attention returns the NORMED patch tokens (:187), so
    patch tokens become x + gamma1*LN1(x) (:219); the second residual adds them to themselves (:226-230), i.e. doubles them."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm, eta=None, tokens_norm=False, precision=None):
        super().__init__()
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("only nn.LayerNorm is built")
        self.norm1 = norm_layer(dim)
        self.attn = ClassAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                                   proj_drop=drop, precision=precision)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.mlp.precision = precision
        if eta is not None:
            self.gamma1 = nn.Parameter(eta * torch.ones(dim), requires_grad=True)
            self.gamma2 = nn.Parameter(eta * torch.ones(dim), requires_grad=True)
        else:
            self.gamma1, self.gamma2 = 1.0, 1.0
        self.tokens_norm = tokens_norm           # xcit.py:221-222: norm2 over every token instead of the cls token only

    def forward(self, x, H, W, mask=None):
        B, N, C = x.shape
        g1 = self.gamma1 if isinstance(self.gamma1, torch.Tensor) else None
        g2 = self.gamma2 if isinstance(self.gamma2, torch.Tensor) else None
        u = F.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        if self.tokens_norm:
            # patch tokens: 2 * LN2(x + g1 * LN1(x)) -- the doubling of the second residual rides in the LayerNorm's affine part
            t1 = torch.empty_like(x)
            F.axpby(x, t1, B * N, C, C, C, alpha=1.0, u=u, ldu=C, gamma=g1)
            w2, b2 = _twice(self, self.norm2.weight, "_n2w"), _twice(self, self.norm2.bias, "_n2b")
            out = F.layernorm(t1, w2, b2, self.norm2.eps)
        else:
            out = torch.empty_like(x)
            # patch tokens: 2 * (x + g1 * LN1(x)); the cls rows written here are replaced below
            F.axpby(x, out, B * N, C, C, C, alpha=2.0, u=u, ldu=C, gamma=None if g1 is None else _twice(self, g1))
        # cls token: c1 = x0 + g1 * proj(attn); c2 = LN2(c1); out0 = c2 + g2 * mlp(c2)
        c1 = torch.empty(B, C, dtype=torch.float32, device=x.device)
        F.axpby(x, c1, B, C, N * C, C, u=self.attn.cls_out(u), ldu=C, gamma=g1)
        c2 = F.layernorm(c1, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        c3 = self.mlp(c2, gamma=g2, resid=c2)
        F.axpby(c3, out, B, C, C, N * C)
        return out


def _twice(owner, g, slot="_g1x2"):
    """2 * parameter for the doubled patch-token path, cached per parameter version on the block."""
    tag = (g._version, g.data_ptr())
    hit = getattr(owner, slot, None)
    if hit is None or hit[0] != tag:
        buf = torch.empty_like(g.detach())
        F.axpby(g.detach(), buf, 1, g.numel(), g.numel(), g.numel(), alpha=2.0)
        hit = (tag, buf)
        setattr(owner, slot, hit)
    return hit[1]


class XCiT(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., norm_layer=None, cls_attn_layers=2,
                 use_pos=True, patch_proj='linear', eta=None, tokens_norm=False, precision=None):
        super().__init__()
        _no_dropout(drop_rate, attn_drop_rate)             # accepted as the reference accepts them: eval-mode identities (xcit.py:333-346)
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = ConvPatchEmbed(img_size=img_size, embed_dim=embed_dim, patch_size=patch_size, precision=precision)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.blocks = nn.ModuleList([
            XCABlock(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                     attn_drop=attn_drop_rate, norm_layer=norm_layer, num_tokens=num_patches, eta=eta, precision=precision)
            for _ in range(depth)])
        self.cls_attn_blocks = nn.ModuleList([
            ClassAttentionBlock(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                drop=drop_rate, attn_drop=attn_drop_rate, norm_layer=norm_layer, eta=eta, tokens_norm=tokens_norm,
                                precision=precision)
            for _ in range(cls_attn_layers)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.pos_embeder = PositionalEncodingFourier(dim=embed_dim)
        self.use_pos = use_pos
        self.precision = precision
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward_features(self, x):
        _dropout_is_identity(self)
        B = x.shape[0]
        ps = self.patch_embed.patch_size
        Hp, Wp = x.shape[2] // ps, x.shape[3] // ps
        pos = self.pos_embeder.tokens(Hp, Wp) if self.use_pos else None
        x, (Hp, Wp) = self.patch_embed(x, pos=pos)                    # + position rows in the last conv's epilogue (:398-400)
        for blk in self.blocks:
            x = blk(x, Hp, Wp)
        N, C = x.shape[1], x.shape[2]
        tok = torch.empty(B, N + 1, C, dtype=torch.float32, device=x.device)       # cat(cls, x) (:405-406)
        F.axpby(self.cls_token, tok, B, C, 0, (N + 1) * C)
        F.axpby(x, tok[:, 1:], B, N * C, N * C, (N + 1) * C)
        x = tok
        for blk in self.cls_attn_blocks:
            x = blk(x, Hp, Wp)
        cls = torch.empty(B, C, dtype=torch.float32, device=x.device)
        F.axpby(x, cls, B, C, (N + 1) * C, C)                                       # gather the cls rows
        return F.layernorm(cls, self.norm.weight, self.norm.bias, self.norm.eps)    # norm(x)[:, 0]: LayerNorm is per token

    def forward(self, x):
        x = self.forward_features(x)
        if isinstance(self.head, nn.Identity):
            return x
        return F.linear(x, self.head.weight, self.head.bias, precision=self.precision)


def xcit_nano_12_p16(pretrained=False, **kwargs):
    if pretrained:
        raise NotImplementedError("no checkpoints are bundled (the reference ships none either)")
    return XCiT(patch_size=16, embed_dim=128, depth=12, num_heads=4, mlp_ratio=4, qkv_bias=True, norm_layer=nn.LayerNorm, eta=1.0,
                tokens_norm=False, **kwargs)
