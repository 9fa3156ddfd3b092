"""Drop-in CSWin modules (reference: vision_transformers/cswin.py), forward routed to libmi355attn.

  LePEAttention  cswin.py:51-127   stripe-window MHSA + locally-enhanced positional encoding (dw 3x3 on v)
  CSWinBlock     cswin.py:130-197  LN -> qkv -> two stripe branches on channel halves -> proj -> MLP

The window partition (img2windows / windows2img, cswin.py:199-216) never materialises: the attention
kernel gathers q/k/v straight from the (B,L,3,C) qkv buffer with window index math and scatters its
output back in (B,L,C) order.
"""
import torch
from torch import nn

from .. import functional as F
from .vit import _fast, _mlp16


def _stripe(resolution, idx, split_size):
    if idx == -1:
        return resolution, resolution
    if idx == 0:
        return resolution, split_size
    if idx == 1:
        return split_size, resolution
    raise ValueError(f"LePEAttention: idx must be -1, 0 or 1, got {idx}")


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        if act_layer is not nn.GELU:
            raise NotImplementedError("only the exact-erf GELU epilogue is built")
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.precision = None

    def forward(self, x, resid=None):
        if _fast(self.precision, self.fc1, self.fc2):
            return _mlp16(x, self.fc1, self.fc2, self.precision, second_gelu=False, resid=resid)
        h = F.linear(x, self.fc1.weight, self.fc1.bias, act=F.ACT_GELU, precision=self.precision)
        return F.linear(h, self.fc2.weight, self.fc2.bias, resid=resid, precision=self.precision)


class LePEAttention(nn.Module):
    def __init__(self, dim, resolution, idx, split_size=7, dim_out=None, num_heads=8, attn_drop=0., proj_drop=0.,
                 qk_scale=None, precision=None):
        super().__init__()
        self.dim, self.dim_out = dim, dim_out or dim
        self.resolution, self.split_size, self.num_heads = resolution, split_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.H_sp, self.W_sp = _stripe(resolution, idx, split_size)
        self.get_v = nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1, groups=dim)
        self.precision = precision

    def run(self, qkv_blc, out, c0):
        """Attend over channels [c0, c0+dim) of a (B,L,3,Ctot) buffer, writing the same slice of `out` (B,L,Ctot).
        fp32 buffers use the converting kernel, 16-bit buffers the 16-bit-I/O kernel."""
        fn = F.cswin_lepe_attention if qkv_blc.dtype == torch.float32 else F.cswin_lepe_attention16
        return fn(qkv_blc, self.get_v.weight, self.get_v.bias, out, self.resolution, c0, self.dim, self.num_heads, self.H_sp,
                  self.W_sp, self.scale, self.precision)

    def forward(self, qkv):
        """qkv: (3,B,L,C) as in the reference (cswin.py:101-105); returns (B,L,C)."""
        three, B, L, C = qkv.shape
        assert three == 3 and C == self.dim and L == self.resolution * self.resolution, "flatten img_tokens has wrong size"
        buf = qkv.permute(1, 2, 0, 3).contiguous().view(B, L, 3 * C)     # no copy when qkv is the usual permuted view
        out = torch.empty(B, L, C, dtype=torch.float32, device=qkv.device)
        return self.run(buf, out, 0)


class CSWinBlock(nn.Module):
    def __init__(self, dim, reso, num_heads, split_size=7, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0.,
                 attn_drop=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, last_stage=False, precision=None):
        super().__init__()
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("only nn.LayerNorm is built")
        self.dim, self.num_heads, self.patches_resolution = dim, num_heads, reso
        self.split_size, self.mlp_ratio, self.precision = split_size, mlp_ratio, precision
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.norm1 = norm_layer(dim)
        last_stage = last_stage or reso == split_size                   # cswin.py:146-147
        self.branch_num = 1 if last_stage else 2
        self.proj = nn.Linear(dim, dim)
        if last_stage:
            branches = [LePEAttention(dim, reso, -1, split_size, dim, num_heads, attn_drop, drop, qk_scale, precision)]
        else:
            branches = [LePEAttention(dim // 2, reso, i, split_size, dim // 2, num_heads // 2, attn_drop, drop, qk_scale,
                                      precision) for i in range(2)]
        self.attns = nn.ModuleList(branches)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), dim, act_layer, drop)
        self.mlp.precision = precision
        self.norm2 = norm_layer(dim)

    def forward(self, x):
        B, L, C = x.shape
        assert L == self.patches_resolution ** 2, "flatten img_tokens has wrong size"
        p = F._prec(self.precision)
        fast = _fast(p, self.qkv, self.proj, self.mlp.fc1, self.mlp.fc2) and self.attns[0].dim // self.attns[0].num_heads == 32
        stripe = fast and self.branch_num == 2 and F.cswin_stripe_ok(C, self.patches_resolution, self.split_size, self.num_heads, p)
        if stripe:                   # narrow stages: LN -> qkv -> both stripe attentions in one kernel, qkv never reaches HBM
            a0, a1 = self.attns[0], self.attns[1]
            att = F.cswin_stripe_attention(x, self.norm1, self.qkv, a0.get_v, a1.get_v, self.patches_resolution, self.num_heads,
                                           self.split_size, a0.scale, p)
        elif fast:                   # LN -> qkv -> attention -> proj with every GEMM operand kept 16-bit in HBM
            if F.ln_linear16_ok(C, 3 * C, p):        # narrow stages: LayerNorm applied on the way into the qkv GEMM
                qkv = F.ln_linear16(x, self.norm1, self.qkv, out16=True, precision=p)
            else:
                u = F.layernorm16(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, p)
                qkv = F.linear16(u, F.weight16(self.qkv.weight, p), self.qkv.bias, out16=True, precision=p)
            att = torch.empty(B, L, C, dtype=qkv.dtype, device=x.device)
        else:
            u = F.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            qkv = F.linear(u, self.qkv.weight, self.qkv.bias, precision=self.precision)    # (B,L,3C) == (B,L,3,C)
            att = torch.empty(B, L, C, dtype=torch.float32, device=x.device)
        if stripe:
            pass
        elif self.branch_num == 2 and fast:
            a0, a1 = self.attns[0], self.attns[1]                  # both stripe branches in one launch
            F.cswin_lepe_attention16_pair(qkv, a0.get_v.weight, a0.get_v.bias, a1.get_v.weight, a1.get_v.bias, att,
                                          self.patches_resolution, a0.num_heads, self.split_size, a0.scale, p)
        elif self.branch_num == 2:
            self.attns[0].run(qkv, att, 0)
            self.attns[1].run(qkv, att, C // 2)
        else:
            self.attns[0].run(qkv, att, 0)
        if fast:
            if F.proj_mlp_fused_ok(C, self.mlp.fc1.weight.shape[0], p) and self.mlp.fc1.bias is not None:
                # proj + residual + LN2 + fc1 + GELU + fc2 + residual in one launch: x1 never reaches HBM
                return F.mlp_fused(x, self.norm2, self.mlp.fc1, self.mlp.fc2, precision=p, ctx16=att, proj=self.proj)
            fused_mlp = F.mlp_fused_ok(C, self.mlp.fc1.weight.shape[0], p) and self.mlp.fc1.bias is not None
            if not fused_mlp:
                # stage 3 (C = 256): the proj GEMM owns whole rows, so it also writes norm2(x) in the operand format (round 6: no LayerNorm launch)
                got = F.linear16_ln16(att, F.weight16(self.proj.weight, p), self.proj.bias, x, self.norm2, p)
                if got is not None:
                    x, u = got
                    return self.mlp(u, resid=x)
            x = F.linear16(att, F.weight16(self.proj.weight, p), self.proj.bias, resid=x, precision=p)
            if fused_mlp:
                return F.mlp_fused(x, self.norm2, self.mlp.fc1, self.mlp.fc2, precision=p)      # LN2 + fc1 + GELU + fc2 + residual
            u = F.layernorm16(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, p)
        else:
            x = F.linear(att, self.proj.weight, self.proj.bias, resid=x, precision=self.precision)
            u = F.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return self.mlp(u, resid=x)


class Merge_Block(nn.Module):
    """cswin.py:218-233: conv 3x3 stride 2 between stages, as an implicit GEMM straight on the token-major activations."""

    def __init__(self, dim, dim_out, norm_layer=nn.LayerNorm, precision=None):
        super().__init__()
        self.conv = nn.Conv2d(dim, dim_out, 3, 2, 1)
        self.norm = norm_layer(dim_out)
        self.precision = precision

    def forward(self, x):
        B, L, C = x.shape
        H = W = int(round(L ** 0.5))
        y, _ = F.conv2d_tokens(x, self.conv.weight, self.conv.bias, 3, 2, 1, in_layout=1, hw=(H, W), precision=self.precision)
        return F.layernorm(y, self.norm.weight, self.norm.bias, self.norm.eps)


class CSWinTransformer(nn.Module):
    """Full CSWin (cswin.py:235-346): stem conv 7x7/4 + LN, four stages of CSWinBlocks with Merge_Blocks between, LN, token mean,
    head.  Same constructor, state_dict and init stream as the reference (activation checkpointing is a training feature: ignored)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=96, depth=[2, 2, 6, 2],
                 split_size=[3, 5, 7], num_heads=12, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 hybrid_backbone=None, norm_layer=nn.LayerNorm, use_chk=False, precision=None):
        super().__init__()
        self.use_chk = use_chk
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.precision = precision
        heads = num_heads
        self.stage1_conv_embed = nn.Sequential(nn.Conv2d(in_chans, embed_dim, 7, 4, 2), nn.Identity(), nn.LayerNorm(embed_dim))

        def stage(dim, n, head, reso, split, last=False):
            return nn.ModuleList([CSWinBlock(dim=dim, num_heads=head, reso=reso, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                             qk_scale=qk_scale, split_size=split, drop=drop_rate, attn_drop=attn_drop_rate,
                                             norm_layer=norm_layer, last_stage=last, precision=precision) for _ in range(n)])

        d = embed_dim
        self.stage1 = stage(d, depth[0], heads[0], img_size // 4, split_size[0])
        self.merge1 = Merge_Block(d, d * 2, precision=precision)
        d *= 2
        self.stage2 = stage(d, depth[1], heads[1], img_size // 8, split_size[1])
        self.merge2 = Merge_Block(d, d * 2, precision=precision)
        d *= 2
        self.stage3 = stage(d, depth[2], heads[2], img_size // 16, split_size[2])
        self.merge3 = Merge_Block(d, d * 2, precision=precision)
        d *= 2
        self.stage4 = stage(d, depth[-1], heads[3], img_size // 32, split_size[-1], last=True)
        self.norm = norm_layer(d)
        self.head = nn.Linear(d, num_classes) if num_classes > 0 else nn.Identity()
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward_features(self, x):
        stem, ln = self.stage1_conv_embed[0], self.stage1_conv_embed[2]
        x, _ = F.conv2d_tokens(x, stem.weight, stem.bias, 7, 4, 2, in_layout=0, precision=self.precision)
        x = F.layernorm(x, ln.weight, ln.bias, ln.eps)
        for blk in self.stage1:
            x = blk(x)
        for pre, blocks in ((self.merge1, self.stage2), (self.merge2, self.stage3), (self.merge3, self.stage4)):
            x = pre(x)
            for blk in blocks:
                x = blk(x)
        x = F.layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return F.token_mean(x)

    def forward(self, x):
        x = self.forward_features(x)
        if isinstance(self.head, nn.Identity):
            return x
        return F.linear(x, self.head.weight, self.head.bias, precision=self.precision)


def CSWin_64_12211_tiny_224(pretrained=False, **kwargs):
    return CSWinTransformer(patch_size=4, embed_dim=64, depth=[1, 2, 21, 1], split_size=[1, 2, 7, 7], num_heads=[2, 4, 8, 16],
                            mlp_ratio=4., **kwargs)


def CSWin_64_24322_small_224(pretrained=False, **kwargs):
    return CSWinTransformer(patch_size=4, embed_dim=64, depth=[2, 4, 32, 2], split_size=[1, 2, 7, 7], num_heads=[2, 4, 8, 16],
                            mlp_ratio=4., **kwargs)
