"""Drop-in modules for four more members of the reference's attention zoo whose gates come from AXIS reductions of
the NCHW activation (csrc/axis_attn.hip):

  GCModule             attention_mechanisms/gc_module.py:17-43          global context (no softmax, as in the reference) -> bottleneck
                                                                        transform with LayerNorm -> added to every position
  CoordinateAttention  attention_mechanisms/coordatten.py:18-44         row / column means -> shared 1x1 conv + BN + ReLU -> two 1x1 convs
  TripletAttention     attention_mechanisms/triplet_attention.py:19-63  three rotated AttentionGates, averaged
  BAM                  attention_mechanisms/bam.py:16-71                channel gate (MLP + BN1d) + spatial gate (dilated convs) -> sigmoid

Same constructor signatures, submodule names and state_dict keys as the reference.  BatchNorm layers run in eval mode (running
statistics); calling forward() in training mode raises.  The helper classes that only exist as parameter containers in the
reference (BasicConv2d, ZPool, AttentionGate, ChannelGate, SpatialGate) keep their names and parameters; the fused forward lives in
the top-level module.
"""
import torch
from torch import nn

from .. import functional as F


def _eval_only(m):
    if m.training:
        raise RuntimeError("inference engine: BatchNorm runs with its running statistics; call .eval()")


class GCModule(nn.Module):
    def __init__(self, channel, reduction=16):
        super().__init__()
        self.conv = nn.Conv2d(channel, 1, kernel_size=1)
        self.softmax = nn.Softmax(dim=2)            # declared and unused in the reference (gc_module.py:22, :30-38)
        self.transform = nn.Sequential(
            nn.Conv2d(channel, channel // reduction, kernel_size=1),
            nn.LayerNorm([channel // reduction, 1, 1]),
            nn.ReLU(inplace=True),
            nn.Conv2d(channel // reduction, channel, kernel_size=1),
        )

    def forward(self, x):
        t = self.transform
        return F.gc_forward(x, self.conv.weight, self.conv.bias, t[0].weight, t[0].bias, t[1].weight, t[1].bias, t[1].eps,
                            t[3].weight, t[3].bias)


class CoordinateAttention(nn.Module):
    def __init__(self, in_dim, out_dim, reduction=32):
        super().__init__()
        self.pool_h = nn.AdaptiveAvgPool2d((None, 1))
        self.pool_w = nn.AdaptiveAvgPool2d((1, None))
        hidden_dim = max(8, in_dim // reduction)
        self.conv1 = nn.Conv2d(in_dim, hidden_dim, kernel_size=1, stride=1, padding=0)
        self.bn1 = nn.BatchNorm2d(hidden_dim)
        self.act = nn.ReLU(inplace=True)
        self.conv_h = nn.Conv2d(hidden_dim, out_dim, kernel_size=1, stride=1, padding=0)
        self.conv_w = nn.Conv2d(hidden_dim, out_dim, kernel_size=1, stride=1, padding=0)
        self.in_dim, self.out_dim = in_dim, out_dim

    def forward(self, x):
        _eval_only(self)
        if self.out_dim != self.in_dim:
            raise ValueError("CoordinateAttention: identity * a_h * a_w (coordatten.py:43) needs out_dim == in_dim")
        scale, shift = F.bn_fold(self.bn1)
        return F.coordatt_forward(x, self.conv1.weight, self.conv1.bias, scale, shift, self.conv_h.weight, self.conv_h.bias,
                                  self.conv_w.weight, self.conv_w.bias)


class BasicConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, ks):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=ks, stride=1, padding=(ks - 1) // 2)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = nn.ReLU(inplace=True)

    def forward(self, x):
        """conv -> bn -> relu (triplet_attention.py:25-29) on its own: the implicit-GEMM conv with the BatchNorm folded into its rows
        and a ReLU epilogue, then the token rows back to NCHW.  (Inside TripletAttention the 2 -> 1 gates run fused instead.)"""
        _eval_only(self)
        ks = self.conv.kernel_size[0]
        s, t = F.bn_fold(self.bn, self.conv.bias)
        tag = (id(s), self.conv.weight._version, self.conv.weight.data_ptr())
        rows = F._derived_get((self.conv.weight, self.bn), ("basicconv_rows",), tag,
                              lambda: F._gemm_rows(self.conv.weight.detach() * s[:, None, None, None], 0))
        B, _, H, W = x.shape
        tok, (oh, ow) = F.conv2d_tokens(x, None, t, ks, 1, (ks - 1) // 2, 0, precision=F.PREC_STRICT, act=F.ACT_RELU, wrows=rows)
        return F.tokens_to_nchw(tok, oh, ow)


class ZPool(nn.Module):
    def forward(self, x):
        return F.zpool(x)                                     # triplet_attention.py:31-36


class AttentionGate(nn.Module):
    def __init__(self, kernel_size=7):
        super().__init__()
        self.compress = ZPool()
        self.conv = BasicConv2d(2, 1, kernel_size)
        self.activation = nn.Sigmoid()

    def forward(self, x):
        """x * sigmoid(conv(compress(x))) (triplet_attention.py:45-49) on its own, through the same gate kernel TripletAttention uses."""
        _eval_only(self)
        s, t = F.bn_fold(self.conv.bn, self.conv.conv.bias)
        tag = (id(s), id(t))
        affine = F._derived_get((self.conv.bn,), ("gate_affine",), tag, lambda: torch.cat([s, t]).contiguous())
        return F.attention_gate(x, self.conv.conv.weight, affine, self.conv.conv.kernel_size[0])


class TripletAttention(nn.Module):
    def __init__(self, kernel_size=7):
        super().__init__()
        self.ch = AttentionGate(kernel_size)
        self.cw = AttentionGate(kernel_size)
        self.hw = AttentionGate(kernel_size)
        self.kernel_size = kernel_size

    def _affine(self):
        gates = (self.ch, self.cw, self.hw)
        folds = [F.bn_fold(g.conv.bn, g.conv.conv.bias) for g in gates]
        tag = tuple(id(t) for f in folds for t in f)

        def build():
            return torch.cat([torch.cat([s, t]) for s, t in folds]).contiguous()

        return F._derived_get((self,), ("triplet_affine",), tag, build), folds

    def forward(self, x):
        _eval_only(self)
        affine, _keep = self._affine()
        return F.triplet_forward(x, self.ch.conv.conv.weight, self.cw.conv.conv.weight, self.hw.conv.conv.weight, affine,
                                 self.kernel_size)


class ChannelGate(nn.Module):
    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.mlp = nn.Sequential(
            nn.Linear(channel, channel // reduction),
            nn.ReLU(inplace=True),
            nn.Linear(channel // reduction, channel),
        )
        self.bn = nn.BatchNorm1d(channel)

    def forward(self, x):
        """bn(mlp(avgpool(x))) expanded over the image (bam.py:28-33), computed by the kernels of the fused BAM forward."""
        _eval_only(self)
        Cr = self.mlp[0].weight.shape[0]
        bn1d = F.bn_fold(self.bn)
        dummy = x.new_zeros(1)
        params = [self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias, bn1d[0], bn1d[1]] + [dummy] * 10
        cg, _ = F.bam_gates(x, params, Cr, 1, channel=True, spatial=False)
        return cg.view(x.shape[0], x.shape[1], 1, 1).expand_as(x)


class SpatialGate(nn.Module):
    def __init__(self, channel, reduction=16, kernel_size=3, dilation_val=4):
        super().__init__()
        self.conv1 = nn.Conv2d(channel, channel // reduction, kernel_size=1)
        self.conv2 = nn.Sequential(
            nn.Conv2d(channel // reduction, channel // reduction, kernel_size, padding=dilation_val, dilation=dilation_val),
            nn.BatchNorm2d(channel // reduction),
            nn.ReLU(inplace=True),
            nn.Conv2d(channel // reduction, channel // reduction, kernel_size, padding=dilation_val, dilation=dilation_val),
            nn.BatchNorm2d(channel // reduction),
            nn.ReLU(inplace=True),
        )
        self.conv3 = nn.Conv2d(channel // reduction, 1, kernel_size=1)
        self.bn = nn.BatchNorm2d(1)
        self.kernel_size, self.dilation = kernel_size, dilation_val

    def _folded(self):
        """Parameters 6..15 of the BAM table: conv1, the two dilated convs with their BatchNorms folded, conv3 with bn folded."""
        if self.kernel_size != 3:
            raise ValueError("SpatialGate: the kernel is written for 3x3 dilated convolutions (bam.py:36)")
        d1 = F.bn_fold(self.conv2[1], self.conv2[0].bias)
        d2 = F.bn_fold(self.conv2[4], self.conv2[3].bias)
        s, t = F.bn_fold(self.bn, self.conv3.bias)
        tag = (id(s), self.conv3.weight._version, self.conv3.weight.data_ptr())
        w3, b3 = F._derived_get((self.conv3.weight, self.bn), ("bam_conv3",), tag,
                                lambda: ((self.conv3.weight.detach().reshape(-1) * s).contiguous(), t))
        return [self.conv1.weight, self.conv1.bias, self.conv2[0].weight, d1[0], d1[1], self.conv2[3].weight, d2[0], d2[1], w3, b3]

    def forward(self, x):
        """bn(conv3(conv2(conv1(x)))) expanded over the channels (bam.py:53-59), computed by the kernels of the fused BAM forward."""
        _eval_only(self)
        Cr = self.conv1.weight.shape[0]
        dummy = x.new_zeros(1)
        _, sg = F.bam_gates(x, [dummy] * 6 + self._folded(), Cr, self.dilation, channel=False, spatial=True)
        return sg.view(x.shape[0], 1, x.shape[2], x.shape[3]).expand_as(x)


class BAM(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.channel_attn = ChannelGate(channel)
        self.spatial_attn = SpatialGate(channel)

    def forward(self, x):
        _eval_only(self)
        ch, sp = self.channel_attn, self.spatial_attn
        Cr = sp.conv1.weight.shape[0]
        bn1d = F.bn_fold(ch.bn)
        params = [ch.mlp[0].weight, ch.mlp[0].bias, ch.mlp[2].weight, ch.mlp[2].bias, bn1d[0], bn1d[1]] + sp._folded()
        return F.bam_forward(x, params, Cr, sp.dilation)


class SKLayer(nn.Module):
    """attention_mechanisms/sk_module.py:17-56 (csrc/sk_dual.hip): both grouped branches in one kernel, branch softmax, blend."""

    def __init__(self, inplanes, planes, groups=32, ratio=16):
        super().__init__()
        d = max(planes // ratio, 32)
        self.planes = planes
        self.split_3x3 = nn.Sequential(
            nn.Conv2d(inplanes, planes, kernel_size=3, padding=1, groups=groups),
            nn.BatchNorm2d(planes),
            nn.ReLU(),
        )
        self.split_5x5 = nn.Sequential(
            nn.Conv2d(inplanes, planes, kernel_size=3, padding=2, dilation=2, groups=groups),
            nn.BatchNorm2d(planes),
            nn.ReLU(),
        )
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(
            nn.Linear(planes, d),
            nn.BatchNorm1d(d),
            nn.ReLU(),
        )
        self.fc1 = nn.Linear(d, planes)
        self.fc2 = nn.Linear(d, planes)
        self.groups, self.d = groups, d

    def forward(self, x):
        _eval_only(self)
        b3 = F.bn_fold(self.split_3x3[1], self.split_3x3[0].bias)
        b5 = F.bn_fold(self.split_5x5[1], self.split_5x5[0].bias)
        bf = F.bn_fold(self.fc[1])
        params = [self.split_3x3[0].weight, b3[0], b3[1], self.split_5x5[0].weight, b5[0], b5[1],
                  self.fc[0].weight, self.fc[0].bias, bf[0], bf[1], self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias]
        return F.sk_forward(x, params, self.planes, self.groups, self.d)


class PAM(nn.Module):
    """Position attention of DANet (dual_attention.py:10-28): the three 1x1 convs as ONE token-major GEMM, the streaming attention
    kernel with a single head of width `dim` and scale 1, and a transposing alpha * y + x epilogue.  The logits are unscaled dot
    products, so the default is the fp32-class precision mode (0).  dim <= 256 is one head; wider maps (DANet itself: 512) run as
    two channel halves whose logits are added through the kernel's bias input."""

    def __init__(self, dim):
        super().__init__()
        self.b = nn.Conv2d(dim, dim, 1)
        self.c = nn.Conv2d(dim, dim, 1)
        self.d = nn.Conv2d(dim, dim, 1)
        self.alpha = nn.Parameter(torch.zeros(1))
        self.precision = F.PREC_STRICT

    def _fused(self):
        ws = (self.b.weight, self.c.weight, self.d.weight, self.b.bias, self.c.bias, self.d.bias)
        tag = tuple((t._version, t.data_ptr()) for t in ws)

        def build():
            w = torch.cat([t.detach() for t in ws[:3]], dim=0).contiguous()
            return F._gemm_rows(w, 0), torch.cat([t.detach() for t in ws[3:]]).contiguous()

        return F._derived_get(ws, ("pam_bcd",), tag, build)

    def forward(self, x):
        n, c, h, w = x.shape
        rows, bias = self._fused()
        bcd, _ = F.conv2d_tokens(x, None, bias, 1, 1, 0, 0, precision=self.precision, wrows=rows)        # (n, hw, 3c)
        q, k, v = bcd[:, :, :c], bcd[:, :, c:2 * c], bcd[:, :, 2 * c:]
        if c <= F.SDPA_WIDTHS[-1]:
            y = F.sdpa_general(q, k, v, 1, 1.0, precision=self.precision)
        else:
            # dim above the kernel's widest head (DANet's own PAM is 512 wide): cut the channel axis in two, c = c0 + c1.  The logits
            # are a SUM over channels, q k^T = q0 k0^T + q1 k1^T, so half i attends with its own product on the matrix pipe and the
            # other half's logits (mi355_qk_logits_fwd, fp32) as the additive bias; its output is the i-th channel slice of y.
            c0 = self._split(c)
            y = torch.empty(n, h * w, c, dtype=torch.float32, device=x.device)
            for lo, hi, blo, bhi in ((0, c0, c0, c), (c0, c, 0, c0)):
                other = F.qk_logits(q[:, :, blo:bhi], k[:, :, blo:bhi], 1, precision=F.PREC_STRICT)        # (n, 1, hw, hw)
                F.sdpa_general(q[:, :, lo:hi], k[:, :, lo:hi], v[:, :, lo:hi], 1, 1.0, bias=other, precision=self.precision, out=y[:, :, lo:hi])
        return F.tokens_to_nchw_axpy(y, x, self.alpha)

    @staticmethod
    def _split(c):
        for c0 in sorted(F.SDPA_WIDTHS, reverse=True):
            if c - c0 in F.SDPA_WIDTHS:
                return c0
        raise NotImplementedError(f"PAM: dim {c} is not a sum of two head widths the attention kernel is built for {F.SDPA_WIDTHS}")


class CAM(nn.Module):
    """Channel attention of DANet (dual_attention.py:30-42)."""

    def __init__(self):
        super().__init__()
        self.beta = nn.Parameter(torch.zeros(1))
        self.precision = None          # operand format of attn @ x; the Gram logits always run in the fp32-class mode

    def forward(self, x):
        return F.cam_forward(x, self.beta, self.precision)
