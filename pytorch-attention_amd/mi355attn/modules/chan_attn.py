"""Drop-in nn.Modules for the channel / spatial attention family, forward routed to libmi355attn.

Class names, constructor signatures and state_dict layouts mirror the reference so that
``mine.load_state_dict(ref.state_dict())`` works unchanged and so that, under the same RNG seed, the
parameters come out bit-identical (same sub-module creation order => same init stream):

  SELayer            attention_mechanisms/se_module.py:18-33      keys fc.0.weight, fc.2.weight
  ECALayer           attention_mechanisms/eca.py:17-30            key  conv.weight
  ChannelAttention   attention_mechanisms/cbam.py:19-35           keys fc.0.weight, fc.2.weight (1x1 convs)
  SpatialAttention   attention_mechanisms/cbam.py:37-48           key  conv.weight
  CBAM               attention_mechanisms/cbam.py:50-59           keys ca.*, sa.*
  DoubleAttention    attention_mechanisms/double_attention.py:19-48

The sub-modules below are parameter containers only; ``forward`` never calls them.
"""
import math

from torch import nn

from .. import functional as F


def _excite_stack(first, second):
    """Index-compatible container: weights live at positions 0 and 2 like the reference's Sequential."""
    return nn.Sequential(first, nn.ReLU(inplace=True), second)


class SELayer(nn.Module):
    """Squeeze-and-Excitation: y = x * sigmoid(W2 relu(W1 mean_hw(x))), one fused HIP pass pair."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        hidden = channel // reduction
        self.fc = _excite_stack(nn.Linear(channel, hidden, bias=False), nn.Linear(hidden, channel, bias=False))
        self.fc.append(nn.Sigmoid())

    def forward(self, x):
        return F.se_forward(x, self.fc[0].weight, self.fc[2].weight)


class ECALayer(nn.Module):
    """Efficient Channel Attention: k-tap conv across the pooled channel vector, k from the reference rule."""

    def __init__(self, channels, gamma=2, b=1):
        super().__init__()
        t = int(abs((math.log(channels, 2) + b) / gamma))
        k = t + (1 - t % 2)                      # next odd >= t   (eca.py:21-22)
        self.conv = nn.Conv1d(1, 1, kernel_size=k, padding=(k - 1) // 2, bias=False)

    def forward(self, x):
        return F.eca_forward(x, self.conv.weight)


class ChannelAttention(nn.Module):
    def __init__(self, channel, reduction=16):
        super().__init__()
        hidden = channel // reduction
        self.fc = _excite_stack(nn.Conv2d(channel, hidden, 1, bias=False), nn.Conv2d(hidden, channel, 1, bias=False))

    def forward(self, x):
        return F.cbam_forward(x, self.fc[0].weight, self.fc[2].weight, None, stage=1)


class SpatialAttention(nn.Module):
    def __init__(self, kernel_size=7):
        super().__init__()
        self.conv = nn.Conv2d(2, 1, kernel_size, padding=kernel_size // 2, bias=False)

    def forward(self, x):
        return F.cbam_forward(x, None, None, self.conv.weight, stage=2)


class CBAM(nn.Module):
    """Channel stage then spatial stage, fused into one launcher call (5 kernels, 3 reads + 1 write of x)."""

    def __init__(self, channel, reduction=16, kernel_size=7):
        super().__init__()
        self.ca = ChannelAttention(channel, reduction)
        self.sa = SpatialAttention(kernel_size)

    def forward(self, x):
        return F.cbam_forward(x, self.ca.fc[0].weight, self.ca.fc[2].weight, self.sa.conv.weight, stage=0)


class DoubleAttention(nn.Module):
    """A2-Net double attention block; `precision` selects the MFMA operand format (None = package default)."""

    def __init__(self, in_channels, c_m, c_n, precision=None):
        super().__init__()
        self.c_m, self.c_n, self.in_channels = c_m, c_n, in_channels
        self.convA = nn.Conv2d(in_channels, c_m, kernel_size=1)
        self.convB = nn.Conv2d(in_channels, c_n, kernel_size=1)
        self.convV = nn.Conv2d(in_channels, c_n, kernel_size=1)
        self.proj = nn.Conv2d(c_m, in_channels, kernel_size=1)
        self.precision = precision

    def forward(self, x):
        return F.double_attention_forward(x, self.convA.weight, self.convA.bias, self.convB.weight, self.convB.bias,
                                          self.convV.weight, self.convV.bias, self.proj.weight, self.proj.bias,
                                          precision=self.precision)
