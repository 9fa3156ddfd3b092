"""Drop-in modules for the squeeze-excite copies inside the reference's CNN files, all on the SE kernels with their
bias / gate options (mi355_se_ex_fwd).  Module level only: the surrounding networks are convolution stacks and are not mirrored.

  SELayerBias      cnns/efficientnet.py:13-28, cnns/mnasnet.py:11-26   Linear layers WITH bias, ReLU, sigmoid
  SELayerBias4     cnns/efficientnetv2.py:14-29                         the same, default ratio 4
  SELayerHidden    cnns/mobilenetv3.py:15-30                            explicit hidden width, no bias
  SqueezeExcite    cnns/ghostnet.py:48-65                               1x1 convs with bias, ReLU, hard-sigmoid gate
(vision_transformers/moat.py:18-33 is the plain bias-free SELayer: attention_mechanisms.se_module.SELayer serves it.)
"""
import torch
from torch import nn

from .. import functional as F


class SELayerBias(nn.Module):
    def __init__(self, channels, ratio=16):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(channels, channels // ratio), nn.ReLU(), nn.Linear(channels // ratio, channels), nn.Sigmoid())

    def forward(self, x):
        return F.se_ex_forward(x, self.fc[0].weight, self.fc[0].bias, self.fc[2].weight, self.fc[2].bias)


class SELayerBias4(SELayerBias):
    """cnns/efficientnetv2.py:14-29: the same module with a default reduction ratio of 4."""

    def __init__(self, channels, ratio=4):
        super().__init__(channels, ratio)


class SELayerHidden(nn.Module):
    def __init__(self, in_channel, hidden_channel):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(in_channel, hidden_channel, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(hidden_channel, in_channel, bias=False), nn.Sigmoid())

    def forward(self, x):
        return F.se_forward(x, self.fc[0].weight, self.fc[2].weight)


def _make_divisible(v, divisor, min_value=None):
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def hard_sigmoid(x, inplace: bool = False):
    return torch.nn.functional.relu6(x + 3.) / 6.


class SqueezeExcite(nn.Module):
    def __init__(self, in_chs, se_ratio=0.25, reduced_base_chs=None, act_layer=nn.ReLU, gate_fn=hard_sigmoid, divisor=4, **_):
        super().__init__()
        if act_layer is not nn.ReLU or gate_fn is not hard_sigmoid:
            raise NotImplementedError("built for ReLU + hard_sigmoid (the reference's defaults)")
        self.gate_fn = gate_fn
        reduced_chs = _make_divisible((reduced_base_chs or in_chs) * se_ratio, divisor)
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_reduce = nn.Conv2d(in_chs, reduced_chs, 1, bias=True)
        self.act1 = act_layer(inplace=True)
        self.conv_expand = nn.Conv2d(reduced_chs, in_chs, 1, bias=True)

    def forward(self, x):
        return F.se_ex_forward(x, self.conv_reduce.weight, self.conv_reduce.bias, self.conv_expand.weight, self.conv_expand.bias,
                               gate="hard_sigmoid")
