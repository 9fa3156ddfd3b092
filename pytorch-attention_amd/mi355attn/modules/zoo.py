"""Drop-in modules for five more members of the reference's channel-attention zoo: per-channel statistics, a tiny
transform and a broadcast scale, x read once (csrc/chan_stat.hip).

  simam_module  attention_mechanisms/simam.py:17-41     parameter-free energy-based gate (per element)
  SRM           attention_mechanisms/srm.py:16-34       style pooling (mean, std) -> channel-wise fc -> BatchNorm1d (eval) -> sigmoid
  GaussianGCT   attention_mechanisms/gct.py:15-30       (class GCT there) Gaussian context transformer, parameter-free
  LCT           attention_mechanisms/lct.py:17-39       group-normalised channel means -> affine -> sigmoid
  GCT           attention_mechanisms/gate_channel_module.py:18-50   gated channel transformation (l2 / l1 embedding)
"""
import torch
from torch import nn

from .. import functional as F


class simam_module(nn.Module):
    def __init__(self, channels=None, e_lambda=1e-4):
        super().__init__()
        self.activaton = nn.Sigmoid()          # attribute name as in the reference (sic)
        self.e_lambda = e_lambda

    def __repr__(self):
        return self.__class__.__name__ + "(" + ("lambda=%f)" % self.e_lambda)

    @staticmethod
    def get_module_name():
        return "simam"

    def forward(self, x):
        return F.simam_forward(x, self.e_lambda)


class SRM(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.cfc = nn.Conv1d(channel, channel, kernel_size=2, groups=channel, bias=False)
        self.bn = nn.BatchNorm1d(channel)

    def forward(self, x):
        if self.training:
            raise RuntimeError("inference engine: BatchNorm runs with its running statistics; call .eval()")
        bn = self.bn
        return F.srm_forward(x, self.cfc.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


class GaussianGCT(nn.Module):
    """`GCT` of attention_mechanisms/gct.py (the gate_channel_module.py class of the same name is `GCT` below)."""

    def __init__(self, channels, c=2, eps=1e-5):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.eps = eps
        self.c = c

    def forward(self, x):
        return F.gct_gauss_forward(x, self.c, self.eps)


class LCT(nn.Module):
    def __init__(self, channels, groups, eps=1e-5):
        super().__init__()
        assert channels % groups == 0, "Number of channels should be evenly divisible by the number of groups"
        self.groups = groups
        self.channels = channels
        self.eps = eps
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.w = nn.Parameter(torch.ones(channels))
        self.b = nn.Parameter(torch.zeros(channels))
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        return F.lct_forward(x, self.w, self.b, self.groups, self.eps)


class GCT(nn.Module):
    def __init__(self, num_channels, epsilon=1e-5, mode='l2', after_relu=False):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1, num_channels, 1, 1))
        self.gamma = nn.Parameter(torch.zeros(1, num_channels, 1, 1))
        self.beta = nn.Parameter(torch.zeros(1, num_channels, 1, 1))
        self.epsilon = epsilon
        self.mode = mode
        self.after_relu = after_relu

    def forward(self, x):
        return F.gct_forward(x, self.alpha, self.gamma, self.beta, self.epsilon, self.mode, self.after_relu)
