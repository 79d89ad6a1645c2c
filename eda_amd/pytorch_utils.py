"""SharedMLP -- the "grouped MLP" of the set-abstraction layers.

Mirrors pointnet2/pytorch_utils.py:11-36,67-120 as used by EDA: a stack of
(1x1 conv without bias -> BatchNorm2d -> ReLU).  Sub-module names are part of the
checkpoint contract: ``layer{i}.conv.weight (Cout,Cin,1,1)`` and
``layer{i}.bn.bn.{weight,bias,running_mean,running_var,num_batches_tracked}``.
"""
import torch
from torch import nn
import torch.nn.functional as F

from .nn_utils import Conv2dK1


class _BNHolder(nn.Module):
    """Gives BatchNorm2d the reference's ``bn.bn`` key prefix (pytorch_utils.py:39-45)."""

    def __init__(self, channels):
        super().__init__()
        self.bn = nn.BatchNorm2d(channels)      # affine init weight=1, bias=0 (torch default)

    def forward(self, x):
        return self.bn(x)


class PointwiseConvBNReLU(nn.Module):
    def __init__(self, cin, cout, bn=True):
        super().__init__()
        # bias is dropped when followed by BN (pytorch_utils.py:87)
        self.conv = Conv2dK1(cin, cout, kernel_size=(1, 1), bias=not bn)
        nn.init.kaiming_normal_(self.conv.weight)
        if not bn:
            nn.init.constant_(self.conv.bias, 0)
        self.bn = _BNHolder(cout) if bn else None

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return F.relu(x, inplace=True)


class SharedMLP(nn.Module):
    def __init__(self, channels, *, bn=False):
        super().__init__()
        self.n_layers = len(channels) - 1
        for i in range(self.n_layers):
            self.add_module(f"layer{i}", PointwiseConvBNReLU(channels[i], channels[i + 1], bn=bn))

    def layers(self):
        return [getattr(self, f"layer{i}") for i in range(self.n_layers)]

    def forward(self, x):
        for layer in self.layers():
            x = layer(x)
        return x
