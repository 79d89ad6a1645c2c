"""Pointwise (kernel-size-1) convolutions as plain GEMMs.

Every convolution on the hot path is 1x1 (SharedMLP, the conv1d heads, the
positional embeddings).  ``nn.Conv1d/Conv2d`` would send them through MIOpen's
solver search (tens of seconds of warm-up "find" kernels and NCHW direct-conv
solvers); a 1x1 convolution is just W (Cout,Cin) @ X (B,Cin,M), so these
subclasses keep the parameter names and shapes of the torch modules (checkpoint
contract) and run one batched GEMM instead.
"""
import torch
from torch import nn
import torch.nn.functional as F


class Conv1dK1(nn.Conv1d):
    def __init__(self, cin, cout, kernel_size=1, bias=True):
        assert kernel_size == 1
        super().__init__(cin, cout, 1, bias=bias)

    def rows(self, x):
        """Channels-last form: x (..., Cin) -> (..., Cout), one row-major GEMM, no transposes."""
        return F.linear(x, self.weight.squeeze(-1), self.bias)

    def forward(self, x):                                  # (B, Cin, M)
        y = torch.matmul(self.weight.squeeze(-1), x)
        return y if self.bias is None else y + self.bias[:, None]


class Conv2dK1(nn.Conv2d):
    def __init__(self, cin, cout, kernel_size=(1, 1), bias=True):
        assert tuple(kernel_size) == (1, 1)
        super().__init__(cin, cout, (1, 1), bias=bias)

    def forward(self, x):                                  # (B, Cin, H, W)
        B, C, H, W = x.shape
        y = torch.matmul(self.weight.view(self.out_channels, C), x.reshape(B, C, H * W))
        if self.bias is not None:
            y = y + self.bias[:, None]
        return y.view(B, self.out_channels, H, W)


def bn_relu_rows(bn, z):
    """relu(BatchNorm1d/2d `bn`(z)) for channels-last rows z (R, C) on the GPU: the fused
    HIP kernel of csrc/sa_cl.hip (batch statistics + running-stat update in training)."""
    from . import sa_ops
    out = sa_ops.BNReLUCL.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                bn.momentum, bn.training, 1)
    if bn.training and bn.track_running_stats:
        bn.num_batches_tracked.add_(1)
    return out


def rows_ok(x, *channels):
    """The rows fast path needs the HIP library (GPU tensors) and channel counts % 4 == 0."""
    return x.is_cuda and all(c % 4 == 0 for c in channels)
