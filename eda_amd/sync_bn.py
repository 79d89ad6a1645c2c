"""SyncBatchNorm-equivalent statistics exchange for the channels-last BN+ReLU sites of the path.

The reference converts every BatchNorm to ``SyncBatchNorm`` when more than one GPU trains
(main_utils.py:336-338): at N > 1 its batch statistics cover the GLOBAL batch.  This repo's default
at N > 1 is per-GPU statistics (8 scenes x >= 2048 positions per channel already);
``enable(group)`` switches every BN+ReLU site (set-abstraction / feature-propagation MLPs, heads,
positional embeddings) to the reference's semantics:

* forward: ONE all-reduce of the packed vector [sum z | sum z^2 | count] (2C+1 doubles) per BN layer
  (torch's SyncBatchNorm all-gathers mean / invstd / count: same information), then mean, biased
  variance, running-statistics update with the unbiased GLOBAL variance;
* backward: ONE all-reduce of [sum gy | sum gy*xhat] (2C doubles), then
  dz = gamma*rstd*(gy - s1/N - xhat*s2/N) with the global N; d(gamma), d(beta) stay LOCAL sums (the
  gradient all-reduce of eda_amd/parallel.py adds them over ranks like every other gradient).

It is written with device-agnostic torch ops (it runs under gloo on CPU in tests/test_parallel_cpu.py
and under RCCL on the GPUs); the fused single-launch kernels cannot stop in the middle for a
collective, so this mode trades their fusion for the reference's exact multi-GPU statistics.
"""
import torch
import torch.distributed as dist
from torch.autograd import Function

_group = None
_enabled = False


def enable(group=None):
    """Use global-batch statistics in every BN+ReLU site (requires an initialised process group)."""
    global _enabled, _group
    if not dist.is_initialized():
        raise RuntimeError("sync_bn.enable() needs torch.distributed to be initialised")
    _enabled, _group = True, group


def disable():
    global _enabled, _group
    _enabled, _group = False, None


def enabled():
    return _enabled and dist.is_initialized() and dist.get_world_size(_group) > 1


class _SyncBNReLU(Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, eps, momentum, training):
        R, C = z.shape
        if training:
            zd = z.double()
            packed = torch.cat([zd.sum(0), (zd * zd).sum(0), torch.full((1,), float(R), dtype=torch.float64, device=z.device)])
            dist.all_reduce(packed, group=_group)
            n = packed[2 * C]
            mean = packed[:C] / n
            var = (packed[C:2 * C] / n - mean * mean).clamp_min(0.0)
            if running_mean is not None:
                unbiased = var * (n / (n - 1.0)) if float(n) > 1 else var
                running_mean.mul_(1.0 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
                running_var.mul_(1.0 - momentum).add_(unbiased.to(running_var.dtype), alpha=momentum)
            mean, var = mean.float(), var.float()
        else:
            mean, var = running_mean, running_var
            n = torch.tensor(float(R), dtype=torch.float64, device=z.device)
        rstd = torch.rsqrt(var + eps)
        xhat = (z - mean) * rstd
        y = torch.relu(xhat * gamma + beta)
        ctx.save_for_backward(xhat, y, gamma, rstd, n)
        ctx.training = bool(training)
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, y, gamma, rstd, n = ctx.saved_tensors
        gy = dy * (y > 0)
        s1 = gy.sum(0, dtype=torch.float64)
        s2 = (gy * xhat).sum(0, dtype=torch.float64)
        dgamma, dbeta = s2.float(), s1.float()
        if ctx.training:
            packed = torch.cat([s1, s2])
            dist.all_reduce(packed, group=_group)
            C = s1.numel()
            m1 = (packed[:C] / n).float()
            m2 = (packed[C:] / n).float()
            dz = (gy - m1 - xhat * m2) * (gamma * rstd)
        else:
            dz = gy * (gamma * rstd)
        return dz, dgamma, dbeta, None, None, None, None, None


def bn_relu(bn, z, pool=1):
    """relu(SyncBatchNorm `bn`(z)) on rows z (R, C), optionally max-pooled over `pool` consecutive rows."""
    training = bn.training or not bn.track_running_stats
    y = _SyncBNReLU.apply(z, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                          bn.running_var if bn.track_running_stats else None, bn.eps, bn.momentum, training)
    if pool > 1:
        R, C = y.shape
        y = y.view(R // pool, pool, C).max(dim=1)[0]
    return y
