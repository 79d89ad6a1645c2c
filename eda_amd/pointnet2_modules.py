"""Set-abstraction and feature-propagation modules of the PointNet++ backbone.

Mirrors the two classes EDA uses from pointnet2/pointnet2_modules.py:
PointnetSAModuleVotes (:164-272) and PointnetFPModule (:356-416), with the same
constructor keywords, return values and sub-module names (``mlp_module``,
``grouper``, ``mlp``).  The MSG / LFP variants of the reference are unused by EDA
and are not part of this path.
"""
from typing import List

import torch
from torch import nn
import torch.nn.functional as F

from . import pointnet2_utils, sa_ops
from .pytorch_utils import SharedMLP


class PointnetSAModuleVotes(nn.Module):
    """FPS -> gather centres -> ball query + group -> shared MLP -> pool."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True, pooling: str = "max",
                 sigma: float = None, normalize_xyz: bool = False, sample_uniformly: bool = False,
                 ret_unique_cnt: bool = False):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.pooling = pooling
        self.use_xyz = use_xyz
        self.sigma = sigma if sigma is not None else (radius / 2 if radius is not None else None)
        self.normalize_xyz = normalize_xyz
        if ret_unique_cnt:
            raise NotImplementedError("ret_unique_cnt is unused by EDA")
        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(
                radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                normalize_xyz=normalize_xyz, sample_uniformly=sample_uniformly)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)
        spec = list(mlp)          # (the reference mutates the caller's list, :204-206)
        if use_xyz and len(spec) > 0:
            spec[0] += 3
        self.mlp_module = SharedMLP(spec, bn=bn)

    def geometry(self, xyz, inds=None, xyz_in_sampling_order=False):
        """(inds, new_xyz, ball-query idx): everything this module derives from the coordinates alone (no gradient
        flows through any of it).  A caller may compute it ahead of the forward -- for the NEXT batch on a second
        stream -- and pass it back through `forward(..., geometry=)`."""
        with torch.no_grad():
            if inds is None:
                inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint, xyz_in_sampling_order)
            new_xyz = pointnet2_utils.gather_operation(
                xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
            idx = pointnet2_utils.ball_query(self.radius, self.nsample, xyz, new_xyz)
        return inds, new_xyz, idx

    def forward(self, xyz, features=None, inds=None, xyz_in_sampling_order=False, geometry=None):
        if geometry is not None and self._fast_path(xyz):
            inds, new_xyz, idx = geometry
            assert inds.shape[1] == self.npoint and idx.shape[1:] == (self.npoint, self.nsample)
            return new_xyz, self._forward_rows(xyz, new_xyz, features, idx=idx), inds
        if inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint, xyz_in_sampling_order)
        else:
            assert inds.shape[1] == self.npoint
        new_xyz = None
        if self.npoint is not None:
            new_xyz = pointnet2_utils.gather_operation(
                xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        if self._fast_path(xyz):
            return new_xyz, self._forward_rows(xyz, new_xyz, features), inds
        out = self.grouper(xyz, new_xyz, features)
        grouped, grouped_xyz = out if isinstance(out, tuple) else (out, None)
        x = self.mlp_module(grouped)                       # (B, C, npoint, nsample)
        if self.pooling == "max":
            x = x.max(dim=3)[0]
        elif self.pooling == "avg":
            x = x.mean(dim=3)
        elif self.pooling == "rbf":
            rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1) / (self.sigma ** 2) / 2)
            x = torch.sum(x * rbf.unsqueeze(1), -1) / float(self.nsample)
        else:
            raise ValueError(self.pooling)
        return new_xyz, x, inds


    # ---- channels-last fast path (csrc/sa_cl.hip) ------------------------------
    def _fast_path(self, xyz):
        layers = self.mlp_module.layers()
        return (xyz.is_cuda and self.npoint is not None and self.pooling == "max" and self.use_xyz
                and self.nsample <= 255 and all(l.bn is not None for l in layers)
                and all(l.conv.out_channels % 4 == 0 for l in layers))

    def _forward_rows(self, xyz, new_xyz, features, idx=None):
        """ball query -> [centred xyz | gathered features] rows -> 3 x (GEMM, fused BN+ReLU)
        -> max over the nsample rows of each centre.  Same math as the generic path."""
        B, m = new_xyz.shape[0], new_xyz.shape[1]
        if idx is None:
            idx = pointnet2_utils.ball_query(self.radius, self.nsample, xyz, new_xyz)
        feats_cl = features.transpose(1, 2).contiguous() if features is not None else None
        if sa_ops._fusable(self.mlp_module.layers()):
            # ball query -> ONE fused native call: neighbourhood rows gathered into LDS, three MFMA
            # GEMMs with BN statistics / BN+ReLU folded in, max-pool (csrc/sa_cl.hip, gemm.hip)
            pooled = sa_ops.fused_mlp(self.mlp_module, self.nsample, xyz=xyz, new_xyz=new_xyz, feats_cl=feats_cl,
                                      idx=idx, radius=self.radius, normalize_xyz=self.normalize_xyz)
            return pooled.view(B, m, -1).transpose(1, 2)       # (B, C, m) VIEW of the channels-last result: the next
            #                                                    module's transpose(1, 2).contiguous() is then free
        rows = sa_ops.GroupConcatCL.apply(xyz, new_xyz, feats_cl, idx, self.radius, self.normalize_xyz)
        pooled = sa_ops.shared_mlp_rows(self.mlp_module, rows.view(B * m * self.nsample, -1), self.nsample)
        return pooled.view(B, m, -1).transpose(1, 2)


class PointnetFPModule(nn.Module):
    """3-NN inverse-distance interpolation -> concat skip -> shared MLP."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = SharedMLP(list(mlp), bn=bn)

    @staticmethod
    def geometry(unknown, known):
        """(idx, weight) of the 3-NN inverse-distance interpolation: a function of the coordinates alone."""
        with torch.no_grad():
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
        return idx, weight

    def forward(self, unknown, known, unknow_feats, known_feats, geometry=None):
        if known is not None:
            if geometry is not None:
                idx, weight = geometry
            else:
                dist, idx = pointnet2_utils.three_nn(unknown, known)
                dist_recip = 1.0 / (dist + 1e-8)
                weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats.contiguous(), idx, weight)
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        layers = self.mlp.layers()
        if (interpolated.is_cuda and all(l.bn is not None for l in layers)
                and all(l.conv.out_channels % 4 == 0 for l in layers)):
            # channels-last rows, fused BN+ReLU.  The concatenation is written in rows layout directly (the skip
            # features usually ARE a transposed view of channels-last rows: no copy on that side)
            parts = [interpolated.transpose(1, 2)] + ([unknow_feats.transpose(1, 2)] if unknow_feats is not None else [])
            x_rows = torch.cat(parts, dim=2) if len(parts) > 1 else parts[0].contiguous()
            B, n, C = x_rows.shape
            rows = sa_ops.shared_mlp_rows(self.mlp, x_rows.view(B * n, C), 1)
            return rows.view(B, n, -1).transpose(1, 2)
        x = torch.cat([interpolated, unknow_feats], dim=1) if unknow_feats is not None else interpolated
        return self.mlp(x.unsqueeze(-1)).squeeze(-1)
