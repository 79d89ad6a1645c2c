"""Small heads around the encoder/decoder: seed objectness, query sampling,
box prediction.  Mirrors models/modules.py:19-178 (names = checkpoint contract).
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .nn_utils import Conv1dK1, bn_relu_rows, rows_ok
from .pointnet2_utils import gather_operation
from .encoder_decoder_layers import PositionEmbeddingLearned  # noqa: F401  (re-exported like the reference)


_GROUPED_HEADS = os.environ.get("EDA_GROUPED_HEADS", "1") != "0"


class PointsObjClsModule(nn.Module):
    """Per-seed objectness logit: (B, C, K) -> (B, 1, K)."""

    def __init__(self, seed_feature_dim):
        super().__init__()
        self.in_dim = seed_feature_dim
        self.conv1 = Conv1dK1(self.in_dim, self.in_dim, 1)
        self.bn1 = nn.BatchNorm1d(self.in_dim)
        self.conv2 = Conv1dK1(self.in_dim, self.in_dim, 1)
        self.bn2 = nn.BatchNorm1d(self.in_dim)
        self.conv3 = Conv1dK1(self.in_dim, 1, 1)

    def forward(self, seed_features, seed_rows=None):
        """seed_features (B, C, K) -> logits (B, 1, K).  seed_rows: the same features as
        (B, K, C) rows when the caller already has them (skips a transpose)."""
        if rows_ok(seed_features, self.in_dim):
            B, C, K = seed_features.shape
            rows = seed_rows if seed_rows is not None else seed_features.transpose(1, 2)
            net = bn_relu_rows(self.bn1, self.conv1.rows(rows.reshape(B * K, C)))
            net = bn_relu_rows(self.bn2, self.conv2.rows(net))
            return self.conv3.rows(net).view(B, 1, K)      # (B*K, 1) rows == (B, 1, K)
        net = F.relu(self.bn1(self.conv1(seed_features)))
        net = F.relu(self.bn2(self.conv2(net)))
        return self.conv3(net)


class GeneralSamplingModule(nn.Module):
    """Gather xyz (B,K,3) and features (B,C,K) at sample_inds (B,Q) int32."""

    def forward(self, xyz, features, sample_inds, features_rows=None):
        """features_rows: the same features channels-last (B,K,C), when the caller has them: the sampled rows are then
        gathered there and `new_features` is a (B,C,Q) view of them (no channels-first copy on either side)."""
        new_xyz = gather_operation(xyz.transpose(1, 2).contiguous(), sample_inds).transpose(1, 2).contiguous()
        if features_rows is not None and features_rows.is_cuda:
            C = features_rows.shape[-1]
            rows = torch.gather(features_rows, 1, sample_inds.long().unsqueeze(-1).expand(-1, -1, C))
            return new_xyz, rows.transpose(1, 2), sample_inds
        new_features = gather_operation(features.contiguous(), sample_inds).contiguous()
        return new_xyz, new_features, sample_inds


class ThreeLayerMLP(nn.Module):
    def __init__(self, dim, out_dim):
        super().__init__()
        self.net = nn.Sequential(
            Conv1dK1(dim, dim, 1, bias=False), nn.BatchNorm1d(dim), nn.ReLU(), nn.Dropout(0.3),
            Conv1dK1(dim, dim, 1, bias=False), nn.BatchNorm1d(dim), nn.ReLU(), nn.Dropout(0.3),
            Conv1dK1(dim, out_dim, 1))

    def forward(self, x):
        return self.net(x)

    def rows(self, x):
        """x (R, dim) rows -> (R, out_dim): GEMM -> fused BN+ReLU -> dropout, twice, then GEMM."""
        n = self.net
        h = bn_relu_rows(n[1], n[0].rows(x), dropout=n[3])
        h = bn_relu_rows(n[5], n[4].rows(h), dropout=n[7])
        return n[8].rows(h)


class ClsAgnosticPredictHead(nn.Module):
    """Box centre residual, size and token-distribution logits per query."""

    def __init__(self, num_class, num_heading_bin, num_proposal, seed_feat_dim=256,
                 objectness=True, heading=False, compute_sem_scores=True):
        super().__init__()
        self.num_class, self.num_heading_bin, self.num_proposal = num_class, num_heading_bin, num_proposal
        self.seed_feat_dim = seed_feat_dim
        self.objectness, self.heading, self.compute_sem_scores = objectness, heading, compute_sem_scores
        if objectness:
            self.objectness_scores_head = ThreeLayerMLP(seed_feat_dim, 1)
        self.center_residual_head = ThreeLayerMLP(seed_feat_dim, 3)
        if heading:
            self.heading_class_head = Conv1dK1(seed_feat_dim, num_heading_bin, 1)
            self.heading_residual_head = Conv1dK1(seed_feat_dim, num_heading_bin, 1)
        self.size_pred_head = ThreeLayerMLP(seed_feat_dim, 3)
        if compute_sem_scores:
            self.sem_cls_scores_head = ThreeLayerMLP(seed_feat_dim, self.num_class)

    def forward(self, features, base_xyz, end_points, prefix="", features_rows=None):
        """features (B, C, Q) as in the reference; features_rows: the same as (B, Q, C) rows when
        the caller has them (the decoder output already is), which skips the transposes."""
        if rows_ok(features, self.seed_feat_dim):
            return self._forward_rows(features, base_xyz, end_points, prefix, features_rows)
        B, _, Q = features.shape
        if self.objectness:
            end_points[f"{prefix}objectness_scores"] = \
                self.objectness_scores_head(features).transpose(2, 1).squeeze(-1)
        center = base_xyz + self.center_residual_head(features).transpose(2, 1)
        if self.heading:
            hs = self.heading_class_head(features).transpose(2, 1)
            hr = self.heading_residual_head(features).transpose(2, 1)
            end_points[f"{prefix}heading_scores"] = hs
            end_points[f"{prefix}heading_residuals_normalized"] = hr
            end_points[f"{prefix}heading_residuals"] = hr * (np.pi / self.num_heading_bin)
        pred_size = self.size_pred_head(features).transpose(2, 1).view(B, Q, 3)
        end_points[f"{prefix}base_xyz"] = base_xyz
        end_points[f"{prefix}center"] = center
        end_points[f"{prefix}pred_size"] = pred_size
        if self.compute_sem_scores:
            end_points[f"{prefix}sem_cls_scores"] = self.sem_cls_scores_head(features).transpose(2, 1)
        return center, pred_size

    def sibling_stacks(self):
        """Names of the ThreeLayerMLP sub-stacks that run side by side on the same features (FlatParams
        lays their same-named parameters out back to back, see eda_amd/parallel.py)."""
        names = ["objectness_scores_head"] if self.objectness else []
        names += ["center_residual_head", "size_pred_head"]
        if self.compute_sem_scores:
            names.append("sem_cls_scores_head")
        return names

    def _grouped_ok(self, rows):
        from . import _lib, sync_bn
        nets = [getattr(self, n).net for n in self.sibling_stacks()]
        bn0 = nets[0][1]
        return (_GROUPED_HEADS and not sync_bn.diverts() and rows.is_cuda and rows.shape[0] <= _lib.lib().eda_bn_relu_dropout_max_rows()
                and self.seed_feat_dim % 16 == 0 and 2 <= len(nets) <= 4
                and all(n[1].training == bn0.training and n[5].training == bn0.training for n in nets)
                # the grouped launches take eps / momentum / running-statistics mode / dropout state from sibling 0
                and all(bn.eps == bn0.eps and bn.momentum == bn0.momentum
                        and bn.track_running_stats == bn0.track_running_stats for n in nets for bn in (n[1], n[5]))
                and all(n[3].training == nets[0][3].training and n[7].training == nets[0][3].training for n in nets)
                and all(n[3].p == nets[0][3].p and n[7].p == nets[0][3].p for n in nets))

    def _sibling_mlps_rows(self, rows):
        """All ThreeLayerMLP stacks of this head on rows (R, C): layer l of every stack in ONE launch
        (eda_amd/grouped.py) -- 5 launches instead of 5 per stack."""
        from . import grouped
        nets = [getattr(self, n).net for n in self.sibling_stacks()]
        z1 = grouped.shared_in_linear(rows, [n[0].weight.squeeze(-1) for n in nets])
        a1 = grouped.grouped_bn_relu(z1, [n[1] for n in nets], [n[3] for n in nets])
        z2 = grouped.block_linear(a1, [n[4].weight.squeeze(-1) for n in nets], [None] * len(nets), pack_out=True)
        a2 = grouped.grouped_bn_relu(z2, [n[5] for n in nets], [n[7] for n in nets])
        outs = grouped.block_linear(a2, [n[8].weight.squeeze(-1) for n in nets], [n[8].bias for n in nets],
                                    pack_out=False)
        return dict(zip(self.sibling_stacks(), outs))

    def _forward_rows(self, features, base_xyz, end_points, prefix, features_rows):
        B, C, Q = features.shape
        rows = (features_rows if features_rows is not None else features.transpose(1, 2)).reshape(B * Q, C)
        if not self.heading and self._grouped_ok(rows):
            o = self._sibling_mlps_rows(rows)
            if self.objectness:
                end_points[f"{prefix}objectness_scores"] = o["objectness_scores_head"].view(B, Q)
            center = base_xyz + o["center_residual_head"].view(B, Q, 3)
            pred_size = o["size_pred_head"].view(B, Q, 3)
            end_points[f"{prefix}base_xyz"] = base_xyz
            end_points[f"{prefix}center"] = center
            end_points[f"{prefix}pred_size"] = pred_size
            if self.compute_sem_scores:
                end_points[f"{prefix}sem_cls_scores"] = o["sem_cls_scores_head"].view(B, Q, -1)
            return center, pred_size
        if self.objectness:
            end_points[f"{prefix}objectness_scores"] = self.objectness_scores_head.rows(rows).view(B, Q)
        center = base_xyz + self.center_residual_head.rows(rows).view(B, Q, 3)
        if self.heading:
            hs = self.heading_class_head.rows(rows).view(B, Q, -1)
            hr = self.heading_residual_head.rows(rows).view(B, Q, -1)
            end_points[f"{prefix}heading_scores"] = hs
            end_points[f"{prefix}heading_residuals_normalized"] = hr
            end_points[f"{prefix}heading_residuals"] = hr * (np.pi / self.num_heading_bin)
        pred_size = self.size_pred_head.rows(rows).view(B, Q, 3)
        end_points[f"{prefix}base_xyz"] = base_xyz
        end_points[f"{prefix}center"] = center
        end_points[f"{prefix}pred_size"] = pred_size
        if self.compute_sem_scores:
            end_points[f"{prefix}sem_cls_scores"] = self.sem_cls_scores_head.rows(rows).view(B, Q, -1)
        return center, pred_size
