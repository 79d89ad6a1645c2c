"""Small heads around the encoder/decoder: seed objectness, query sampling,
box prediction.  Mirrors models/modules.py:19-178 (names = checkpoint contract).
"""
import numpy as np
import torch.nn as nn
import torch.nn.functional as F

from .nn_utils import Conv1dK1, bn_relu_rows, rows_ok
from .pointnet2_utils import gather_operation
from .encoder_decoder_layers import PositionEmbeddingLearned  # noqa: F401  (re-exported like the reference)


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

    def forward(self, xyz, features, sample_inds):
        new_xyz = gather_operation(xyz.transpose(1, 2).contiguous(), sample_inds).transpose(1, 2).contiguous()
        new_features = gather_operation(features, sample_inds).contiguous()
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

    def _forward_rows(self, features, base_xyz, end_points, prefix, features_rows):
        B, C, Q = features.shape
        rows = (features_rows if features_rows is not None else features.transpose(1, 2)).reshape(B * Q, C)
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
