"""PointNet++ backbone: 4 set-abstraction + 2 feature-propagation layers.

Mirrors models/backbone_module.py:23-144 (layer specs :44-81, outputs :113-143):
50 000 points -> 2048 -> 1024 -> 512 -> 256 centres, upsampled back to the 1024
seeds with 288 channels.
"""
from torch import nn

from .pointnet2_modules import PointnetFPModule, PointnetSAModuleVotes


class Pointnet2Backbone(nn.Module):
    def __init__(self, input_feature_dim=0, width=1, depth=2, output_dim=288):
        super().__init__()
        self.depth, self.width = depth, width
        w = width

        def sa(npoint, radius, nsample, cin, mid, cout):
            return PointnetSAModuleVotes(npoint=npoint, radius=radius, nsample=nsample,
                                         mlp=[cin] + [mid] * depth + [cout],
                                         use_xyz=True, normalize_xyz=True)

        self.sa1 = sa(2048, 0.2, 64, input_feature_dim, 64 * w, 128 * w)
        self.sa2 = sa(1024, 0.4, 32, 128 * w, 128 * w, 256 * w)
        self.sa3 = sa(512, 0.8, 16, 256 * w, 128 * w, 256 * w)
        self.sa4 = sa(256, 1.2, 16, 256 * w, 128 * w, 256 * w)
        self.fp1 = PointnetFPModule(mlp=[512 * w, 256 * w, 256 * w])
        self.fp2 = PointnetFPModule(mlp=[512 * w, 256 * w, output_dim])

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def geometry(self, xyz, sa1_inds=None):
        """Everything the backbone derives from the coordinates (B, N, 3) alone: per SA level (inds, new_xyz, ball-query
        idx), per FP level (3-NN idx, weights) -- four samplings, four ball queries, two 3-NN searches, none of which
        any gradient flows through.  Returns a flat dict of tensors; `forward(..., geometry=)` takes it back."""
        g, cur = {}, xyz
        for name, layer in (("sa1", self.sa1), ("sa2", self.sa2), ("sa3", self.sa3), ("sa4", self.sa4)):
            inds, new_xyz, idx = layer.geometry(cur, inds=sa1_inds if name == "sa1" else None,
                                                xyz_in_sampling_order=name != "sa1")
            g[f"{name}_inds"], g[f"{name}_xyz"], g[f"{name}_idx"] = inds, new_xyz, idx
            cur = new_xyz
        g["fp1_idx"], g["fp1_weight"] = self.fp1.geometry(g["sa3_xyz"], g["sa4_xyz"])
        g["fp2_idx"], g["fp2_weight"] = self.fp2.geometry(g["sa2_xyz"], g["sa3_xyz"])
        return g

    def forward(self, pointcloud, end_points=None, sa1_inds=None, geometry=None):
        """sa1_inds: optional (B, 2048) int32 furthest-point-sampling indices of `pointcloud` computed by the caller
        (the `inds` argument of the reference's PointnetSAModuleVotes.forward, pointnet2_modules.py:217-235): they
        depend on the input coordinates only, so an input pipeline can sample batch i+1 while step i trains."""
        end_points = end_points if end_points else {}
        xyz, features = self._break_up_pc(pointcloud)
        for name, layer in (("sa1", self.sa1), ("sa2", self.sa2), ("sa3", self.sa3), ("sa4", self.sa4)):
            # SA2..SA4 sample from the previous level's samples, which are in sampling order: their FPS is the
            # prefix 0..m-1 unless a tie intervenes (the reference notes it, backbone_module.py:122-131);
            # the library verifies that instead of running the dependent rounds
            geo = (geometry[f"{name}_inds"], geometry[f"{name}_xyz"], geometry[f"{name}_idx"]) if geometry else None
            xyz, features, inds = layer(xyz, features, inds=sa1_inds if name == "sa1" else None,
                                        xyz_in_sampling_order=name != "sa1", geometry=geo)
            if name in ("sa1", "sa2"):
                end_points[f"{name}_inds"] = inds
            end_points[f"{name}_xyz"] = xyz
            end_points[f"{name}_features"] = features
        f = self.fp1(end_points["sa3_xyz"], end_points["sa4_xyz"], end_points["sa3_features"],
                     end_points["sa4_features"],
                     geometry=(geometry["fp1_idx"], geometry["fp1_weight"]) if geometry else None)
        f = self.fp2(end_points["sa2_xyz"], end_points["sa3_xyz"], end_points["sa2_features"], f,
                     geometry=(geometry["fp2_idx"], geometry["fp2_weight"]) if geometry else None)
        end_points["fp2_features"] = f
        end_points["fp2_xyz"] = end_points["sa2_xyz"]
        num_seed = end_points["fp2_xyz"].shape[1]
        end_points["fp2_inds"] = end_points["sa1_inds"][:, 0:num_seed]
        return end_points
