"""Inference form of the backbone's first set-abstraction level in ONE launch (csrc/sa_eval.hip): ball-query rows gathered,
three conv1x1 + BatchNorm(running statistics) + ReLU layers and the max over the neighbourhood without any intermediate
tensor in HBM.  Against (a) an fp64 composition of the same chain (pointnet2/pointnet2_utils.py:317-376,
pytorch_utils.py:67-120 in eval mode, pointnet2_modules.py:251-257) within the north star's 1e-4, (b) the three-launch
eval path of eda_sa_fused_fwd_f32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(B, N, m, ns, seed):
    from eda_amd import pointnet2_utils as PU
    from eda_amd.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    xyz = torch.from_numpy(rng.uniform(-2, 2, (B, N, 3)).astype(np.float32)).cuda()
    feats = torch.from_numpy(rng.uniform(-1, 1, (B, 3, N)).astype(np.float32)).cuda()
    sa = PointnetSAModuleVotes(npoint=m, radius=0.3, nsample=ns, mlp=[3, 64, 64, 128], use_xyz=True, normalize_xyz=True).cuda()
    for mod in sa.modules():                         # non-trivial running statistics and affine parameters
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.3); mod.running_var.uniform_(0.5, 2.0)
            mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.2)
    sa.eval()
    return sa, xyz, feats


@pytest.mark.parametrize("B,N,m,ns", [(2, 4096, 256, 64), (1, 3000, 100, 64), (2, 2048, 130, 32), (1, 1000, 33, 48)])
def test_one_pass_eval_matches_fp64_and_the_three_launch_path(B, N, m, ns):
    from eda_amd import sa_ops
    sa, xyz, feats = _setup(B, N, m, ns, seed=N + m)
    calls = []
    orig = sa_ops._one_pass_eval

    def spy(*a, **k):
        out = orig(*a, **k)
        calls.append(out is not None)
        return out
    sa_ops._one_pass_eval = spy
    try:
        with torch.no_grad():
            new_xyz, out_one, inds = sa(xyz, feats)
        assert calls == [True]
        sa_ops._ONE_PASS_EVAL = False
        with torch.no_grad():
            _, out_three, inds3 = sa(xyz, feats)
    finally:
        sa_ops._one_pass_eval = orig
        sa_ops._ONE_PASS_EVAL = True
    assert torch.equal(inds, inds3)
    # fp64 composition on the same indices
    from eda_amd import pointnet2_utils as PU
    idx = PU.ball_query(0.3, ns, xyz, new_xyz).long()
    bidx = torch.arange(B, device="cuda").view(B, 1, 1)
    gx = (xyz[bidx, idx] - new_xyz[:, :, None, :]).double() / 0.3                 # (B, m, ns, 3): true division, as the reference
    gf = feats.transpose(1, 2)[bidx, idx].double()
    h = torch.cat([gx, gf], -1)
    for layer in sa.mlp_module.layers():
        w = layer.conv.weight.reshape(layer.conv.weight.shape[0], -1).double()
        bn = layer.bn.bn
        h = h @ w.t()
        h = (h - bn.running_mean.double()) / torch.sqrt(bn.running_var.double() + bn.eps) * bn.weight.double() + bn.bias.double()
        h = h.relu()
    ref = h.max(dim=2)[0].transpose(1, 2)                                          # (B, C, m)
    scale = ref.abs().max().item()
    err1 = (out_one.double() - ref).abs().max().item()
    err3 = (out_three.double() - ref).abs().max().item()
    assert err1 <= 1e-4 * scale, (err1, scale)
    assert err1 <= 4 * err3 + 1e-6 * scale, (err1, err3)          # the bf16 x 3 chain is as accurate as the fp32-MFMA one
    torch.testing.assert_close(out_one, out_three, rtol=1e-4, atol=1e-5 * scale)
