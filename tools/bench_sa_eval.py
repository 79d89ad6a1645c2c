"""SA1 of the backbone in inference mode at the headline size (8 x 50 000 points, 2048 centres x 64 neighbours): the one-launch
kernel (csrc/sa_eval.hip) against the three-launch eval path; HIP-event time of the level's MLP part only (sampling /
ball query are done once outside).  Run under rocprofv3 for the kernel durations."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import sa_ops, synthetic, pointnet2_utils as PU  # noqa: E402
from eda_amd.pointnet2_modules import PointnetSAModuleVotes  # noqa: E402


def main():
    B, N, m, ns = 8, 50000, 2048, 64
    pc = torch.from_numpy(synthetic.batch(list(range(B)), N)).cuda()
    xyz = pc[:, :, :3].contiguous()
    feats_cl = pc[:, :, 3:6].contiguous()
    torch.manual_seed(0)
    sa = PointnetSAModuleVotes(npoint=m, radius=0.2, nsample=ns, mlp=[3, 64, 64, 128], use_xyz=True, normalize_xyz=True).cuda().eval()
    inds = PU.furthest_point_sample(xyz, m)
    new_xyz = PU.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    idx = PU.ball_query(0.2, ns, xyz, new_xyz)
    layers = sa.mlp_module.layers()
    bns = [l.bn.bn for l in layers]
    cfg = dict(gather=True, radius=0.2, normalize_xyz=True, pool=ns, training=False, eps=bns[0].eps, momentum=bns[0].momentum,
               running=[(bn.running_mean, bn.running_var) for bn in bns])
    params = []
    for layer, bn in zip(layers, bns):
        params += [layer.conv.weight, bn.weight, bn.bias]
    iters = int(os.environ.get("ITERS", 10))
    with torch.no_grad():
        for name, fn in (("one launch", lambda: sa_ops._one_pass_eval(cfg, xyz, new_xyz, feats_cl, idx, layers, bns, ns)),
                         ("three launches + pool", lambda: sa_ops.FusedMLP.apply(cfg, None, xyz, new_xyz, feats_cl, idx, *params))):
            out = fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                out = fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            alg = B * 2848128                                  # fused SA1 forward bytes per scene
            print(f"{name:24s} {ms*1e3:8.1f} us   {alg/ms/1e6:8.1f} GB/s on the {alg/1e6:.1f} MB of SURVEY 8d = {alg/ms/1e6/8000:.4f} of 8 TB/s;"
                  f" 26.6 GFLOP -> {26.6/ms:.1f} TFLOP/s", flush=True)
    print("checksum", float(out.sum()))


if __name__ == "__main__":
    main()
