"""For a maintainer with an NVIDIA box: settle which fp32 contraction nvcc applies to the reference's
squared-distance expression (DESIGN.md "Canonical arithmetic"; this repo's mode 0 vs mode 1).

Run INSIDE the reference checkout with its CUDA extension built (pointnet2/_ext):

    python dump_fma_probe.py out.npz

It samples FPS / ball-query / 3-NN indices on clouds built so that the two candidate roundings give
DIFFERENT indices (quantised coordinates: many distances differ by one ulp between
t = b*b; t = fma(a,a,t); t = fma(c,c,t)   [mode 0]   and   ((a*a + b*b) + c*c)   [mode 1]).
Back here:  python tools/check_fma_probe.py out.npz   reports which mode reproduces the CUDA output.
"""
import sys

import numpy as np
import torch
from pointnet2 import pointnet2_utils as PU


def main():
    rng = np.random.default_rng(2024)
    out = {}
    for i, (n, m, q) in enumerate([(4096, 512, 1 / 64), (20000, 1024, 1 / 256), (50000, 2048, 1 / 1024)]):
        xyz = (np.round(rng.uniform(-3, 3, (2, n, 3)) / q) * q + rng.uniform(-1e-4, 1e-4, (2, n, 3))).astype(np.float32)
        t = torch.from_numpy(xyz).cuda()
        idx = PU.furthest_point_sample(t, m)
        new = PU.gather_operation(t.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
        bq = PU.ball_query(0.2, 32, t, new)
        d, nn = PU.three_nn(t[:, :2048].contiguous(), new)
        out.update({f"xyz{i}": xyz, f"fps{i}": idx.cpu().numpy(), f"bq{i}": bq.cpu().numpy(),
                    f"nn{i}": nn.cpu().numpy(), f"nnd{i}": d.cpu().numpy()})
    np.savez_compressed(sys.argv[1], **out)


if __name__ == "__main__":
    main()
