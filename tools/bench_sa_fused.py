"""Forward + backward of the four set-abstraction MLPs at the headline sizes (B=8), through the
fused native calls.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import sa_ops, pointnet2_utils as PU  # noqa: E402

CFG = [(8, 50000, 2048, 64, 3, [64, 64, 128], 0.2), (8, 2048, 1024, 32, 128, [128, 128, 256], 0.4),
       (8, 1024, 512, 16, 256, [128, 128, 256], 0.8), (8, 512, 256, 16, 256, [128, 128, 256], 1.2)]
which = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3]
dev = "cuda"
for ci in which:
    B, N, m, ns, C, chans, radius = CFG[ci]
    rng = np.random.default_rng(ci)
    xyz = torch.from_numpy(rng.uniform(-3, 3, (B, N, 3)).astype(np.float32)).to(dev)
    new_xyz = xyz[:, :m].contiguous()
    idx = PU.ball_query(radius, ns, xyz, new_xyz)
    chans = [3 + C] + chans
    Ws = [torch.randn(chans[l + 1], chans[l], 1, 1, device=dev).mul_(0.1).requires_grad_(True) for l in range(3)]
    gs = [torch.ones(c, device=dev, requires_grad=True) for c in chans[1:]]
    bs = [torch.zeros(c, device=dev, requires_grad=True) for c in chans[1:]]
    running = [(torch.zeros(c, device=dev), torch.ones(c, device=dev)) for c in chans[1:]]
    feats = torch.randn(B, N, C, device=dev, requires_grad=ci > 0)
    cfg = dict(gather=True, radius=radius, normalize_xyz=True, pool=ns, training=True, eps=1e-5, momentum=0.1, running=running)
    params = []
    for W, g, b in zip(Ws, gs, bs):
        params += [W, g, b]

    def step():
        out = sa_ops.FusedMLP.apply(cfg, None, xyz, new_xyz, feats, idx, *params)
        out.backward(torch.ones_like(out))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        step()
    e.record(); torch.cuda.synchronize()
    print(f"SA{ci + 1}: {s.elapsed_time(e) / 10:.3f} ms fwd+bwd", flush=True)
