"""Micro-timings of the native ops at the headline shapes (B=8, N=50 000)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eda_amd import ext, synthetic  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    B = int(os.environ.get("B", 8))
    pc = torch.from_numpy(synthetic.batch(range(B), 50000)[:, :, :3].copy()).cuda()
    xyz = pc
    levels = [(2048, 0.2, 64, 3), (1024, 0.4, 32, 128), (512, 0.8, 16, 256), (256, 1.2, 16, 256)]
    for m, r, ns, c in levels:
        n = xyz.shape[1]
        t_fps = timeit(lambda: ext.furthest_point_sampling(xyz, m), iters=5)
        idx = ext.furthest_point_sampling(xyz, m)
        xyz_t = xyz.transpose(1, 2).contiguous()
        new = ext.gather_points(xyz_t, idx).transpose(1, 2).contiguous()
        t_bq = timeit(lambda: ext.ball_query(new, xyz, r, ns))
        bq = ext.ball_query(new, xyz, r, ns)
        feats = torch.randn(B, c, n, device="cuda")
        t_gx = timeit(lambda: ext.group_points(xyz_t, bq))
        t_gf = timeit(lambda: ext.group_points(feats, bq))
        go = torch.randn(B, c, m, ns, device="cuda")
        t_gg = timeit(lambda: ext.group_points_grad(go, bq, n))
        bq_bytes = B * (12 * n + 12 * m + 4 * m * ns)
        grp_bytes = B * (4 * c * n + 4 * m * ns + 4 * c * m * ns)
        print(f"N={n:6d} m={m:5d} ns={ns:3d} C={c:4d} | fps {t_fps*1e3:9.1f} us ({t_fps*1e3/(m-1):.2f} us/round)"
              f" | bq {t_bq*1e3:8.1f} us | group xyz {t_gx*1e3:7.1f} us feat {t_gf*1e3:7.1f} us"
              f" ({grp_bytes/t_gf/1e6:.0f} GB/s) | group_grad {t_gg*1e3:7.1f} us", flush=True)
        xyz = new


if __name__ == "__main__":
    main()
