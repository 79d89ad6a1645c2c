"""Generate tests/golden/ops_*.npz -- golden vectors for the nine native ops.

The reference ships no CPU implementation and no vectors for these ops
(SURVEY.md §8c), so these come from the C oracle (oracle/eda_oracle.c), which
restates the CUDA kernels line by line and is cross-checked against an
independent numpy restatement (tests/ref_numpy.py) in tests/test_oracle.py.
They pin the oracle against regressions and travel to the GPU box, where the
HIP kernels are compared against them.

Run from the repo root:  python tools/gen_golden_ops.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_ext as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def cloud(rng, b, n, dup=0.0, origin=0.0, quant=None):
    p = rng.uniform(-1.5, 1.5, (b, n, 3)).astype(np.float32)
    p[..., 2] = np.abs(p[..., 2])
    if quant:
        p = (np.round(p * quant) / quant).astype(np.float32)
    for i in range(b):
        if dup > 0:
            k = max(1, int(n * dup))
            src = rng.integers(0, n, k); dst = rng.integers(0, n, k)
            p[i, dst] = p[i, src]                      # exact duplicates -> FPS ties
        if origin > 0:
            k = max(1, int(n * origin))
            p[i, rng.integers(0, n, k)] = rng.uniform(-0.015, 0.015, (k, 3)).astype(np.float32)
    return p


def make_case(name, seed, b, n, m, radius, nsample, c, **kw):
    rng = np.random.default_rng(seed)
    xyz = cloud(rng, b, n, **kw)
    t = torch.from_numpy
    out = {"xyz": xyz, "m": np.int32(m), "radius": np.float32(radius), "nsample": np.int32(nsample)}
    O.set_fma_mode(0)
    fps = O.furthest_point_sampling(t(xyz), m)
    out["fps_idx"] = fps.numpy()
    O.set_fma_mode(1)
    out["fps_idx_strict"] = O.furthest_point_sampling(t(xyz), m).numpy()
    O.set_fma_mode(0)
    xyz_t = t(xyz).transpose(1, 2).contiguous()                       # (b,3,n)
    new_xyz = O.gather_points(xyz_t, fps).transpose(1, 2).contiguous()  # (b,m,3)
    out["new_xyz"] = new_xyz.numpy()
    # also a few foreign centres far away -> empty balls (all-zero rows)
    centres = new_xyz.clone()
    centres[:, -1, :] = 50.0
    out["centres"] = centres.numpy()
    bq = O.ball_query(centres, t(xyz), radius, nsample)
    out["bq_idx"] = bq.numpy()
    feats = rng.standard_normal((b, c, n)).astype(np.float32)
    out["feats"] = feats
    grouped = O.group_points(t(feats), bq)
    out["grouped"] = grouped.numpy()
    gout = rng.standard_normal(grouped.shape).astype(np.float32)
    out["grouped_gout"] = gout
    out["group_grad"] = O.group_points_grad(t(gout), bq, n).numpy()
    gathered = O.gather_points(t(feats), fps)
    out["gathered"] = gathered.numpy()
    g2 = rng.standard_normal(gathered.shape).astype(np.float32)
    out["gathered_gout"] = g2
    out["gather_grad"] = O.gather_points_grad(t(g2), fps, n).numpy()
    # three_nn / interpolate: unknown = all points (first 256), known = sampled centres
    unk = t(xyz[:, :min(n, 256)].copy())
    d2, nn = O.three_nn(unk, new_xyz)
    out["nn_unknown"] = unk.numpy(); out["nn_dist2"] = d2.numpy(); out["nn_idx"] = nn.numpy()
    w = rng.uniform(0.05, 1, nn.shape).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    kf = rng.standard_normal((b, c, m)).astype(np.float32)
    out["interp_weight"] = w; out["interp_feats"] = kf
    interp = O.three_interpolate(t(kf), nn, t(w))
    out["interp"] = interp.numpy()
    g3 = rng.standard_normal(interp.shape).astype(np.float32)
    out["interp_gout"] = g3
    out["interp_grad"] = O.three_interpolate_grad(t(g3), nn, t(w), m).numpy()
    path = os.path.join(OUT, f"ops_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    O.build()
    os.makedirs(OUT, exist_ok=True)
    make_case("n9", 1, b=2, n=9, m=5, radius=0.9, nsample=4, c=3, origin=0.2)
    make_case("n512_ties", 2, b=2, n=512, m=128, radius=0.4, nsample=16, c=5, dup=0.3, origin=0.02)
    make_case("n600_quant", 3, b=2, n=600, m=200, radius=0.5, nsample=8, c=4, quant=4, origin=0.01)
    make_case("n4096", 4, b=2, n=4096, m=1024, radius=0.2, nsample=64, c=2, dup=0.05, origin=0.002)
