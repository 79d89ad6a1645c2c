"""CPU tests of the oracle (oracle/eda_oracle.c): against the independent numpy
restatement (tests/ref_numpy.py), against the committed golden vectors, and on
the edge cases the reference semantics define."""
import glob
import os

import numpy as np
import pytest
import torch

import ref_numpy as R

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ops_*.npz")))
t = torch.from_numpy


def _cloud(rng, n, dup=0.0, origin=0.0, quant=None):
    p = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    if quant:
        p = (np.round(p * quant) / quant).astype(np.float32)
    if dup > 0:
        k = int(n * dup)
        p[rng.integers(0, n, k)] = p[rng.integers(0, n, k)]
    if origin > 0:
        k = max(1, int(n * origin))
        p[rng.integers(0, n, k)] = rng.uniform(-0.02, 0.02, (k, 3)).astype(np.float32)
    return p


def test_opt_n_threads(oracle):
    # cuda_utils.h:20-24
    for w, exp in [(1, 1), (2, 2), (3, 2), (9, 8), (511, 256), (512, 512), (513, 512),
                   (1024, 512), (2048, 512), (4096, 512), (50000, 512)]:
        assert oracle.opt_n_threads(w) == exp
        assert R.opt_n_threads(w) == exp


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n,m,kw", [
    (9, 5, {}), (64, 16, {}), (512, 128, dict(dup=0.3)), (600, 100, dict(dup=0.2, origin=0.05)),
    (1000, 64, dict(quant=4)), (37, 37, dict(quant=2)), (3, 3, dict(origin=1.0)), (1, 1, {}),
    (5, 9, {}),
])
def test_oracle_vs_numpy(oracle, mode, n, m, kw):
    rng = np.random.default_rng(n * 131 + m)
    oracle.set_fma_mode(mode)
    try:
        p = _cloud(rng, n, **kw)
        got = oracle.furthest_point_sampling(t(p)[None], m)[0].numpy()
        exp = R.fps(p, m, mode)
        assert (got == exp).all()
        ctr = np.ascontiguousarray(p[exp[:max(1, m // 2)]])
        for r, ns in [(0.5, 8), (1.5, 16), (0.05, 4)]:
            g = oracle.ball_query(t(ctr)[None], t(p)[None], r, ns)[0].numpy()
            assert (g == R.ball_query(ctr, p, r, ns, mode)).all()
        unk = _cloud(rng, 50, quant=kw.get("quant"))
        d2, ix = oracle.three_nn(t(unk)[None], t(p)[None])
        e2, ei = R.three_nn(unk, p, mode)
        assert (ix[0].numpy() == ei).all()
        assert (d2[0].numpy() == e2).all()
    finally:
        oracle.set_fma_mode(0)


def test_fps_tie_rule_bitreversal(oracle):
    """Ties at reference threads {1, 256}: the tree keeps slot idx1, so 256 wins
    (winner = min over (bitreverse9(k mod 512), k))."""
    n = 1024
    p = np.zeros((n, 3), np.float32)
    p[:, 0] = 1.0                       # everything at the same place, outside the origin ball
    p[1] = [5, 0, 0]
    p[256] = [5, 0, 0]                  # exact tie for the farthest point from point 0
    p[513] = [5, 0, 0]                  # same thread as k=1, higher k -> loses to k=1
    got = oracle.furthest_point_sampling(t(p)[None], 2)[0].numpy()
    assert got.tolist() == [0, 256]
    assert R.fps(p, 2).tolist() == [0, 256]


def test_fps_origin_ball_and_index0(oracle):
    # points within sqrt(1e-3) of the origin are never candidates, but index 0 is always emitted
    p = np.array([[0.01, 0.0, 0.0], [0.02, 0.01, 0.0], [1, 1, 1], [2, 0, 0], [0.03, 0, 0]], np.float32)
    got = oracle.furthest_point_sampling(t(p)[None], 4)[0].numpy()
    assert got[0] == 0 and set(got[1:3].tolist()) == {2, 3}
    # all invalid -> every round emits index 0
    q = np.full((7, 3), 0.01, np.float32)
    assert oracle.furthest_point_sampling(t(q)[None], 5)[0].tolist() == [0] * 5


def test_ball_query_padding_and_empty(oracle):
    xyz = np.array([[0, 0, 0], [0.1, 0, 0], [5, 5, 5], [0.05, 0, 0]], np.float32)
    ctr = np.array([[0, 0, 0], [9, 9, 9], [5, 5, 5]], np.float32)
    idx = oracle.ball_query(t(ctr)[None], t(xyz)[None], 0.2, 4)[0].numpy()
    assert idx[0].tolist() == [0, 1, 3, 0]      # padded with the first hit
    assert idx[1].tolist() == [0, 0, 0, 0]      # empty ball: zero row
    assert idx[2].tolist() == [2, 2, 2, 2]
    # strict '<': a point exactly at the radius is excluded (0.25 is exact in fp32)
    xyz2 = np.array([[0.5, 0, 0], [0.25, 0, 0]], np.float32)
    idx2 = oracle.ball_query(t(np.zeros((1, 3), np.float32))[None], t(xyz2)[None], 0.5, 2)[0].numpy()
    assert idx2[0].tolist() == [1, 1]


def test_three_nn_fewer_than_three(oracle):
    unk = np.zeros((2, 3), np.float32)
    kn = np.array([[1, 0, 0], [0, 2, 0]], np.float32)
    d2, ix = oracle.three_nn(t(unk)[None], t(kn)[None])
    assert ix[0, 0].tolist() == [0, 1, 0]
    assert d2[0, 0, 0] == 1 and d2[0, 0, 1] == 4 and np.isinf(d2[0, 0, 2].item())


def test_argument_checks(oracle):
    x = torch.zeros(1, 4, 3)
    with pytest.raises(RuntimeError, match="must be a contiguous tensor"):
        oracle.furthest_point_sampling(x.transpose(1, 2), 2)
    with pytest.raises(RuntimeError, match="must be a float tensor"):
        oracle.furthest_point_sampling(x.double(), 2)
    with pytest.raises(RuntimeError, match="must be an int tensor"):
        oracle.gather_points(torch.zeros(1, 3, 4), torch.zeros(1, 2, dtype=torch.int64))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_golden(oracle, path):
    g = np.load(path)
    xyz = t(g["xyz"]); m = int(g["m"]); r = float(g["radius"]); ns = int(g["nsample"])
    n = xyz.shape[1]
    fps = oracle.furthest_point_sampling(xyz, m)
    assert (fps.numpy() == g["fps_idx"]).all()
    oracle.set_fma_mode(1)
    try:
        assert (oracle.furthest_point_sampling(xyz, m).numpy() == g["fps_idx_strict"]).all()
    finally:
        oracle.set_fma_mode(0)
    bq = oracle.ball_query(t(g["centres"]), xyz, r, ns)
    assert (bq.numpy() == g["bq_idx"]).all()
    grouped = oracle.group_points(t(g["feats"]), bq)
    assert (grouped.numpy() == g["grouped"]).all()
    # group is a pure gather: independent numpy check
    feats = g["feats"]; b, c, _ = feats.shape
    exp = np.take_along_axis(feats, g["bq_idx"].reshape(b, 1, -1).astype(np.int64).repeat(c, 1), 2)
    assert (grouped.numpy().reshape(b, c, -1) == exp).all()
    assert np.allclose(oracle.group_points_grad(t(g["grouped_gout"]), bq, n).numpy(), g["group_grad"])
    assert (oracle.gather_points(t(g["feats"]), fps).numpy() == g["gathered"]).all()
    assert np.allclose(oracle.gather_points_grad(t(g["gathered_gout"]), fps, n).numpy(), g["gather_grad"])
    d2, nn = oracle.three_nn(t(g["nn_unknown"]), t(g["new_xyz"]))
    assert (nn.numpy() == g["nn_idx"]).all() and (d2.numpy() == g["nn_dist2"]).all()
    it = oracle.three_interpolate(t(g["interp_feats"]), nn, t(g["interp_weight"]))
    assert (it.numpy() == g["interp"]).all()
    ig = oracle.three_interpolate_grad(t(g["interp_gout"]), nn, t(g["interp_weight"]), m)
    assert np.allclose(ig.numpy(), g["interp_grad"])


def test_fps_prefix_property(oracle):
    """FPS of an FPS-ordered set returns 0..m-1 (reference comment
    models/backbone_module.py:122; holds when there are no exact ties)."""
    rng = np.random.default_rng(5)
    p = rng.uniform(0.5, 3, (1, 2000, 3)).astype(np.float32)
    i1 = oracle.furthest_point_sampling(t(p), 512)
    q = t(p)[0][i1[0].long()][None].contiguous()
    i2 = oracle.furthest_point_sampling(q, 256)
    assert i2[0].tolist() == list(range(256))


def test_mt_variants_identical(oracle):
    rng = np.random.default_rng(6)
    p = t(rng.uniform(-2, 2, (3, 3000, 3)).astype(np.float32))
    oracle.set_threads(4)
    a = oracle.furthest_point_sampling(p, 200)
    b = oracle.furthest_point_sampling(p, 200, mt=True)
    assert (a == b).all()
    ctr = torch.stack([p[i][a[i].long()] for i in range(3)]).contiguous()
    assert (oracle.ball_query(ctr, p, 0.4, 16) == oracle.ball_query(ctr, p, 0.4, 16, mt=True)).all()
    bq = oracle.ball_query(ctr, p, 0.4, 16)
    f = t(rng.standard_normal((3, 7, 3000)).astype(np.float32))
    assert (oracle.group_points(f, bq) == oracle.group_points(f, bq, mt=True)).all()
    oracle.set_threads(1)
