"""GPU parity: every native op through the C ABI (eda_amd.ext -> libeda_hip.so)
against the CPU oracle on the same seeded inputs, against the committed golden
vectors, and -- at BASELINE.json's full size (B=8, N=50 000) -- through
size-independent properties.  Bit-exact for indices and copies; <= 1e-4 relative
for the atomically accumulated gradients (the reference itself is run-to-run
non-deterministic there)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ops_*.npz")))
t = torch.from_numpy


@pytest.fixture(scope="module")
def ext():
    from eda_amd import ext as e
    return e


def dev(a):
    return (t(a) if isinstance(a, np.ndarray) else a).cuda()


def _cloud(rng, b, n, dup=0.0, origin=0.0, quant=None):
    p = rng.uniform(-2, 2, (b, n, 3)).astype(np.float32)
    if quant:
        p = (np.round(p * quant) / quant).astype(np.float32)
    for i in range(b):
        if dup > 0:
            k = int(n * dup)
            p[i, rng.integers(0, n, k)] = p[i, rng.integers(0, n, k)]
        if origin > 0:
            k = max(1, int(n * origin))
            p[i, rng.integers(0, n, k)] = rng.uniform(-0.02, 0.02, (k, 3)).astype(np.float32)
    return p


def _check_fps_status(ext):
    torch.cuda.synchronize()


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_vectors(ext, path):
    g = np.load(path)
    xyz = dev(g["xyz"]); m = int(g["m"]); r = float(g["radius"]); ns = int(g["nsample"])
    n = xyz.shape[1]
    fps = ext.furthest_point_sampling(xyz, m)
    assert (fps.cpu().numpy() == g["fps_idx"]).all()
    ext.set_fma_mode(1)
    try:
        assert (ext.furthest_point_sampling(xyz, m).cpu().numpy() == g["fps_idx_strict"]).all()
    finally:
        ext.set_fma_mode(0)
    bq = ext.ball_query(dev(g["centres"]), xyz, r, ns)
    assert (bq.cpu().numpy() == g["bq_idx"]).all()
    assert (ext.group_points(dev(g["feats"]), bq).cpu().numpy() == g["grouped"]).all()
    gg = ext.group_points_grad(dev(g["grouped_gout"]), bq, n).cpu().numpy()
    np.testing.assert_allclose(gg, g["group_grad"], rtol=1e-4, atol=1e-5)
    assert (ext.gather_points(dev(g["feats"]), fps).cpu().numpy() == g["gathered"]).all()
    np.testing.assert_allclose(ext.gather_points_grad(dev(g["gathered_gout"]), fps, n).cpu().numpy(),
                               g["gather_grad"], rtol=1e-4, atol=1e-5)
    d2, nn = ext.three_nn(dev(g["nn_unknown"]), dev(g["new_xyz"]))
    assert (nn.cpu().numpy() == g["nn_idx"]).all()
    assert (d2.cpu().numpy() == g["nn_dist2"]).all()
    it = ext.three_interpolate(dev(g["interp_feats"]), nn, dev(g["interp_weight"]))
    assert (it.cpu().numpy() == g["interp"]).all()
    ig = ext.three_interpolate_grad(dev(g["interp_gout"]), nn, dev(g["interp_weight"]), m)
    np.testing.assert_allclose(ig.cpu().numpy(), g["interp_grad"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("b,n,m,kw", [
    (1, 1, 1, {}), (2, 9, 5, {}), (3, 64, 64, {}), (2, 512, 128, dict(dup=0.3)),
    (2, 600, 100, dict(dup=0.2, origin=0.05)), (2, 1000, 64, dict(quant=4)),
    (1, 37, 50, dict(quant=2)), (1, 3, 3, dict(origin=1.0)),
    (2, 2048, 1024, {}), (2, 2049, 300, dict(dup=0.1)), (1, 8192, 512, dict(quant=8)),
    (2, 8193, 200, dict(dup=0.05)), (1, 20000, 700, dict(quant=16, origin=0.001)),
    (3, 50000, 300, dict(dup=0.02, origin=0.0005)),
    (8, 50000, 64, {}),                       # bench geometry: 8 clusters of 7 workgroups, one per XCD
    (38, 50000, 12, dict(dup=0.01)),          # more scenes than one launch holds (36): two launches
])
def test_fps_vs_oracle(ext, oracle, mode, b, n, m, kw):
    rng = np.random.default_rng(n * 7 + m + mode)
    p = _cloud(rng, b, n, **kw)
    oracle.set_fma_mode(mode); ext.set_fma_mode(mode)
    try:
        exp = oracle.furthest_point_sampling(t(p), m).numpy()
        got = ext.furthest_point_sampling(dev(p), m).cpu().numpy()
    finally:
        oracle.set_fma_mode(0); ext.set_fma_mode(0)
    assert (got == exp).all(), (np.argwhere(got != exp)[:5], got[got != exp][:5], exp[got != exp][:5])


@pytest.mark.parametrize("b,n,m,r,ns", [
    (2, 9, 5, 0.9, 4), (2, 700, 100, 0.4, 16), (1, 5000, 777, 0.3, 32), (2, 4096, 1024, 0.2, 64),
    (1, 1030, 17, 5.0, 128), (1, 100, 3, 0.01, 1), (2, 2048, 1024, 0.4, 32),
])
def test_ball_query_vs_oracle(ext, oracle, b, n, m, r, ns):
    rng = np.random.default_rng(n + m)
    p = _cloud(rng, b, n, dup=0.05)
    ctr = np.ascontiguousarray(p[:, rng.permutation(n)[:m]]) if m <= n else _cloud(rng, b, m)
    ctr[:, -1] = 77.0                                     # empty ball
    exp = oracle.ball_query(t(ctr), t(p), r, ns).numpy()
    got = ext.ball_query(dev(ctr), dev(p), r, ns).cpu().numpy()
    assert (got == exp).all()


@pytest.mark.parametrize("b,c,n,m,ns", [(2, 3, 100, 10, 4), (2, 131, 2048, 64, 32), (1, 7, 999, 33, 5),
                                        (2, 128, 2048, 1024, 32)])
def test_group_and_grad_vs_oracle(ext, oracle, b, c, n, m, ns):
    rng = np.random.default_rng(c + n)
    f = rng.standard_normal((b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    idx[:, :, ns // 2:] = idx[:, :, :1]                   # padded duplicates, like ball query rows
    exp = oracle.group_points(t(f), t(idx)).numpy()
    got = ext.group_points(dev(f), dev(idx))
    assert (got.cpu().numpy() == exp).all()
    assert got.data_ptr() != dev(f).data_ptr() and got.is_contiguous()   # fresh tensor, callers mutate it
    go = rng.standard_normal(exp.shape).astype(np.float32)
    eg = oracle.group_points_grad(t(go), t(idx), n).numpy()
    gg = ext.group_points_grad(dev(go), dev(idx), n).cpu().numpy()
    np.testing.assert_allclose(gg, eg, rtol=1e-4, atol=1e-4)
    fi = rng.integers(0, n, (b, m)).astype(np.int32)
    assert (ext.gather_points(dev(f), dev(fi)).cpu().numpy() == oracle.gather_points(t(f), t(fi)).numpy()).all()
    g2 = rng.standard_normal((b, c, m)).astype(np.float32)
    np.testing.assert_allclose(ext.gather_points_grad(dev(g2), dev(fi), n).cpu().numpy(),
                               oracle.gather_points_grad(t(g2), t(fi), n).numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("b,n,m,c", [(2, 512, 256, 256), (1, 1024, 512, 17), (2, 50, 2, 3), (1, 300, 1500, 4)])
def test_three_nn_interpolate_vs_oracle(ext, oracle, b, n, m, c):
    rng = np.random.default_rng(n + m)
    unk = _cloud(rng, b, n, quant=8)
    kn = _cloud(rng, b, m, quant=8)
    ed, ei = oracle.three_nn(t(unk), t(kn))
    gd, gi = ext.three_nn(dev(unk), dev(kn))
    assert (gi.cpu() == ei).all() and (gd.cpu() == ed).all()
    w = rng.uniform(0.05, 1, (b, n, 3)).astype(np.float32)
    f = rng.standard_normal((b, c, m)).astype(np.float32)
    assert (ext.three_interpolate(dev(f), gi, dev(w)).cpu() == oracle.three_interpolate(t(f), ei, t(w))).all()
    go = rng.standard_normal((b, c, n)).astype(np.float32)
    np.testing.assert_allclose(ext.three_interpolate_grad(dev(go), gi, dev(w), m).cpu().numpy(),
                               oracle.three_interpolate_grad(t(go), ei, t(w), m).numpy(), rtol=1e-4, atol=1e-4)


def test_full_size_sa_stack_bit_exact(ext, oracle):
    """BASELINE.json configs[1]: B=8 synthetic 50k-point scenes through the four
    SA levels' index ops; FPS and ball-query indices bit-identical to the oracle."""
    from eda_amd import synthetic
    pc = synthetic.batch(range(8), 50000)[:, :, :3].copy()
    oracle.set_threads(os.cpu_count() or 1)
    xyz_c = t(pc); xyz_g = dev(pc)
    for m, r, ns in [(2048, 0.2, 64), (1024, 0.4, 32), (512, 0.8, 16), (256, 1.2, 16)]:
        ei = oracle.furthest_point_sampling(xyz_c, m, mt=True)
        gi = ext.furthest_point_sampling(xyz_g, m)
        assert (gi.cpu() == ei).all()
        new_c = torch.gather(xyz_c, 1, ei.long()[..., None].expand(-1, -1, 3)).contiguous()
        new_g = ext.gather_points(xyz_g.transpose(1, 2).contiguous(), gi).transpose(1, 2).contiguous()
        assert (new_g.cpu() == new_c).all()
        eb = oracle.ball_query(new_c, xyz_c, r, ns, mt=True)
        gb = ext.ball_query(new_g, xyz_g, r, ns)
        assert (gb.cpu() == eb).all()
        xyz_c, xyz_g = new_c, new_g
    oracle.set_threads(1)


def test_full_size_properties(ext):
    """Size-independent properties at N=50 000 (no oracle): FPS indices are
    distinct and start at 0; FPS of the FPS-ordered prefix is the identity; every
    ball-query row is ascending up to its padding, lies inside the radius and
    that is not full contains its own centre."""
    from eda_amd import synthetic
    pc = dev(synthetic.batch([11, 12], 50000)[:, :, :3].copy())
    idx = ext.furthest_point_sampling(pc, 2048)
    assert (idx[:, 0] == 0).all()
    for b in range(2):
        assert idx[b].unique().numel() == 2048
    new = torch.gather(pc, 1, idx.long()[..., None].expand(-1, -1, 3)).contiguous()
    again = ext.furthest_point_sampling(new, 1024)
    assert (again.cpu() == torch.arange(1024, dtype=torch.int32)).all()
    bq = ext.ball_query(new, pc, 0.2, 64).long()
    nb = torch.gather(pc[:, None].expand(-1, 2048, -1, -1), 2, bq[..., None].expand(-1, -1, -1, 3))
    d2 = ((nb - new[:, :, None]) ** 2).sum(-1)
    assert (d2 < 0.2 * 0.2 + 1e-6).all()
    first = bq[..., :1]
    diffs = bq[..., 1:] - bq[..., :-1]
    # ascending until the padding starts (padding repeats the first hit)
    is_pad = bq[..., 1:] == first
    assert ((diffs > 0) | is_pad).all()
    # a centre is one of the points (d2 = 0), so a row that did not fill up contains it
    not_full = bq[..., -1] == bq[..., 0]
    has_self = (bq == idx.long()[..., None]).any(-1)
    assert (has_self | ~not_full).all()


@pytest.mark.parametrize("b,n,m,r,ns,kind", [
    (2, 4096, 512, 0.2, 64, "room"), (2, 50000, 2048, 0.2, 64, "room"), (1, 50000, 300, 0.05, 8, "room"),
    (2, 20000, 1000, 0.7, 32, "uniform"), (1, 8000, 256, 3.0, 16, "uniform"), (1, 5000, 64, 50.0, 128, "uniform"),
    (1, 6000, 100, 0.3, 32, "flat"), (1, 7000, 50, 0.4, 2000, "blob"),
])
def test_grid_ball_query_vs_oracle(ext, oracle, b, n, m, r, ns, kind):
    """The uniform-grid path (n >= 4096) must equal the index-ordered scan bit for bit: dense and
    sparse balls, radius larger than the scene, degenerate (planar / single-cell) extents, foreign
    centres outside the bounding box, and a > 1024-hit ball (in-kernel overflow fallback)."""
    from eda_amd import synthetic
    rng = np.random.default_rng(n + m)
    if kind == "room":
        p = synthetic.batch(range(b), n)[:, :, :3].copy()
    elif kind == "uniform":
        p = rng.uniform(-3, 3, (b, n, 3)).astype(np.float32)
    elif kind == "flat":
        p = rng.uniform(-3, 3, (b, n, 3)).astype(np.float32); p[..., 2] = 1.25
    else:  # blob: thousands of points inside one ball
        p = (rng.normal(0, 0.05, (b, n, 3)) + 1.0).astype(np.float32)
    ctr = np.ascontiguousarray(p[:, rng.permutation(n)[:m]])
    ctr[:, -1] = 77.0                                     # far outside the bounding box: empty ball
    ctr[:, -2] = p.min(axis=1) - 0.01                     # just outside a corner
    oracle.set_threads(os.cpu_count() or 1)
    exp = oracle.ball_query(t(ctr), t(p), r, ns, mt=True).numpy()
    oracle.set_threads(1)
    got = ext.ball_query(dev(ctr), dev(p), r, ns).cpu().numpy()
    assert (got == exp).all(), np.argwhere(got != exp)[:5]


@pytest.mark.parametrize("fma_mode", [0, 1])
def test_fps_prefix_verification_equals_oracle_on_every_kind_of_input(oracle, fma_mode):
    """eda_furthest_point_sampling_prefix_f32 (used for SA2..SA4) must return exactly the oracle's indices
    whether its shortcut applies (input in sampling order, no tie: 0..m-1 verified) or not (unordered
    input; duplicated points = exact ties in the reference's arg-max order; points inside the origin ball;
    a first point inside the ball; fewer valid points than samples)."""
    import numpy as np
    from eda_amd import ext, _lib, synthetic
    L = _lib.lib()
    L.eda_set_fma_mode(fma_mode); oracle.set_fma_mode(fma_mode)
    try:
        rng = np.random.default_rng(7)
        base = torch.from_numpy(synthetic.batch([21, 22, 23], 50000)[:, :, :3].copy())
        sa1 = oracle.furthest_point_sampling(base, 2048)
        ordered = torch.gather(base, 1, sa1.long()[..., None].expand(-1, -1, 3)).contiguous()       # SA1 samples in sampling order
        cases = {"ordered": (ordered, 1024), "ordered_short": (ordered[:, :1024].contiguous(), 512),
                 "unordered": (base[:, :2048].contiguous(), 1024)}
        dup = ordered.clone()
        dup[0, 700] = dup[0, 3]; dup[1, 5] = dup[1, 4]; dup[2, 1500:1510] = dup[2, 100:110]       # exact duplicates
        cases["duplicates"] = (dup, 1024)
        ball = ordered.clone()
        ball[0, 0] = torch.tensor([0.01, 0.0, 0.01]); ball[1, 7] = torch.tensor([0.0, 0.02, 0.0]); ball[2, 600:650] *= 1e-3
        cases["origin_ball"] = (ball, 1024)
        few = torch.zeros(2, 600, 3); few[:, :40] = torch.from_numpy(rng.uniform(1, 2, (2, 40, 3)).astype(np.float32))
        cases["few_valid"] = (few, 256)
        quant = (torch.round(ordered * 16) / 16).contiguous()                                      # many equal distances
        cases["quantised"] = (quant, 512)
        for name, (xyz, m) in cases.items():
            want = oracle.furthest_point_sampling(xyz, m)
            got = ext.furthest_point_sampling_prefix(xyz.cuda(), m).cpu()
            assert torch.equal(got, want), (name, fma_mode, int((got != want).sum()))
            if name in ("ordered", "ordered_short"):
                assert torch.equal(want, torch.arange(m, dtype=torch.int32)[None].expand_as(want)), name
        assert ext.fps_status() == 0
    finally:
        L.eda_set_fma_mode(0); oracle.set_fma_mode(0)


# ---- the single-workgroup bucket sampler (csrc/fps_bucket.hip: scenes of 8193..65536 points) -----------------------
def _bucket_cases():
    rng = np.random.default_rng(2024)
    c = {}
    c["uniform_65536"] = (_cloud(rng, 1, 65536), 1500)
    c["min_size_8193"] = (_cloud(rng, 2, 8193), 700)
    p = _cloud(rng, 1, 30000); p[..., 2] = 0.25                          # planar: a degenerate scene box
    c["planar"] = (p, 900)
    p = _cloud(rng, 1, 20000); p[0, :, 1:] = 0.5                          # a line
    c["line"] = (p, 400)
    p = _cloud(rng, 1, 12000, dup=0.9)                                    # almost every point duplicated: exact ties
    c["mostly_duplicates"] = (p, 3000)
    p = np.repeat(_cloud(rng, 1, 40), 250, axis=1)                        # 40 distinct points x 250: m > distinct points
    c["forty_distinct_points"] = (np.ascontiguousarray(p[:, rng.permutation(10000)]), 120)
    p = _cloud(rng, 1, 9000); p[:] = p[:, :1]                             # one point, 9000 times
    c["one_point"] = (p, 50)
    c["quantised_coarse"] = (_cloud(rng, 1, 25000, quant=2), 600)         # 9^3 lattice: massive distance ties
    c["quantised_fine"] = (_cloud(rng, 2, 50000, quant=64, origin=0.002), 1200)
    p = _cloud(rng, 1, 10000, origin=0.3); p[0, 0] = (0.01, 0.0, 0.01)    # first point and 30 % inside the origin ball
    c["origin_ball"] = (p, 800)
    p = (_cloud(rng, 1, 9000) * 0.005).astype(np.float32)                 # EVERY point inside the ball: all indices 0
    c["all_in_origin_ball"] = (p, 30)
    p = _cloud(rng, 1, 16000); p[0, 8000:] += 50.0                        # two far-apart clusters
    c["two_clusters"] = (p, 1000)
    p = _cloud(rng, 1, 9000)                                              # sample EVERY point (m = n)
    c["all_points"] = (p, 9000)
    return c


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("name", sorted(_bucket_cases()))
def test_fps_bucket_sampler_vs_oracle(ext, oracle, mode, name, monkeypatch):
    monkeypatch.setenv("EDA_FPS_BUCKET", "1")        # (default policy: the cluster kernels, repaired by this sampler)
    p, m = _bucket_cases()[name]
    oracle.set_fma_mode(mode); ext.set_fma_mode(mode)
    oracle.set_threads(os.cpu_count() or 1)
    try:
        exp = oracle.furthest_point_sampling(t(p), m, mt=True).numpy()
        got = ext.furthest_point_sampling(dev(p), m).cpu().numpy()
    finally:
        oracle.set_fma_mode(0); ext.set_fma_mode(0); oracle.set_threads(1)
    bad = np.argwhere(got != exp)
    assert bad.size == 0, (name, mode, bad[:5], got[got != exp][:5], exp[got != exp][:5])
    assert ext.fps_status() == 0


def test_fps_bucket_sampler_equals_the_cluster_kernels_and_needs_no_co_residency(ext, oracle, monkeypatch):
    """Bench geometry (8 x 50 000 -> 2048, synthetic rooms): the bucket sampler, the round-3 cluster kernels
    (EDA_FPS_BUCKET=0) and the oracle agree; the bucket sampler also while another stream keeps every CU busy (the
    cluster kernels need ~100 co-resident workgroups and give up without them)."""
    from eda_amd import synthetic
    xyz = torch.from_numpy(synthetic.batch(range(40, 48), 50000)[:, :, :3].copy()).cuda()
    monkeypatch.setenv("EDA_FPS_BUCKET", "1")
    got = ext.furthest_point_sampling(xyz, 2048)
    monkeypatch.setenv("EDA_FPS_BUCKET", "0")
    old = ext.furthest_point_sampling(xyz, 2048)
    monkeypatch.setenv("EDA_FPS_BUCKET", "1")
    assert torch.equal(got, old)
    oracle.set_threads(os.cpu_count() or 1)
    try:
        exp = oracle.furthest_point_sampling(xyz[:2].cpu(), 2048, mt=True)
    finally:
        oracle.set_threads(1)
    assert torch.equal(got[:2].cpu(), exp)
    busy = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device="cuda")
    with torch.cuda.stream(busy):
        for _ in range(6):
            a = (a @ a).clamp_(-1, 1)
    again = ext.furthest_point_sampling(xyz, 2048)
    torch.cuda.synchronize()
    assert torch.equal(again, got) and ext.fps_status() == 0


def test_fps_background_polling_gives_the_same_indices(ext, monkeypatch):
    """eda_fps_set_background / EDA_FPS_BACKGROUND=1 (the sampler as a prefetch underneath other work: one polled granule per
    hand-off record instead of five) changes the polling traffic, not the result: bench geometry, cluster kernels, bit-equal
    indices, no give-up -- also while another stream keeps the memory system busy."""
    from eda_amd import _lib, synthetic
    xyz = torch.from_numpy(synthetic.batch(range(60, 68), 50000)[:, :, :3].copy()).cuda()
    monkeypatch.setenv("EDA_FPS_BUCKET", "0")
    wide = ext.furthest_point_sampling(xyz, 2048)
    L = _lib.lib()
    try:
        assert L.eda_fps_set_background(1) == 0
        lean = ext.furthest_point_sampling(xyz, 2048)
        busy = torch.cuda.Stream()
        big = torch.empty(1 << 27, device="cuda"); big2 = torch.empty_like(big)
        with torch.cuda.stream(busy):
            for _ in range(12):
                big2.copy_(big)
        lean_busy = ext.furthest_point_sampling(xyz, 2048)
        torch.cuda.synchronize()
    finally:
        L.eda_fps_set_background(0)
    assert torch.equal(lean, wide) and torch.equal(lean_busy, wide) and ext.fps_status() == 0
    monkeypatch.setenv("EDA_FPS_BACKGROUND", "1")
    assert torch.equal(ext.furthest_point_sampling(xyz, 2048), wide)
