"""The deterministic mode (eda_set_deterministic / EDA_DETERMINISTIC=1, csrc/scatter_det.hip): every fp32-atomic gradient
scatter becomes an ordered per-owner sum -- the counterpart of the reference's `cudnn.deterministic = True`
(train_dist_mod.py:342-344).  Op level: two calls give the same bits, and the bits are the ORACLE's (it accumulates in
ascending (j, k) order too, oracle/eda_oracle.c).  Model level: graph replay against eager launches with the strict
bounds the atomics noise did not allow."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.fixture
def det():
    from eda_amd import deterministic
    deterministic.enable(True)
    yield deterministic
    deterministic.enable(False)


def _cloud(B, n, seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.uniform(-1, 1, (B, n, 3)).astype(np.float32))


@pytest.mark.parametrize("B,C,n,m,ns", [(2, 5, 300, 40, 8), (2, 131, 2048, 256, 32), (1, 259, 1024, 512, 16), (3, 64, 5000, 70, 64)])
def test_group_points_grad_is_ordered_and_equals_the_oracle_bitwise(det, oracle, B, C, n, m, ns):
    from eda_amd import ext
    g = torch.Generator().manual_seed(n + m)
    idx = torch.randint(0, n, (B, m, ns), generator=g, dtype=torch.int32)
    idx[:, :, ns // 2:] = idx[:, :, :1]                 # ball-query style padding: many repeats of one target
    go = torch.randn(B, C, m, ns, generator=g)
    want = oracle.group_points_grad(go, idx, n)
    a = ext.group_points_grad(go.cuda(), idx.cuda(), n)
    b = ext.group_points_grad(go.cuda(), idx.cuda(), n)
    assert torch.equal(a, b)
    assert torch.equal(a.cpu(), want)
    det.enable(False)                                    # the atomic form agrees to rounding
    c = ext.group_points_grad(go.cuda(), idx.cuda(), n)
    torch.testing.assert_close(c.cpu(), want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,C,n,m", [(2, 288, 1024, 256), (1, 3, 50, 7), (2, 300, 512, 600)])
def test_gather_points_grad_is_ordered_and_equals_the_oracle_bitwise(det, oracle, B, C, n, m):
    from eda_amd import ext
    g = torch.Generator().manual_seed(n * 3 + m)
    idx = torch.randint(0, n, (B, m), generator=g, dtype=torch.int32)
    go = torch.randn(B, C, m, generator=g)
    want = oracle.gather_points_grad(go, idx, n)
    a = ext.gather_points_grad(go.cuda(), idx.cuda(), n)
    assert torch.equal(a, ext.gather_points_grad(go.cuda(), idx.cuda(), n))
    assert torch.equal(a.cpu(), want)


@pytest.mark.parametrize("B,C,n,m", [(2, 256, 512, 256), (2, 256, 1024, 512), (1, 7, 100, 3), (2, 40, 3000, 2000)])
def test_three_interpolate_grad_is_ordered_and_equals_the_oracle_bitwise(det, oracle, B, C, n, m):
    from eda_amd import ext
    g = torch.Generator().manual_seed(n + 7 * m)
    idx = torch.randint(0, m, (B, n, 3), generator=g, dtype=torch.int32)
    w = torch.rand(B, n, 3, generator=g)
    w = w / w.sum(-1, keepdim=True)
    go = torch.randn(B, C, n, generator=g)
    want = oracle.three_interpolate_grad(go, idx, w, m)
    a = ext.three_interpolate_grad(go.cuda(), idx.cuda(), w.cuda(), m)
    assert torch.equal(a, ext.three_interpolate_grad(go.cuda(), idx.cuda(), w.cuda(), m))
    assert torch.equal(a.cpu(), want)


def test_embedding_weight_gradient_is_ordered(det):
    from eda_amd.nn_utils import embedding_rows
    torch.manual_seed(0)
    emb = torch.nn.Embedding(485, 768).cuda()
    ids = torch.randint(0, 485, (8, 132), device="cuda")
    ids[:, 60:] = ids[:, :1]
    w = torch.randn(8, 132, 768, device="cuda")
    grads = []
    for _ in range(2):
        emb.weight.grad = None
        (embedding_rows(emb, ids) * w).sum().backward()
        grads.append(emb.weight.grad.clone())
    assert torch.equal(grads[0], grads[1])
    ref = torch.zeros(485, 768, dtype=torch.float64, device="cuda").index_add_(0, ids.reshape(-1), w.reshape(-1, 768).double())
    torch.testing.assert_close(grads[0].double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("level", ["sa2", "sa3"])
def test_fused_sa_backward_feature_gradient_is_reproducible(det, level):
    """d(features) of a gathering SA stack (SA2: sa_gather_layer_bwd_kernel + the scatter; SA3: the input-gradient product
    + the scatter): identical bits in two calls with the ordered sums, and equal to the atomic form to fp32 rounding."""
    from eda_amd import sa_ops, pointnet2_utils as PU
    B, N, m, ns, C, chans, radius = (2, 2048, 1024, 32, 128, [128, 128, 256], 0.4) if level == "sa2" else \
        (2, 1024, 512, 16, 256, [128, 128, 256], 0.8)
    rng = np.random.default_rng(1)
    xyz = torch.from_numpy(rng.uniform(-2, 2, (B, N, 3)).astype(np.float32)).cuda()
    new_xyz = xyz[:, :m].contiguous()
    idx = PU.ball_query(radius, ns, xyz, new_xyz)
    ch = [3 + C] + chans
    torch.manual_seed(3)
    Ws = [torch.randn(ch[l + 1], ch[l], 1, 1, device="cuda").mul_(0.1).requires_grad_(True) for l in range(3)]
    gs = [torch.ones(c, device="cuda", requires_grad=True) for c in ch[1:]]
    bs = [torch.zeros(c, device="cuda", requires_grad=True) for c in ch[1:]]
    feats = torch.randn(B, N, C, device="cuda", requires_grad=True)
    dout = None

    def run():
        nonlocal dout
        running = [(torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")) for c in ch[1:]]
        cfg = dict(gather=True, radius=radius, normalize_xyz=True, pool=ns, training=True, eps=1e-5, momentum=0.1, running=running)
        params = []
        for W, g_, b_ in zip(Ws, gs, bs):
            params += [W, g_, b_]
        out = sa_ops.FusedMLP.apply(cfg, None, xyz, new_xyz, feats, idx, *params)
        if dout is None:
            dout = torch.randn_like(out)
        feats.grad = None
        out.backward(dout)
        return feats.grad.clone()
    a, b = run(), run()
    assert torch.equal(a, b)
    det.enable(False)
    c = run()
    assert (a - c).abs().max().item() <= 1e-5 * c.abs().max().item()


def test_graph_replay_equals_eager_in_the_deterministic_mode(det):
    """VERDICT r04 item 6 / ADVICE r04: with the atomics gone, no three-seed retry and no loosened later steps -- every
    one of 4 steps (3 replays: a buffer that goes stale from the second replay on shows) within 1e-5 of the eager run, the
    parameters after 4 optimizer steps within 1e-5 of the largest.  Both runs defer their weight gradients (the same
    kernels; eager-immediate vs deferred is another summation order, tested separately)."""
    import check_graph_vs_eager as C
    losses, bad, rel = C.compare(steps=4, scenes=2, points=20000, tokens=24, verbose=False, seed=0, num_queries=64,
                                 num_decoder_layers=2, defer_in_eager=True)
    e, g = losses["eager"], losses["graph"]
    print(losses)
    assert all(abs(x - y) <= 1e-5 * abs(x) for x, y in zip(e, g)), losses
    assert rel <= 1e-5, rel


def test_two_runs_of_five_training_steps_agree(det):
    """Two fresh processes-worth of state (same seed) trained for five eager steps: every loss and every parameter agree
    to 1e-6 relative -- in the default mode the fifth loss differs in the third digit (tests/test_graph_gpu.py).  (Bit
    identity holds whenever no BatchNorm column sum -- fp64 atomics over ~2000 workgroup partials, the one order-dependent
    reduction left, DESIGN.md section 7 -- lands within 1e-16 of an fp32 rounding boundary; the number of bit-identical
    runs is printed.)"""
    import copy
    import bench
    import check_graph_vs_eager as C
    from eda_amd.parallel import FlatParams
    dev = torch.device("cuda", 0)
    base = C.make(0, dev, num_queries=64, num_decoder_layers=2)
    inputs = bench.make_inputs(0, 2, dev, 20000, 24)
    outs = []
    for _ in range(2):
        model = copy.deepcopy(base)
        flat = FlatParams(model)
        opt = torch.optim.AdamW(list(flat.groups.values()), lr=1e-4, weight_decay=5e-4, fused=True)
        hist = []
        for _s in range(5):
            loss = bench.synthetic_loss(model(inputs))
            with flat.deferred_wgrad():
                loss.backward()
            flat.collect_grads()
            flat.clip_grad_norm_(0.1)
            opt.step()
            hist.append(float(loss.detach()))
        outs.append((hist, flat.flat_param.clone()))
    (h1, p1), (h2, p2) = outs
    print("losses", h1, h2, "bit-identical parameters:", bool(torch.equal(p1, p2)))
    assert all(abs(x - y) <= 1e-6 * abs(x) for x, y in zip(h1, h2)), (h1, h2)
    assert (p1 - p2).abs().max().item() <= 1e-6 * p1.abs().max().item()
