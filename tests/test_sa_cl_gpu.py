"""GPU parity of the channels-last set-abstraction kernels (csrc/sa_cl.hip) against
plain fp32 torch references of the same ops, and of the fused SA fast path against
the generic (reference-shaped) path of the same module at full size."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_bn_relu(z, gamma, beta, rm, rv, eps, momentum, training, pool):
    y = F.batch_norm(z, rm, rv, gamma, beta, training, momentum, eps)
    y = F.relu(y)
    if pool > 1:
        R, C = y.shape
        y = y.view(R // pool, pool, C).max(dim=1)[0]
    return y


@pytest.mark.parametrize("R,C,pool,training", [
    (4096, 64, 1, True), (8192, 128, 16, True), (2048, 288, 1, True), (6144, 256, 32, True),
    (640, 16, 64, True), (4096, 64, 1, False), (2048, 32, 16, False), (100352, 128, 64, True),
])
def test_bn_relu_forward_backward(R, C, pool, training):
    from eda_amd import sa_ops
    torch.manual_seed(R + C)
    dev = "cuda"
    z = (torch.randn(R, C, device=dev) * 1.7 + 0.4).requires_grad_(True)
    gamma = (torch.rand(C, device=dev) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, device=dev) * 0.2).requires_grad_(True)
    rm0 = torch.randn(C, device=dev) * 0.1
    rv0 = torch.rand(C, device=dev) + 0.5
    rm_a, rv_a, rm_b, rv_b = rm0.clone(), rv0.clone(), rm0.clone(), rv0.clone()
    out = sa_ops.BNReLUCL.apply(z, gamma, beta, rm_a, rv_a, 1e-5, 0.1, training, pool)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    got = [out.detach(), z.grad.clone(), gamma.grad.clone(), beta.grad.clone()]
    for t in (z, gamma, beta):
        t.grad = None
    exp_out = _ref_bn_relu(z, gamma, beta, rm_b, rv_b, 1e-5, 0.1, training, pool)
    (exp_out * w).sum().backward()
    exp = [exp_out.detach(), z.grad, gamma.grad, beta.grad]
    for name, g, e in zip(["out", "dz", "dgamma", "dbeta"], got, exp):
        scale = e.abs().max().item() + 1e-9
        err = (g - e).abs().max().item()
        assert err <= 2e-4 * scale + 1e-6, (name, err, scale)
    if training:
        torch.testing.assert_close(rm_a, rm_b, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rv_a, rv_b, rtol=1e-4, atol=1e-6)
    else:
        assert torch.equal(rm_a, rm0) and torch.equal(rv_a, rv0)


@pytest.mark.parametrize("B,N,m,ns,C", [(2, 500, 40, 8, 0), (2, 3000, 128, 16, 3), (2, 2048, 256, 32, 128), (1, 700, 33, 5, 7)])
def test_group_concat_cl_vs_reference_ops(B, N, m, ns, C):
    """Fused gather/centre/scale/concat rows == QueryAndGroup of the reference-shaped ops."""
    from eda_amd import sa_ops, pointnet2_utils as PU
    rng = np.random.default_rng(N + C)
    dev = "cuda"
    xyz = torch.from_numpy(rng.uniform(-2, 2, (B, N, 3)).astype(np.float32)).to(dev)
    new_xyz = xyz[:, :m].contiguous()
    feats = torch.randn(B, C, N, device=dev, requires_grad=True) if C else None
    idx = PU.ball_query(0.6, ns, xyz, new_xyz)
    rows = sa_ops.GroupConcatCL.apply(xyz, new_xyz, feats.transpose(1, 2).contiguous() if C else None,
                                      idx, 0.6, True)
    qg = PU.QueryAndGroup(0.6, ns, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=True)
    ref = qg(xyz, new_xyz, feats)                                    # (B, 3+C, m, ns)
    ref_rows = ref.permute(0, 2, 3, 1).reshape(B, m * ns, 3 + C)
    assert torch.equal(rows, ref_rows)       # same fp32 ops (sub, mul by 1/r), pure copies otherwise
    if C:
        w = torch.randn_like(rows)
        fcl = feats.detach().transpose(1, 2).contiguous().requires_grad_(True)
        (sa_ops.GroupConcatCL.apply(xyz, new_xyz, fcl, idx, 0.6, True) * w).sum().backward()
        (ref_rows * w).sum().backward()
        torch.testing.assert_close(fcl.grad.transpose(1, 2), feats.grad, rtol=1e-4, atol=1e-4)


def test_pointwise_linear_split_k_gradient():
    from eda_amd import sa_ops
    torch.manual_seed(1)
    a = torch.randn(131072, 24, device="cuda", requires_grad=True)
    w = torch.randn(40, 24, 1, 1, device="cuda", requires_grad=True)
    z = sa_ops.PointwiseLinearCL.apply(a, w)
    g = torch.randn_like(z)
    z.backward(g)
    ga, gw = a.grad.clone(), w.grad.clone()
    a.grad = None; w.grad = None
    (a @ w.view(40, 24).t()).backward(g)
    torch.testing.assert_close(ga, a.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gw, w.grad, rtol=1e-4, atol=1e-2)     # 131072-term sums in fp32


@pytest.mark.parametrize("training", [False, True])
def test_sa_fast_path_matches_generic_path_full_size(training):
    """SA1 at BASELINE.json's size (B=2 here, N=50 000, 2048 centres x 64 neighbours): the
    channels-last fast path and the reference-shaped generic path of the same module agree."""
    from eda_amd import synthetic
    from eda_amd.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(0)
    pc = torch.from_numpy(synthetic.batch([3, 4], 50000)).cuda()
    xyz = pc[..., :3].contiguous()
    feats = pc[..., 3:].transpose(1, 2).contiguous()
    sa = PointnetSAModuleVotes(npoint=2048, radius=0.2, nsample=64, mlp=[3, 64, 64, 128], use_xyz=True,
                               normalize_xyz=True).cuda().train(training)
    import copy
    sb = copy.deepcopy(sa)
    f1 = feats.clone().requires_grad_(True)
    f2 = feats.clone().requires_grad_(True)
    x1, o1, i1 = sa(xyz, f1)
    sb._fast_path = lambda _xyz: False          # force the generic path
    x2, o2, i2 = sb(xyz, f2)
    assert torch.equal(i1, i2) and torch.equal(x1, x2)
    torch.testing.assert_close(o1, o2, rtol=2e-4, atol=2e-5)
    w = torch.randn_like(o1)
    (o1 * w).sum().backward()
    (o2 * w).sum().backward()
    # max-pool arg-max / ReLU decisions can flip on near-ties between two fp32 evaluation orders.  A flip re-routes
    # the gradient of isolated elements -- and, in training mode, ONE flipped element of a hidden layer changes that
    # channel's BatchNorm-backward constants (sum of gy), which shifts the channel's gradient in EVERY row by a
    # fraction of a per cent (observed: one flip in layer 1 -> 204 of 100 000 points off by ~1 %; which element sits
    # on the tie depends on the summation grouping of the statistics, e.g. on the launch grid).  So: nearly all
    # elements agree tightly, the rest agree loosely, and the two gradients point the same way.
    bad = ((f1.grad - f2.grad).abs() > 1e-4 + 1e-3 * f2.grad.abs()).float().mean().item()
    assert bad <= (2e-3 if training else 1e-4), bad
    assert (f1.grad - f2.grad).abs().max().item() <= 0.05 * f2.grad.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(f1.grad.flatten().double(), f2.grad.flatten().double(), dim=0).item()
    assert cos >= 0.99999, cos
    for (n1, p1), (n2, p2) in zip(sa.named_parameters(), sb.named_parameters()):
        scale = p2.grad.abs().max().item() + 1e-9
        assert (p1.grad - p2.grad).abs().max().item() <= (1e-2 if training else 2e-3) * scale, n1     # (same flips)
    if training:
        for (n1, b1), (n2, b2) in zip(sa.named_buffers(), sb.named_buffers()):
            torch.testing.assert_close(b1.float(), b2.float(), rtol=1e-4, atol=1e-5, msg=n1)


@pytest.mark.parametrize("R,C,p", [(2048, 288, 0.3), (100, 64, 0.5), (4096, 128, 0.1)])
def test_bn_relu_fused_dropout(R, C, p):
    """Dropout behind the ReLU inside the small-row BN kernel: keep-rate, 1/(1-p) scaling, and a
    backward that uses exactly the forward's mask (checked against torch ops with that mask)."""
    from eda_amd import attention
    from eda_amd.sa_ops import BNReLUCL
    torch.manual_seed(R + C)
    z = (torch.randn(R, C, device="cuda") * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(C, device="cuda") + 0.5).requires_grad_(True)
    beta = (torch.randn(C, device="cuda") * 0.3).requires_grad_(True)
    w = torch.randn(R, C, device="cuda")
    attention.dropout_state("cuda").fill_(5)

    def run(pd, salt):
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        return BNReLUCL.apply(z, gamma, beta, rm, rv, 1e-5, 0.1, True, 1, pd, salt)
    a = run(0.0, 0).detach()
    out = run(p, 77)
    assert torch.equal(out, run(p, 77))                     # same step + site -> same mask
    assert not torch.equal(out, run(p, 78))
    pos = a > 0
    kept = (out != 0) & pos
    rate = 1.0 - kept.sum().item() / pos.sum().item()
    assert abs(rate - p) < 0.02, rate
    torch.testing.assert_close(out[kept], a[kept] / (1 - p), rtol=1e-5, atol=1e-6)
    assert (out[~kept] == 0).all()
    got = torch.autograd.grad((out * w).sum(), [z, gamma, beta])
    mask = kept.float() / (1 - p)
    ref = torch.relu(torch.nn.functional.batch_norm(z, None, None, gamma, beta, True, 0.1, 1e-5))
    exp = torch.autograd.grad((ref * mask * w).sum(), [z, gamma, beta])
    for g, e, name in zip(got, exp, ["dz", "dgamma", "dbeta"]):
        scale = e.abs().max().item() + 1e-9
        assert (g - e).abs().max().item() <= 2e-4 * scale, (name, (g - e).abs().max().item(), scale)


def test_bn_relu_fused_dropout_refuses_large_rows():
    from eda_amd import _lib
    from eda_amd.sa_ops import BNReLUCL
    R = _lib.lib().eda_bn_relu_dropout_max_rows() + 64
    z = torch.randn(R, 64, device="cuda")
    g, b = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
    with pytest.raises(RuntimeError, match="fused dropout"):
        BNReLUCL.apply(z, g, b, torch.zeros(64, device="cuda"), torch.ones(64, device="cuda"), 1e-5, 0.1, True, 1, 0.3, 1)


@pytest.mark.parametrize("training", [False, True])
def test_backbone_full_size_activations_and_gradients_vs_oracle_backed_cpu_model(training, oracle, monkeypatch):
    """BASELINE.json configs[1] at full size (50 000 points per scene, B=2 to keep the CPU side short):
    every `end_points` tensor of Pointnet2Backbone and the gradients of all its parameters, computed by
    the HIP path, against THE SAME module run on the CPU with the oracle (oracle/eda_oracle.c, the line-
    cited restatement of the reference's CUDA ops) as its op backend and stock torch for everything else.
    Indices must be identical; activations within 1e-4 per element + 1e-5 of the tensor's scale (apart
    from <= 1e-4 of the elements).
    Parameter gradients: two fp32 evaluations of a 12-layer ReLU / max-pool network disagree on the
    DECISIONS (sign of a pre-activation, arg-max of a neighbourhood) wherever the two candidates are
    closer than the activations' own 1e-5..1e-4 agreement -- ~1e-4 of 10^5..10^8 decisions -- and every
    flipped decision moves one O(1) term of a bias / weight gradient (measured: max error / max |g| up to
    4e-3 in eval mode, 7e-2 with batch statistics, on tensors whose fp64-checked small-size versions
    agree to 2e-4, tests/test_sa_fused_gpu.py).  So the full-size gradient check is direction + norm:
    cosine >= 0.9995 and max error <= 1e-2 (eval) / 1e-1 (train) of the tensor's scale."""
    import copy
    from eda_amd import synthetic, pointnet2_utils
    from eda_amd.backbone_module import Pointnet2Backbone
    torch.manual_seed(1)
    pc = torch.from_numpy(synthetic.batch([5, 6], 50000))
    bb_cpu = Pointnet2Backbone(input_feature_dim=3, width=1).train(training)
    bb_gpu = copy.deepcopy(bb_cpu).cuda()
    ep_g = bb_gpu(pc.cuda(), {})
    w = {k: torch.randn(v.shape, generator=torch.Generator().manual_seed(i)) for i, (k, v) in enumerate(sorted(ep_g.items()))
         if v.dtype == torch.float32 and "features" in k}
    sum((ep_g[k] * w[k].cuda()).sum() for k in w).backward()
    torch.cuda.synchronize()
    monkeypatch.setattr(pointnet2_utils, "_ext", oracle)          # CPU side: oracle ops (test infrastructure)
    ep_c = bb_cpu(pc, {})
    sum((ep_c[k] * w[k]).sum() for k in w).backward()
    assert sorted(ep_g) == sorted(ep_c)
    for k in sorted(ep_c):
        a, b = ep_g[k].detach().cpu(), ep_c[k].detach()
        if b.dtype in (torch.int32, torch.int64):
            assert torch.equal(a.long(), b.long()), k
        elif "xyz" in k:
            assert torch.equal(a, b), k
        else:
            tol = 1e-4 * b.abs() + 1e-5 * b.abs().max()
            bad = ((a - b).abs() > tol).float().mean().item()
            assert bad <= 1e-4, (k, bad, (a - b).abs().max().item(), b.abs().max().item())
    import os
    worst = []
    for (n, pg), (_, pc_) in zip(bb_gpu.named_parameters(), bb_cpu.named_parameters()):
        g, c = pg.grad.cpu(), pc_.grad
        scale = c.abs().max().item() + 1e-12
        rel = (g - c).abs().max().item() / scale
        cos = torch.nn.functional.cosine_similarity(g.flatten().double(), c.flatten().double(), dim=0).item()
        worst.append((rel, n))
        if os.environ.get("EDA_TEST_VERBOSE"):
            print(f"  GRAD {n}: max err / max |g| = {rel:.2e} (scale {scale:.3e}) cosine {cos:.6f}")
        assert cos >= 0.9995, (n, cos)
    assert max(worst)[0] <= (1e-1 if training else 1e-2), sorted(worst, reverse=True)[:5]
    if training:
        for (n, bg), (_, bc) in zip(bb_gpu.named_buffers(), bb_cpu.named_buffers()):
            torch.testing.assert_close(bg.cpu().float(), bc.float(), rtol=1e-4, atol=1e-5, msg=n)


def test_backbone_with_precomputed_geometry_equals_the_plain_forward():
    """Pointnet2Backbone.geometry(xyz) (samplings, ball queries, 3-NN: functions of the coordinates alone, which bench.py
    computes for the next batch on a second stream) handed back through forward(..., geometry=) must reproduce the
    plain forward bit for bit: every end_points tensor and every parameter gradient."""
    import copy
    from eda_amd import synthetic
    from eda_amd.backbone_module import Pointnet2Backbone
    torch.manual_seed(3)
    bb = Pointnet2Backbone(input_feature_dim=3, width=1).cuda().train()
    bb2 = copy.deepcopy(bb)
    pc = torch.from_numpy(synthetic.batch(range(2), 20000)).cuda()
    ep1 = bb(pc)
    geo = bb2.geometry(pc[..., 0:3].contiguous())
    assert set(geo) == {f"sa{i}_{k}" for i in (1, 2, 3, 4) for k in ("inds", "xyz", "idx")} | {"fp1_idx", "fp1_weight", "fp2_idx", "fp2_weight"}
    ep2 = bb2(pc, geometry=geo)
    for k in ep1:
        assert torch.equal(ep1[k], ep2[k]), k
    w = torch.randn_like(ep1["fp2_features"])
    (ep1["fp2_features"] * w).sum().backward()
    (ep2["fp2_features"] * w).sum().backward()
    for (n, p), (_, q) in zip(bb.named_parameters(), bb2.named_parameters()):
        assert p.grad is not None and (p.grad - q.grad).abs().max().item() <= 1e-5 * (p.grad.abs().max().item() + 1e-12), n
    # and SA1's indices alone (the reference's own `inds` argument)
    ep3 = copy.deepcopy(bb2)(pc, sa1_inds=geo["sa1_inds"])
    assert torch.equal(ep3["sa1_inds"], ep1["sa1_inds"]) and torch.equal(ep3["sa2_xyz"], ep1["sa2_xyz"])
