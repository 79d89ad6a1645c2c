"""Sibling ThreeLayerMLPs of a prediction head in grouped launches (eda_amd/grouped.py) against the
per-module path of the same head (which the reference-generated goldens pin, tests/model_cases.py).

Same arithmetic per output element (same k order inside the GEMM kernels, same BatchNorm kernel), so
outputs agree to fp32 rounding of the differently tiled sums; gradients within 1e-5 of the tensor's max.
"""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _head(objectness):
    from eda_amd.modules import ClsAgnosticPredictHead
    torch.manual_seed(3)
    h = ClsAgnosticPredictHead(256, 1, 256, 288, objectness=objectness, heading=False, compute_sem_scores=True).cuda()
    for m in h.modules():                      # non-trivial BatchNorm parameters / statistics
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    return h


def _run(h, feats, base, grouped, train, dropout):
    from eda_amd import modules
    modules._GROUPED_HEADS = grouped
    h.train(train)
    for m in h.modules():
        if isinstance(m, torch.nn.Dropout):
            m.train(train and dropout)
    f = feats.clone().requires_grad_(True)
    ep = {}
    h(f.transpose(1, 2), base, ep, prefix="p_", features_rows=f)
    keys = sorted(k for k in ep if k != "p_base_xyz")
    loss = sum((ep[k] * torch.linspace(0.5, 1.5, ep[k].numel(), device="cuda").view_as(ep[k])).sum() for k in keys)
    for p in h.parameters():
        p.grad = None
    loss.backward()
    modules._GROUPED_HEADS = True
    return {k: ep[k].detach() for k in keys}, f.grad, {n: p.grad.clone() for n, p in h.named_parameters()}, \
        {n: b.clone() for n, b in h.named_buffers()}


@pytest.mark.parametrize("objectness", [False, True])
@pytest.mark.parametrize("train", [False, True])
def test_grouped_head_matches_per_module_path(objectness, train):
    feats = torch.randn(8, 256, 288, device="cuda")
    base = torch.randn(8, 256, 3, device="cuda")
    h1 = _head(objectness)
    h2 = copy.deepcopy(h1)
    o1, g1, p1, b1 = _run(h1, feats, base, True, train, False)
    o2, g2, p2, b2 = _run(h2, feats, base, False, train, False)
    for k in o1:
        torch.testing.assert_close(o1[k], o2[k], rtol=1e-5, atol=1e-5, msg=k)
    torch.testing.assert_close(g1, g2, rtol=1e-4, atol=1e-5 * g2.abs().max().item())
    for n in p1:
        torch.testing.assert_close(p1[n], p2[n], rtol=1e-4, atol=2e-5 * p2[n].abs().max().item() + 1e-7, msg=n)
    for n in b1:
        torch.testing.assert_close(b1[n].float(), b2[n].float(), rtol=1e-5, atol=1e-6, msg=n)


def test_grouped_head_with_flat_parameters_and_deferred_weight_gradients():
    """The training configuration: parameters in the flat buffer (siblings adjacent -> packed weights, one
    input-gradient GEMM), weight gradients through the deferred queue."""
    from eda_amd.parallel import FlatParams
    from eda_amd import modules
    feats = torch.randn(8, 256, 288, device="cuda")
    base = torch.randn(8, 256, 3, device="cuda")
    h1 = _head(False)
    h2 = copy.deepcopy(h1)
    o2, g2, p2, _ = _run(h2, feats, base, False, True, False)
    fp = FlatParams(h1)
    h1.train(True)
    for m in h1.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    f = feats.clone().requires_grad_(True)
    ep = {}
    with fp.deferred_wgrad():
        h1(f.transpose(1, 2), base, ep, prefix="p_", features_rows=f)
        keys = sorted(k for k in ep if k != "p_base_xyz")
        loss = sum((ep[k] * torch.linspace(0.5, 1.5, ep[k].numel(), device="cuda").view_as(ep[k])).sum() for k in keys)
        loss.backward()
    fp.collect_grads()
    torch.testing.assert_close(f.grad, g2, rtol=1e-4, atol=1e-5 * g2.abs().max().item())
    for (n, p), gv in zip([(n, p) for n, p in h1.named_parameters()], [None] * 0 or []):
        pass
    got = {n: fp._locate_grad(p.data) for n, p in h1.named_parameters()}
    for n in p2:
        torch.testing.assert_close(got[n], p2[n], rtol=1e-4, atol=2e-5 * p2[n].abs().max().item() + 1e-7, msg=n)


def test_grouped_head_dropout_keep_rate_and_determinism():
    from eda_amd import attention
    feats = torch.randn(8, 256, 288, device="cuda")
    base = torch.randn(8, 256, 3, device="cuda")
    h = _head(False)
    attention.dropout_state("cuda").fill_(11)
    o1, g1, _, _ = _run(h, feats, base, True, True, True)
    o1b, g1b, _, _ = _run(h, feats, base, True, True, True)
    for k in o1:
        assert torch.equal(o1[k], o1b[k])          # same step, same call sites -> same masks
    assert torch.equal(g1, g1b)
    attention.dropout_state("cuda").fill_(12)
    o2, _, _, _ = _run(h, feats, base, True, True, True)
    assert not torch.equal(o1["p_sem_cls_scores"], o2["p_sem_cls_scores"])
    assert torch.isfinite(g1).all()


@pytest.mark.parametrize("G,R,C,MW", [(14, 2048, 288, 3), (1, 37, 288, 1), (16, 1000, 64, 4), (5, 2049, 132, 2), (2, 3, 288, 3)])
def test_tiny_out_bwd_multi_vs_fp64(G, R, C, MW):
    """eda_tiny_out_bwd_multi_f32 (the heads' 1-4-channel last layers: input, weight and bias gradients of several layers
    in one streaming pass) against fp64, ragged row counts, strided operands, repeats bit-identical."""
    from eda_amd.heads_batched import _tiny_out_bwd
    torch.manual_seed(G * 1000 + R + C + MW)
    dys = [torch.randn(R, MW, device="cuda") for _ in range(G)]
    Ws = [torch.randn(MW, C, device="cuda") for _ in range(G)]
    apack = torch.randn(R, G * C + 8, device="cuda")
    a_blocks = [apack[:, g * C:(g + 1) * C] for g in range(G)]                  # column blocks of a packed matrix
    outs = []
    for rep in range(2):
        dpack = torch.full((R, G * C), float("nan"), device="cuda")
        da_blocks = [dpack[:, g * C:(g + 1) * C] for g in range(G)]
        dWs, dbs = _tiny_out_bwd(dys, Ws, a_blocks, da_blocks, [torch.empty(MW, device="cuda")] * G)
        outs.append((dpack.clone(), [w.clone() for w in dWs], [b.clone() for b in dbs]))
    assert torch.equal(outs[0][0], outs[1][0]) and all(torch.equal(x, y) for x, y in zip(outs[0][1], outs[1][1]))
    dpack, dWs, dbs = outs[0]
    for g in range(G):
        eda = dys[g].double() @ Ws[g].double()
        assert (dpack[:, g * C:(g + 1) * C].double() - eda).abs().max().item() <= 1e-5 * max(eda.abs().max().item(), 1.0)
        eW = dys[g].double().t() @ a_blocks[g].double()
        bound = 2e-5 * (dys[g].abs().double().t() @ a_blocks[g].abs().double()).max().item() + 1e-6
        assert (dWs[g].double() - eW).abs().max().item() <= bound
        eb = dys[g].double().sum(0)
        assert (dbs[g].double() - eb).abs().max().item() <= 2e-5 * dys[g].abs().double().sum(0).max().item() + 1e-6


@pytest.mark.parametrize("nmat,R,G,C,p,train", [(7, 2048, 3, 288, 0.3, True), (1, 100, 1, 16, 0.0, True), (8, 513, 4, 32, 0.3, True),
                                                (6, 2048, 1, 288, 0.0, False)])
def test_bn_relu_grouped_bwd_multi_equals_single_launches(nmat, R, G, C, p, train):
    """eda_bn_relu_grouped_bwd_multi_f32 (several packed matrices in one launch) = eda_bn_relu_grouped_bwd_f32 per matrix,
    bit for bit (the same kernel body)."""
    import ctypes
    from eda_amd import _lib
    from eda_amd.attention import dropout_state
    from eda_amd.grouped import _parr, _stream
    from eda_amd.heads_batched import _bn_bwd_multi
    torch.manual_seed(nmat + R + G + C)
    dev = torch.device("cuda", 0)
    zs = [torch.randn(R, G * C, device=dev) for _ in range(nmat)]
    das = [torch.randn(R, G * C, device=dev) for _ in range(nmat)]
    gammas = [[torch.rand(C, device=dev) + 0.5 for _ in range(G)] for _ in range(nmat)]
    stats = []
    for z, gs in zip(zs, gammas):
        mean, var = z.mean(0), z.var(0, unbiased=False)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        sc = torch.cat(gs) * rstd
        stats.append(torch.stack([mean, rstd, sc, torch.randn(G * C, device=dev) * 0.1 - mean * sc]).contiguous())
    cfgs = [(G, C, train, p, [1000 * m + g + 1 for g in range(G)]) for m in range(nmat)]
    dzs, dgbs = _bn_bwd_multi(das, zs, stats, gammas, cfgs)
    seed = dropout_state(dev) if p > 0 else None
    for m in range(nmat):
        dz = torch.empty_like(zs[m]); dgb = torch.empty((2, G * C), device=dev)
        salt_arr = (ctypes.c_uint * G)(*cfgs[m][4])
        st = stats[m]
        rc = _lib.lib().eda_bn_relu_grouped_bwd_f32(
            das[m].data_ptr(), zs[m].data_ptr(), R, G, C, _parr(gammas[m]), st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(),
            st[3].data_ptr(), int(train), dgb[0].data_ptr(), dgb[1].data_ptr(), dz.data_ptr(), float(p),
            seed.data_ptr() if seed is not None else None, salt_arr, _stream())
        assert rc == 0
        assert torch.equal(dz, dzs[m]) and torch.equal(dgb, dgbs[m])
        assert torch.isfinite(dz).all()
