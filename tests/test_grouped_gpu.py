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
