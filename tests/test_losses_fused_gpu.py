"""csrc/loss.hip (eda_amd/losses_fused.py) against the torch form of the same loss (eda_amd/losses.py, itself pinned to the
reference's models/losses.py goldens in tests/test_losses.py): values of every part and head, the assignment, and the gradients
with respect to every prediction tensor, on padded targets with empty scenes, full scenes and scattered valid slots."""
import numpy as np
import pytest
import torch

import loss_fixtures as LF

pytestmark = pytest.mark.gpu


def _criterion(soft=True, names=("boxes", "labels", "contrastive_align")):
    from eda_amd import losses
    return losses, losses.SetCriterion(losses.HungarianMatcher(1, 5, 2, soft), losses=list(names), eos_coef=0.1, temperature=0.07)


def _end_points(seed, dataset, B, Q, L, counts):
    """Scattered valid slots; WITHOUT the seed-objectness inputs: most of these slots own no seed, their top-k is a pick among equal
    distances, which torch.topk and the kernel are both free to make differently (test_seed_objectness_* cover that branch)."""
    ep = LF.make_end_points(seed, B=B, Q=Q, L=L, dataset=dataset)
    ep.pop("seeds_obj_cls_logits")
    rng = np.random.default_rng(seed + 100)
    mask = torch.zeros(B, LF.G)
    for b, n in enumerate(counts):
        mask[b, torch.from_numpy(rng.permutation(LF.G)[:n].astype(np.int64))] = 1          # scattered valid slots
    ep["box_label_mask"] = mask
    ep = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ep.items()}
    ep["tokenized"] = {"attention_mask": ep["tokenized"]["attention_mask"].cuda()}
    return ep


def _run(ep, crit, losses, fused, monkeypatch):
    monkeypatch.setenv("EDA_FUSED_LOSS", "1" if fused else "0")
    ep = dict(ep)
    for k in GRAD_KEYS:
        ep[k] = ep[k].detach().clone().requires_grad_(True)
    loss, ep = losses.compute_hungarian_loss(ep, 2, crit, query_points_obj_topk=5)
    loss.backward()
    return loss, ep


GRAD_KEYS = [k for k in LF.GRAD_KEYS if k != "seeds_obj_cls_logits"]


@pytest.mark.parametrize("dataset,B,Q,L,counts", [
    ("scanrefer", 4, 256, 40, (0, 1, 37, 132)),
    ("sr3d", 3, 32, 20, (2, 0, 5)),
    ("scanrefer", 8, 256, 130, (3, 1, 2, 8, 1, 1, 4, 2)),
    ("scanrefer", 2, 64, 256, (1, 6)),
    ("sr3d", 1, 4, 21, (3,)),                       # one scene, four queries, an odd token count
    ("scanrefer", 5, 68, 33, (4, 4, 0, 0, 1)),      # query count not a multiple of 64 / of the chunk size
    ("sr3d", 2, 128, 255, (9, 2)),                  # LDS nearly full: the column sums meet a few waves at a time
])
def test_fused_loss_equals_torch_form(dataset, B, Q, L, counts, monkeypatch):
    from eda_amd import losses_fused
    losses, crit = _criterion()
    ep0 = _end_points(21, dataset, B, Q, L, counts)
    assert losses_fused.usable(ep0, crit, None)
    lt, et = _run(ep0, crit, losses, False, monkeypatch)
    lf, ef = _run(ep0, crit, losses, True, monkeypatch)
    for p in LF.PREFIXES:
        assert torch.equal(et[f"{p}assign"], ef[f"{p}assign"]), p
        for k in ("loss_ce", "loss_bbox", "loss_giou", "loss_sem_align"):
            torch.testing.assert_close(ef[f"{p}_{k}"], et[f"{p}_{k}"], rtol=3e-5, atol=1e-5, msg=lambda m: f"{p}{k}: {m}")
    for k in ("loss_ce", "loss_bbox", "loss_giou", "loss_sem_align"):
        torch.testing.assert_close(ef[k], et[k], rtol=3e-5, atol=1e-5, msg=lambda m: f"{k}: {m}")
    torch.testing.assert_close(lf, lt, rtol=2e-5, atol=1e-5)
    for k in GRAD_KEYS:
        e, g = et[k].grad, ef[k].grad
        torch.testing.assert_close(g, e, rtol=5e-4, atol=5e-6 * (float(e.abs().max()) + 1), msg=lambda m: f"grad {k}: {m}")


@pytest.mark.parametrize("names", [("boxes",), ("labels",), ("boxes", "contrastive_align")])
def test_fused_loss_subsets(names, monkeypatch):
    losses, crit = _criterion(names=names)
    ep0 = _end_points(5, "scanrefer", 3, 32, 20, (2, 1, 3))
    lt, et = _run(ep0, crit, losses, False, monkeypatch)
    lf, ef = _run(ep0, crit, losses, True, monkeypatch)
    torch.testing.assert_close(lf, lt, rtol=2e-5, atol=1e-5)
    for k in GRAD_KEYS:
        e, g = et[k].grad, ef[k].grad
        if e is None:
            assert g is None or float(g.abs().max()) == 0.0, k
        else:
            torch.testing.assert_close(g, e, rtol=5e-4, atol=5e-6 * (float(e.abs().max()) + 1), msg=lambda m: f"grad {k}: {m}")


@pytest.mark.parametrize("soft", [True, False])
def test_match_cost_and_slots(soft):
    from eda_amd import losses_fused
    losses, crit = _criterion(soft)
    torch.manual_seed(3)
    P, B, Q, G, C = 3, 4, 96, 20, 256
    logits = torch.randn(P * B, Q, C, device="cuda")
    pred = torch.cat([torch.randn(P * B, Q, 3, device="cuda"), torch.rand(P * B, Q, 3, device="cuda") + 0.2], -1)
    tgt = torch.cat([torch.randn(B, G, 3, device="cuda"), torch.rand(B, G, 3, device="cuda") + 0.2], -1)
    pmap = torch.rand(B, G, C, device="cuda") * (torch.rand(B, G, C, device="cuda") < 0.1)
    labels = torch.randint(0, C, (B, G), device="cuda")
    nt = torch.tensor([0, 20, 7, 1], dtype=torch.int32, device="cuda")
    got = losses_fused.match_cost(logits, pred, tgt, nt, crit.matcher, pmap, labels)
    rep = lambda t: t.unsqueeze(0).expand(P, *t.shape).reshape(P * t.shape[0], *t.shape[1:])       # noqa: E731
    exp = crit.matcher.cost_matrix(logits, pred, rep(tgt), rep(pmap), rep(labels))
    real = (torch.arange(G, device="cuda")[None, :] < rep(nt)[:, None])[:, None, :].expand_as(exp)
    torch.testing.assert_close(got[real], exp[real], rtol=2e-5, atol=2e-5)
    assert float(got[~real].abs().max()) == 0.0
    # the same assignment from either matrix, and its inverse
    a1 = losses.solve_assignment(got, rep(nt))
    a2 = losses.solve_assignment(torch.nan_to_num(exp), rep(nt))
    assert torch.equal(a1, a2)
    tq = losses_fused.match_slots(a1, nt, Q).cpu()
    a = a1.cpu()
    exp_tq = torch.full((P * B, Q), -1, dtype=torch.long)
    for pb in range(P * B):
        for g in range(int(nt[pb % B])):
            exp_tq[pb, a[pb, g]] = g
    assert torch.equal(tq, exp_tq)


def test_fused_path_is_what_the_bench_loss_runs(monkeypatch):
    """usable() takes the bench's --loss hungarian inputs; a hand-given assignment or EDA_FUSED_LOSS=0 keeps the torch form."""
    from eda_amd import losses_fused
    losses, crit = _criterion()
    ep = _end_points(9, "scanrefer", 2, 32, 20, (1, 2))
    assert losses_fused.usable(ep, crit, None)
    assert not losses_fused.usable(ep, crit, {"last_": None})
    monkeypatch.setenv("EDA_FUSED_LOSS", "0")
    assert not losses_fused.usable(ep, crit, None)


def _objectness_inputs(seed, B, K, npts, ninst, valid_slots):
    rng = np.random.default_rng(seed)
    G = LF.G
    ep = {"seed_inds": torch.from_numpy(np.stack([rng.permutation(npts)[:K] for _ in range(B)]).astype(np.int32)),
          "seed_xyz": torch.from_numpy(rng.uniform(-2, 2, (B, K, 3)).astype(np.float32)),
          "seeds_obj_cls_logits": torch.from_numpy(rng.standard_normal((B, 1, K)).astype(np.float32) * 2),
          "point_instance_label": torch.from_numpy(rng.integers(-1, ninst, (B, npts)).astype(np.int64)),
          "center_label": torch.from_numpy(rng.uniform(-2, 2, (B, G, 3)).astype(np.float32)),
          "size_gts": torch.from_numpy(rng.uniform(0.2, 1.5, (B, G, 3)).astype(np.float32))}
    mask = torch.zeros(B, G)
    for b in range(B):
        mask[b, list(valid_slots[b % len(valid_slots)])] = 1
    ep["box_label_mask"] = mask
    return {k: v.cuda() for k, v in ep.items()}


@pytest.mark.parametrize("topk,K", [(4, 1024), (5, 1024), (1, 200), (8, 2048)])
def test_seed_objectness_equals_torch_form(topk, K):
    """Every valid slot owns at least topk seeds (ids 0..9 over >= 200 seeds; slot G-1 owns the background seeds), so the k nearest are
    unique: value and gradient equal the torch form's."""
    from eda_amd import losses, losses_fused
    ep = _objectness_inputs(4, 4, K, 6000, 10, [(0, 3, 4, 9), (), (1, 2, LF.G - 1), tuple(range(10))])
    assert losses_fused.seed_objectness_usable(ep, topk)
    out = []
    for fn in (losses.compute_points_obj_cls_loss_hard_topk, losses_fused.seed_objectness_loss):
        e = dict(ep)
        e["seeds_obj_cls_logits"] = ep["seeds_obj_cls_logits"].clone().requires_grad_(True)
        v = fn(e, topk)
        v.backward()
        out.append((v.detach(), e["seeds_obj_cls_logits"].grad))
    torch.testing.assert_close(out[1][0], out[0][0], rtol=2e-5, atol=1e-7)
    torch.testing.assert_close(out[1][1], out[0][1], rtol=2e-4, atol=1e-9)


def test_seed_objectness_equal_distances_take_the_lowest_index():
    """Instances with fewer seeds than k (and slots that own none): the rest of the k come from the seeds at distance 100, lowest index
    first; those that are not background become positives, exactly as in the reference's scatter."""
    from eda_amd import losses_fused
    topk, B, K, G = 5, 3, 96, LF.G
    ep = _objectness_inputs(8, B, K, 300, 40, [(0, 1, 2, 17, 39, 100), (5,), (G - 1, 7, 8)])
    e = dict(ep)
    e["seeds_obj_cls_logits"] = ep["seeds_obj_cls_logits"].clone().requires_grad_(True)
    losses_fused.seed_objectness_loss(e, topk).backward()
    got = (e["seeds_obj_cls_logits"].grad.view(B, K) < 0).cpu().numpy()       # the focal loss falls with the logit exactly on positives
    inst = torch.gather(ep["point_instance_label"], 1, ep["seed_inds"].long()).cpu().numpy()
    xyz, c, z = ep["seed_xyz"].cpu().numpy(), ep["center_label"].cpu().numpy(), ep["size_gts"].cpu().numpy()
    mask = ep["box_label_mask"].cpu().numpy()
    exp = np.zeros((B, K), bool)
    fewer = 0
    for b in range(B):
        owner = np.where(inst[b] < 0, G - 1, inst[b])
        for g in np.nonzero(mask[b])[0]:
            d = ((xyz[b] - c[b, g]) / (z[b, g] + np.float32(1e-6))).astype(np.float32)
            v = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2] + np.float32(1e-6)).astype(np.float32)
            v = np.where(owner == g, v, np.float32(100.0))
            fewer += int((owner == g).sum() < topk)
            exp[b, np.argsort(v, kind="stable")[:topk]] = True
        exp[b] &= inst[b] >= 0
    assert fewer >= 4
    assert (got == exp).all()


def test_compact_targets_one_launch():
    from eda_amd import losses, losses_fused
    rng = np.random.default_rng(2)
    B, G, W = 5, LF.G, 256
    mask = torch.from_numpy((rng.uniform(0, 1, (B, G)) < 0.2).astype(np.float32))
    mask[1] = 0
    mask[2] = 1
    center6 = torch.from_numpy(rng.standard_normal((B, G, 6)).astype(np.float32)).cuda()      # only [..., :3] is the centre
    size = torch.from_numpy(rng.uniform(0.2, 1.5, (B, G, 3)).astype(np.float32)).cuda()
    labels = torch.from_numpy(rng.integers(0, 2 ** 40, (B, G))).cuda()
    maps = [torch.from_numpy(rng.standard_normal((B, G, W)).astype(np.float32)).cuda() for _ in range(5)]
    mask = mask.cuda()
    nt, valid, boxes, lab, outs, nb = losses_fused.compact_targets(mask, center6, size, labels, maps)
    ent, evalid, packed = losses.compact_targets(mask, torch.cat([center6[:, :, :3], size], -1), labels, *maps)
    assert torch.equal(nt, ent) and torch.equal(valid.bool(), evalid)
    assert float(nb) == float(mask.sum())
    for got, exp in zip([boxes, lab] + outs, packed):
        keep = evalid.reshape(B, G, *([1] * (exp.dim() - 2))).expand_as(exp)
        assert torch.equal(got[keep], exp[keep])
        assert float(got[~keep].abs().max() if (~keep).any() else 0) == 0.0
