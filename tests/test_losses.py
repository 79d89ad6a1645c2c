"""The training loss (eda_amd/losses.py) against goldens produced by the
REFERENCE's models/losses.py on the inputs of tests/loss_fixtures.py (tools/gen_golden_loss.py),
and the device-side assignment solver (csrc/lsa.hip) against the numpy restatement
(oracle/lsa_ref.py), which itself is pinned against scipy's linear_sum_assignment."""
import os

import numpy as np
import pytest
import torch

import loss_fixtures as LF

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _criterion():
    from eda_amd import losses
    matcher = losses.HungarianMatcher(1, 0, 2, True)
    return losses, losses.SetCriterion(matcher, losses=["boxes", "labels", "contrastive_align"], eos_coef=0.1,
                                       temperature=0.07)


def _oracle_assign(losses, crit, ep):
    """Assignments per prefix from the oracle solver on the product's cost matrix (CPU tests)."""
    from oracle import lsa_ref
    gt_box = torch.cat([ep["center_label"][:, :, :3], ep["size_gts"]], -1)
    nt, valid, (boxes, pmap) = losses.compact_targets(ep["box_label_mask"], gt_box, ep["positive_map"])
    out = {}
    for p in LF.PREFIXES:
        cost = crit.matcher.cost_matrix(ep[f"{p}sem_cls_scores"].detach(),
                                        torch.cat([ep[f"{p}center"], ep[f"{p}pred_size"]], -1).detach(), boxes, pmap)
        a = torch.full((cost.shape[0], cost.shape[2]), -1, dtype=torch.int32)
        for b in range(cost.shape[0]):
            n = int(nt[b])
            a[b, :n] = torch.from_numpy(lsa_ref.assign_targets(cost[b, :, :n].double().numpy()))
        out[p] = a
    return out


def _f(x):
    return float(np.asarray(x.detach().cpu() if torch.is_tensor(x) else x).reshape(-1)[0])


def _check_against_golden(ep, g, loss, rtol):
    for k in ("loss_ce", "loss_bbox", "loss_giou", "loss_sem_align", "query_points_generation_loss"):
        np.testing.assert_allclose(_f(ep[k]), _f(g[k]), rtol=rtol, atol=1e-5, err_msg=k)
    for p in LF.PREFIXES:
        for k in ("loss_ce", "loss_bbox", "loss_giou", "loss_sem_align"):
            np.testing.assert_allclose(_f(ep[f"{p}_{k}"]), _f(g[f"{p}_{k}"]), rtol=rtol, atol=1e-5,
                                       err_msg=p + k)
    np.testing.assert_allclose(_f(loss), _f(g["loss"]), rtol=rtol)


@pytest.mark.parametrize("name,seed", [("scanrefer", 11), ("sr3d", 12)])
def test_loss_math_matches_reference_goldens_cpu(name, seed):
    """Host logic on CPU tensors, assignment from the oracle: every loss part, the total and the
    gradients w.r.t. all prediction tensors equal the reference's."""
    losses, crit = _criterion()
    g = np.load(os.path.join(GOLD, f"loss_{name}.npz"))
    ep = LF.make_end_points(seed, dataset=name)
    for k in LF.GRAD_KEYS:
        ep[k].requires_grad_(True)
    assign = _oracle_assign(losses, crit, ep)
    loss, ep = losses.compute_hungarian_loss(ep, 2, crit, query_points_obj_topk=5, assign=assign)
    _check_against_golden(ep, g, loss, 2e-5)
    loss.backward()
    for k in LF.GRAD_KEYS:
        e = g["grad_" + k]
        np.testing.assert_allclose(ep[k].grad.numpy(), e, rtol=2e-4, atol=2e-6 * (np.abs(e).max() + 1), err_msg=k)
    # the oracle's assignment is the reference's (scipy's) assignment
    pairs = g["last_match_pairs"]
    for b in range(pairs.shape[0]):
        n = int((pairs[b, :, 0] >= 0).sum())
        exp = np.empty(n, np.int64)
        exp[pairs[b, :n, 1]] = pairs[b, :n, 0]
        assert (assign["last_"][b, :n].numpy() == exp).all()


def test_oracle_solver_is_scipy():
    from scipy.optimize import linear_sum_assignment
    from oracle.lsa_ref import assign_targets
    rng = np.random.default_rng(0)
    for it in range(200):
        Q = int(rng.integers(1, 48)); T = int(rng.integers(1, Q + 1))
        c = rng.standard_normal((Q, T)).astype(np.float32)
        if it % 7 == 0:
            c = np.round(c * 4) / 4 + rng.uniform(0, 1e-3, c.shape).astype(np.float32)   # near-ties
        r, cc = linear_sum_assignment(c)
        exp = np.empty(T, np.int64); exp[cc] = r
        assert (assign_targets(c) == exp).all()


def test_matcher_refuses_cpu():
    from eda_amd import losses
    with pytest.raises(RuntimeError, match="CPU not supported"):
        losses.solve_assignment(torch.zeros(1, 4, 2), torch.ones(1, dtype=torch.int32))


@pytest.mark.gpu
def test_lsa_kernel_vs_oracle():
    from eda_amd import losses
    from oracle.lsa_ref import assign_targets
    rng = np.random.default_rng(5)
    for B, Q, G in [(8, 256, 132), (3, 32, 8), (5, 1, 1), (2, 300, 40), (4, 1024, 16)]:
        cost = torch.from_numpy(rng.standard_normal((B, Q, G)).astype(np.float32))
        nt = torch.from_numpy(rng.integers(0, min(Q, G, 12) + 1, B).astype(np.int32))
        nt[0] = min(Q, G, 12)
        got = losses.solve_assignment(cost.cuda(), nt.cuda()).cpu().numpy()
        for b in range(B):
            n = int(nt[b])
            exp = assign_targets(cost[b, :, :n].numpy()) if n else np.zeros(0, np.int64)
            assert (got[b, :n] == exp).all(), (B, Q, G, b)
            assert (got[b, n:] == -1).all()
    # a full 132-target scene, and a strided (transposed-storage) cost
    cost = torch.from_numpy(rng.standard_normal((1, 256, 132)).astype(np.float32))
    got = losses.solve_assignment(cost.cuda(), torch.tensor([132], dtype=torch.int32).cuda()).cpu().numpy()[0]
    assert (got == assign_targets(cost[0].numpy())).all()
    ct = cost.cuda().transpose(1, 2).contiguous().transpose(1, 2)
    assert (losses.solve_assignment(ct, torch.tensor([132], dtype=torch.int32).cuda()).cpu().numpy()[0] == got).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name,seed", [("scanrefer", 11), ("sr3d", 12)])
def test_full_loss_on_device_matches_reference_goldens(name, seed):
    """Matching solved by the HIP kernel, no host round trip: same losses, gradients and
    (query, target) pairs as the reference's scipy path."""
    losses, crit = _criterion()
    g = np.load(os.path.join(GOLD, f"loss_{name}.npz"))
    ep = LF.make_end_points(seed, dataset=name)
    ep = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ep.items()}
    ep["tokenized"] = {"attention_mask": ep["tokenized"]["attention_mask"].cuda()}
    for k in LF.GRAD_KEYS:
        ep[k].requires_grad_(True)
    loss, ep = losses.compute_hungarian_loss(ep, 2, crit, query_points_obj_topk=5)
    _check_against_golden({k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in ep.items()}, g,
                          loss.detach().cpu(), 5e-5)
    loss.backward()
    for k in LF.GRAD_KEYS:
        e = g["grad_" + k]
        np.testing.assert_allclose(ep[k].grad.cpu().numpy(), e, rtol=5e-4, atol=5e-6 * (np.abs(e).max() + 1), err_msg=k)
    # drop-in matcher API: list-of-dict targets -> [(queries ascending, targets)]
    B = ep["box_label_mask"].shape[0]
    m = ep["box_label_mask"].bool()
    gt_box = torch.cat([ep["center_label"], ep["size_gts"]], -1)
    tgt = [{"labels": ep["sem_cls_label"][b, m[b]], "boxes": gt_box[b, m[b]], "positive_map": ep["positive_map"][b, m[b]]}
           for b in range(B)]
    ind = crit.matcher({"pred_logits": ep["last_sem_cls_scores"].detach(),
                        "pred_boxes": torch.cat([ep["last_center"], ep["last_pred_size"]], -1).detach()}, tgt)
    pairs = g["last_match_pairs"]
    for b, (i, j) in enumerate(ind):
        n = len(i)
        assert (i.numpy() == pairs[b, :n, 0]).all() and (j.numpy() == pairs[b, :n, 1]).all()
        assert (pairs[b, n:] == -1).all()
