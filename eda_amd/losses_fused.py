"""The training loss on the GPU as a handful of launches (csrc/loss.hip) instead of ~690 (eda_amd/losses.py's batched torch
form: tools/loss_census.py).  Same public behaviour as ``losses.compute_hungarian_loss`` -- the reference's
``compute_hungarian_loss`` (models/losses.py:650-738): same ``end_points`` keys, same numbers to fp32 rounding
(tests/test_losses_fused_gpu.py against the torch form's values AND autograd gradients; tests/test_losses.py against the
reference's own goldens) -- and used by it automatically on CUDA tensors (``EDA_FUSED_LOSS=0`` keeps the torch form).

What is fused: the matching cost (only the real target slots: the solver reads nothing else), the slot -> query inverse of
the assignment, and the three criterion losses, each as ONE forward launch that also forms its gradient with respect to the
predictions and ONE backward launch that scales it.  What stays in torch: compacting the padded targets (no gradient, a dozen
launches), stacking the heads, the query x token product of the alignment loss (a batched GEMM and its two backward GEMMs), the
seed-objectness focal loss with its top-k (``losses.compute_points_obj_cls_loss_hard_topk``), the final weighted sum.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib


def _s():
    return torch.cuda.current_stream().cuda_stream


def _parr(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def match_cost(logits, pred_boxes, tgt_boxes, ntargets, matcher, pmap=None, labels=None):
    """(PB,Q,G) cost of HungarianMatcher.cost_matrix for the real target slots (zeros beyond); targets per scene (B,...)."""
    PB, Q, C = logits.shape
    B, G = tgt_boxes.shape[:2]
    cost = torch.empty((PB, Q, G), dtype=torch.float32, device=logits.device)
    soft = bool(matcher.soft_token)
    pm = pmap.contiguous() if soft else None
    with torch.cuda.device(logits.device):
        rc = _lib.lib().eda_match_cost_f32(
            logits.data_ptr(), pred_boxes.data_ptr(), tgt_boxes.data_ptr(), pm.data_ptr() if soft else None,
            pm.stride(1) if soft else 0, None if soft else labels.contiguous().data_ptr(), ntargets.data_ptr(), PB, B, Q, G, C,
            float(matcher.cost_class), float(matcher.cost_bbox), float(matcher.cost_giou), cost.data_ptr(), _s())
    _lib.check(rc, "eda_match_cost_f32")
    return cost


def match_slots(assign, ntargets, Q):
    """tq (PB,Q) int64: the target slot matched to each query or -1."""
    PB, G = assign.shape
    B = ntargets.shape[0]
    tq = torch.empty((PB, Q), dtype=torch.int64, device=assign.device)
    with torch.cuda.device(assign.device):
        rc = _lib.lib().eda_match_slots_i64(assign.data_ptr(), ntargets.data_ptr(), PB, B, Q, G, tq.data_ptr(), _s())
    _lib.check(rc, "eda_match_slots_i64")
    return tq


class _BoxLoss(Function):
    @staticmethod
    def forward(ctx, pred, tgt, assign, valid_u8, tq, nb):
        PB, Q = pred.shape[:2]
        B, G = tgt.shape[:2]
        pred = pred if pred.stride(2) == 1 else pred.contiguous()
        dev = pred.device
        l1 = torch.empty((PB,), dtype=torch.float32, device=dev)
        gi = torch.empty((PB,), dtype=torch.float32, device=dev)
        g1 = torch.empty((PB, G, 6), dtype=torch.float32, device=dev)
        g2 = torch.empty((PB, G, 6), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().eda_box_loss_fwd_f32(pred.data_ptr(), pred.stride(0), pred.stride(1), tgt.data_ptr(), assign.data_ptr(),
                                                 valid_u8.data_ptr(), nb.data_ptr(), PB, B, Q, G, l1.data_ptr(), gi.data_ptr(),
                                                 g1.data_ptr(), g2.data_ptr(), _s())
        _lib.check(rc, "eda_box_loss_fwd_f32")
        ctx.save_for_backward(g1, g2, tq, nb)
        ctx.dims = (PB, Q, G)
        return l1, gi

    @staticmethod
    def backward(ctx, w1, w2):
        g1, g2, tq, nb = ctx.saved_tensors
        PB, Q, G = ctx.dims
        w1 = (w1 if w1 is not None else torch.full((PB,), 0.0, device=g1.device)).contiguous()
        w2 = (w2 if w2 is not None else torch.full((PB,), 0.0, device=g1.device)).contiguous()
        dpred = torch.empty((PB, Q, 6), dtype=torch.float32, device=g1.device)
        with torch.cuda.device(g1.device):
            rc = _lib.lib().eda_box_loss_bwd_f32(g1.data_ptr(), g2.data_ptr(), tq.data_ptr(), w1.data_ptr(), w2.data_ptr(),
                                                 nb.data_ptr(), PB, Q, G, dpred.data_ptr(), _s())
        _lib.check(rc, "eda_box_loss_bwd_f32")
        return dpred, None, None, None, None, None


def _scaled(grad0, w, nb):
    PB = grad0.shape[0]
    out = torch.empty_like(grad0)
    w = w.contiguous()
    with torch.cuda.device(grad0.device):
        rc = _lib.lib().eda_scale_by_scene_f32(grad0.data_ptr(), w.data_ptr(), nb.data_ptr(), PB, grad0.numel() // PB,
                                               out.data_ptr(), _s())
    _lib.check(rc, "eda_scale_by_scene_f32")
    return out


class _PosAlign(Function):
    @staticmethod
    def forward(ctx, logits, tq, nb, eos, weights, *maps):
        PB, Q, C = logits.shape
        B, G = maps[0].shape[:2]
        logits = logits.contiguous()
        loss = torch.empty((PB,), dtype=torch.float32, device=logits.device)
        grad0 = torch.empty_like(logits)
        w = (ctypes.c_float * 4)(*weights)
        with torch.cuda.device(logits.device):
            rc = _lib.lib().eda_pos_align_fwd_f32(logits.data_ptr(), tq.data_ptr(), _parr(maps), w, maps[0].stride(0),
                                                  maps[0].stride(1), nb.data_ptr(), PB, B, Q, G, C, float(eos), loss.data_ptr(),
                                                  grad0.data_ptr(), _s())
        _lib.check(rc, "eda_pos_align_fwd_f32")
        ctx.save_for_backward(grad0, nb)
        return loss

    @staticmethod
    def backward(ctx, w):
        grad0, nb = ctx.saved_tensors
        return (_scaled(grad0, w, nb),) + (None,) * 8


class _SemAlign(Function):
    @staticmethod
    def forward(ctx, logits, tq, nb, eos, attn_mask, *maps):
        PB, Q, L = logits.shape
        B, G = maps[0].shape[:2]
        logits = logits.contiguous()
        loss = torch.empty((PB,), dtype=torch.float32, device=logits.device)
        grad0 = torch.empty_like(logits)
        with torch.cuda.device(logits.device):
            rc = _lib.lib().eda_sem_align_fwd_f32(logits.data_ptr(), tq.data_ptr(), _parr(maps), maps[0].stride(0), maps[0].stride(1),
                                                  attn_mask.data_ptr(), nb.data_ptr(), PB, B, Q, G, L, float(eos), loss.data_ptr(),
                                                  grad0.data_ptr(), _s())
        _lib.check(rc, "eda_sem_align_fwd_f32")
        ctx.save_for_backward(grad0, nb)
        return loss

    @staticmethod
    def backward(ctx, w):
        grad0, nb = ctx.saved_tensors
        return (_scaled(grad0, w, nb),) + (None,) * 9


_KEYS = ["positive_map", "modify_positive_map", "pron_positive_map", "other_entity_map", "rel_positive_map"]


def usable(end_points, set_criterion, assign):
    """Does the fused path take this call?  CUDA tensors, the three criterion losses (any subset), <= 512 token classes, a query count that is a multiple of 4, and a
    (queries x tokens) tile that fits the alignment kernel's LDS."""
    if assign is not None or os.environ.get("EDA_FUSED_LOSS", "1") == "0":
        return False
    t = end_points.get("last_sem_cls_scores")
    if t is None or not t.is_cuda or t.dtype != torch.float32 or t.shape[-1] > 512 or t.shape[1] % 4:
        return False
    if any(n not in ("boxes", "labels", "contrastive_align") for n in set_criterion.losses):
        return False
    if "contrastive_align" in set_criterion.losses:
        if "proj_tokens" not in end_points:
            return False
        if not _lib.lib().eda_sem_align_supported(int(t.shape[1]), int(end_points["proj_tokens"].shape[1])):
            return False
    return True


def compute_hungarian_loss(end_points, num_decoder_layers, set_criterion, query_points_obj_topk=5):
    from . import losses as LT
    prefixes = ["proposal_", "last_"] + [f"{i}head_" for i in range(num_decoder_layers - 1)]
    P = len(prefixes)
    crit = set_criterion
    gt_box = torch.cat([end_points["center_label"][:, :, 0:3], end_points["size_gts"]], dim=-1)
    nt, valid, packed = LT.compact_targets(end_points["box_label_mask"], gt_box, end_points["sem_cls_label"],
                                           *[end_points[k] for k in _KEYS])
    tgt_boxes, tgt_labels = packed[0].contiguous(), packed[1]
    maps = {k: packed[2 + i].contiguous() for i, k in enumerate(_KEYS)}
    B, G = tgt_boxes.shape[:2]
    stack = lambda name: torch.cat([end_points[f"{p}{name}"] for p in prefixes], dim=0)         # noqa: E731
    logits = stack("sem_cls_scores")
    pred_boxes = torch.cat([stack("center"), stack("pred_size")], dim=-1)
    PB, Q, C = logits.shape
    nb = LT.count_boxes(nt)
    with torch.no_grad():
        cost = match_cost(logits.detach(), pred_boxes.detach(), tgt_boxes, nt, crit.matcher, maps["positive_map"], tgt_labels)
        assign = LT.solve_assignment(cost, nt.repeat(P))
        tq = match_slots(assign, nt, Q)
    valid_u8 = valid.contiguous().view(torch.uint8)
    rows = {}
    if "boxes" in crit.losses:
        rows["loss_bbox"], rows["loss_giou"] = _BoxLoss.apply(pred_boxes, tgt_boxes, assign, valid_u8, tq, nb)
    if "labels" in crit.losses:
        w = (0.625, 0.125, 0.125, 0.125) if end_points["language_dataset"][0] == "sr3d" else (0.6, 0.2, 0.2, 0.1)
        rows["loss_ce"] = _PosAlign.apply(logits, tq, nb, crit.eos_coef, w, maps["positive_map"], maps["modify_positive_map"],
                                          maps["pron_positive_map"], maps["rel_positive_map"])
    if "contrastive_align" in crit.losses:
        pq = stack("proj_queries").view(P, B, Q, -1)
        sim = (torch.matmul(pq, end_points["proj_tokens"].transpose(-1, -2)) / crit.temperature).view(PB, Q, -1)
        am = end_points["tokenized"]["attention_mask"].contiguous()
        am = am if am.dtype == torch.int64 else am.long()
        rows["loss_sem_align"] = _SemAlign.apply(sim, tq, nb, crit.eos_coef, am, *[maps[k] for k in _KEYS])
    names = ["loss_ce", "loss_bbox", "loss_giou", "loss_sem_align"]
    present = [n for n in names if n in rows]
    tot = {n: 0 for n in names}
    if present:
        per_head = torch.stack([rows[n] for n in present]).view(len(present), P, B).sum(2)       # (parts, heads)
        totals = per_head.sum(1)
        for j, n in enumerate(present):
            tot[n] = totals[j]
            for i, prefix in enumerate(prefixes):
                end_points[f"{prefix}_{n}"] = per_head[j, i]
    for i, prefix in enumerate(prefixes):
        end_points[f"{prefix}assign"] = assign[i * B:(i + 1) * B]
    qp = (LT.compute_points_obj_cls_loss_hard_topk(end_points, query_points_obj_topk)
          if "seeds_obj_cls_logits" in end_points else 0.0)
    weight = 0.5 if end_points["language_dataset"][0] == "scanrefer" else 1
    loss = 8 * qp + 1.0 / (num_decoder_layers + 1) * (
        weight * tot["loss_ce"] + 5 * tot["loss_bbox"] + tot["loss_giou"] + weight * tot["loss_sem_align"])
    end_points.update(loss_ce=tot["loss_ce"], loss_bbox=tot["loss_bbox"], loss_giou=tot["loss_giou"],
                      query_points_generation_loss=qp, loss_sem_align=tot["loss_sem_align"], loss=loss)
    return loss, end_points
