"""The training loss on the GPU as a handful of launches (csrc/loss.hip) instead of ~690 (eda_amd/losses.py's batched torch
form: tools/loss_census.py).  Same public behaviour as ``losses.compute_hungarian_loss`` -- the reference's
``compute_hungarian_loss`` (models/losses.py:650-738): same ``end_points`` keys, same numbers to fp32 rounding
(tests/test_losses_fused_gpu.py against the torch form's values AND autograd gradients; tests/test_losses.py against the
reference's own goldens) -- and used by it automatically on CUDA tensors (``EDA_FUSED_LOSS=0`` keeps the torch form).

What is fused: compacting the padded targets (valid slots first) with the box count, the matching cost (only the real target
slots: the solver reads nothing else), the slot -> query inverse of the assignment, the three criterion losses and the seed
objectness loss -- each ONE forward launch that also forms its gradient with respect to the predictions and ONE backward launch
that scales it -- and the final weighted sum with the per-head values (one launch each way).  What stays in torch: stacking the
heads (three ``cat``) and the query x token product of the alignment loss (a batched GEMM and its two backward GEMMs).
Two documented differences from the torch form: where the reference's ``torch.topk`` meets EQUAL distances in the objectness loss
(an instance with fewer seeds than k) its choice is the library's, here the lowest index; and the per-head / per-part values written
into ``end_points`` are plain values (the reference back-propagates ``loss`` only).
"""
import ctypes
import os

import torch
import torch.distributed as dist
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


def _scaled(grad0, w, nb, per_part=None):
    """grad0 (PB, ...) times the upstream gradient w (PB, S) of its scene's part / num_boxes."""
    PB = grad0.shape[0]
    per = grad0.numel() // PB
    out = torch.empty_like(grad0)
    w = w.contiguous()
    S = w.numel() // PB
    with torch.cuda.device(grad0.device):
        rc = _lib.lib().eda_scale_by_scene_f32(grad0.data_ptr(), w.data_ptr(), nb.data_ptr(), PB, per, per_part or per, S,
                                               out.data_ptr(), _s())
    _lib.check(rc, "eda_scale_by_scene_f32")
    return out


class _PosAlign(Function):
    @staticmethod
    def forward(ctx, logits, tq, nb, eos, weights, *maps):
        PB, Q, C = logits.shape
        B, G = maps[0].shape[:2]
        logits = logits.contiguous()
        S = 8 if Q >= 64 else 1               # a scene's rows over S workgroups: (PB, S) partial sums
        loss = torch.empty((PB, S), dtype=torch.float32, device=logits.device)
        grad0 = torch.empty_like(logits)
        w = (ctypes.c_float * 4)(*weights)
        with torch.cuda.device(logits.device):
            rc = _lib.lib().eda_pos_align_fwd_f32(logits.data_ptr(), tq.data_ptr(), _parr(maps), w, maps[0].stride(0),
                                                  maps[0].stride(1), nb.data_ptr(), PB, B, Q, G, C, S, float(eos), loss.data_ptr(),
                                                  grad0.data_ptr(), _s())
        _lib.check(rc, "eda_pos_align_fwd_f32")
        ctx.save_for_backward(grad0, nb)
        ctx.per_part = _lib.lib().eda_pos_align_chunk(Q, S) * C
        return loss

    @staticmethod
    def backward(ctx, w):
        grad0, nb = ctx.saved_tensors
        return (_scaled(grad0, w, nb, ctx.per_part),) + (None,) * 8


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


def _words(t):
    return t.element_size() // 4


def compact_targets(mask, center, size, labels, maps):
    """losses.compact_targets + count_boxes as one launch: (ntargets (B,) int32, valid (B,G) uint8, boxes (B,G,6), labels (B,G)
    int64, [maps (B,G,W)], num_boxes (1,) float32 before any all-reduce).  Rows beyond a scene's count are zero."""
    B, G = mask.shape
    dev = mask.device
    mask = mask.contiguous() if mask.dtype == torch.float32 else mask.float()
    boxes = torch.empty((B, G, 6), dtype=torch.float32, device=dev)
    lab = torch.empty((B, G), dtype=labels.dtype, device=dev)
    outs = [torch.empty((B, G, m.shape[2]), dtype=m.dtype, device=dev) for m in maps]
    # (source, destination, destination word offset, words per row)
    jobs = [(center, boxes, 0, 3), (size, boxes, 3, 3), (labels, lab, 0, _words(labels))]
    jobs += [(m, o, 0, m.shape[2] * _words(m)) for m, o in zip(maps, outs)]
    n = len(jobs)
    for src, _, _, _ in jobs:
        assert src.element_size() % 4 == 0 and (src.dim() == 2 or src.stride(2) == 1), "rows of 4-byte words expected"
    P, L, I = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
    nt = torch.empty((B,), dtype=torch.int32, device=dev)
    valid = torch.empty((B, G), dtype=torch.uint8, device=dev)
    nb = torch.empty((1,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().eda_compact_targets(
            mask.data_ptr(), n, P(*[j[0].data_ptr() for j in jobs]), L(*[j[0].stride(0) * _words(j[0]) for j in jobs]),
            L(*[j[0].stride(1) * _words(j[0]) for j in jobs]), P(*[j[1].data_ptr() for j in jobs]),
            L(*[(j[1].stride(1) if j[1].dim() > 1 else 1) * _words(j[1]) for j in jobs]), L(*[j[2] for j in jobs]),
            I(*[j[3] for j in jobs]), B, G, nt.data_ptr(), valid.data_ptr(), nb.data_ptr(), _s())
    _lib.check(rc, "eda_compact_targets")
    return nt, valid, boxes, lab, outs, nb


class _Combine(Function):
    """loss = w_obj * sum(obj) + inv * (w . totals); per_head (4, P) and totals (5,) are reported values (not differentiable)."""

    @staticmethod
    def forward(ctx, P, B, w, inv, w_obj, obj, *rows):
        dev = next(r for r in rows + (obj,) if r is not None).device
        rows = tuple(None if r is None else r.contiguous() for r in rows)
        per_head = torch.empty((4, P), dtype=torch.float32, device=dev)
        totals = torch.empty((5,), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        ptr = (ctypes.c_void_p * 4)(*[None if r is None else r.data_ptr() for r in rows])
        parts = (ctypes.c_int * 4)(*[0 if r is None else r.numel() // (P * B) for r in rows])
        wv = (ctypes.c_float * 4)(*w)
        objc = None if obj is None else obj.contiguous()
        with torch.cuda.device(dev):
            rc = _lib.lib().eda_loss_combine_fwd_f32(ptr, parts, None if objc is None else objc.data_ptr(), wv, float(inv), float(w_obj),
                                                     P, B, per_head.data_ptr(), totals.data_ptr(), loss.data_ptr(), _s())
        _lib.check(rc, "eda_loss_combine_fwd_f32")
        ctx.mark_non_differentiable(per_head, totals)
        ctx.set_materialize_grads(False)
        ctx.consts = (tuple(w), float(inv), float(w_obj), B, [None if r is None else r.shape for r in rows],
                      None if obj is None else obj.shape, dev)
        return loss, per_head, totals

    @staticmethod
    def backward(ctx, g, _ph, _tot):
        w, inv, w_obj, B, shapes, oshape, dev = ctx.consts
        g = g.contiguous()
        d_rows = [None if sh is None else torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes]
        d_obj = None if oshape is None else torch.empty(oshape, dtype=torch.float32, device=dev)
        ptr = (ctypes.c_void_p * 4)(*[None if d is None else d.data_ptr() for d in d_rows])
        n = (ctypes.c_int * 4)(*[0 if d is None else d.numel() for d in d_rows])
        with torch.cuda.device(dev):
            rc = _lib.lib().eda_loss_combine_bwd_f32(g.data_ptr(), (ctypes.c_float * 4)(*w), inv, w_obj, ptr, n,
                                                     None if d_obj is None else d_obj.data_ptr(), 0 if d_obj is None else d_obj.numel(),
                                                     _s())
        _lib.check(rc, "eda_loss_combine_bwd_f32")
        return (None, None, None, None, None, d_obj) + tuple(d_rows)


class _SeedObjectness(Function):
    @staticmethod
    def forward(ctx, logits, seed_xyz, seed_inds, instance_label, centre, size, mask, topk):
        B, K = seed_xyz.shape[:2]
        G = centre.shape[1]
        lg = logits.reshape(B, K).contiguous()
        loss = torch.empty((B,), dtype=torch.float32, device=lg.device)
        grad0 = torch.empty_like(lg)
        with torch.cuda.device(lg.device):
            rc = _lib.lib().eda_seed_objectness_fwd_f32(
                lg.data_ptr(), seed_xyz.data_ptr(), seed_inds.data_ptr(), instance_label.data_ptr(), instance_label.shape[1],
                centre.data_ptr(), centre.stride(1), size.data_ptr(), size.stride(1), mask.data_ptr(), B, K, G, int(topk),
                loss.data_ptr(), grad0.data_ptr(), _s())
        _lib.check(rc, "eda_seed_objectness_fwd_f32")
        ctx.save_for_backward(grad0)
        ctx.shape = logits.shape
        return loss

    @staticmethod
    def backward(ctx, w):
        (grad0,) = ctx.saved_tensors
        return ((grad0 * w[:, None]).view(ctx.shape),) + (None,) * 7


def seed_objectness_usable(end_points, topk):
    lg, cl, sz = end_points["seeds_obj_cls_logits"], end_points["center_label"], end_points["size_gts"]
    K = end_points["seed_xyz"].shape[1]
    return (lg.is_cuda and lg.dtype == torch.float32 and 1 <= topk <= 8 and K >= topk and K * 20 <= 150 * 1024
            and cl.stride(0) == cl.shape[1] * cl.stride(1) and cl.stride(2) == 1
            and sz.stride(0) == sz.shape[1] * sz.stride(1) and sz.stride(2) == 1)


def seed_objectness_loss(end_points, topk):
    """compute_points_obj_cls_loss_hard_topk (models/losses.py:166-228) as one launch; equal distances inside a top-k are taken
    lowest seed index first (torch.topk, which the reference calls, leaves that to the library)."""
    return seed_objectness_per_scene(end_points, topk).sum()


def seed_objectness_per_scene(end_points, topk):
    """(B,) shares of seed_objectness_loss (their sum is the reference's scalar)."""
    si = end_points["seed_inds"]
    si = si.contiguous() if si.dtype == torch.int32 else si.int()
    pil = end_points["point_instance_label"]
    pil = pil.contiguous() if pil.dtype == torch.int64 else pil.long()
    mask = end_points["box_label_mask"]
    mask = mask.contiguous() if mask.dtype == torch.float32 else mask.float()
    return _SeedObjectness.apply(end_points["seeds_obj_cls_logits"], end_points["seed_xyz"].contiguous(), si, pil,
                                 end_points["center_label"], end_points["size_gts"], mask, topk)


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
    maps = [end_points.get(k) for k in _KEYS]
    if any(m is None or m.dtype != torch.float32 or m.dim() != 3 or m.stride(2) != 1 or m.shape != maps[0].shape for m in maps):
        return False
    if maps[0].shape[2] < t.shape[-1]:                      # (the kernels read map columns [0, classes) and [0, tokens))
        return False
    for k in ("center_label", "size_gts"):
        v = end_points.get(k)
        if v is None or v.dtype != torch.float32 or v.dim() != 3 or v.stride(2) != 1 or v.shape[2] < 3:
            return False
    if end_points["sem_cls_label"].dtype != torch.int64:
        return False
    if "contrastive_align" in set_criterion.losses:
        if "proj_tokens" not in end_points:
            return False
        L = int(end_points["proj_tokens"].shape[1])
        if L > maps[0].shape[2] or not _lib.lib().eda_sem_align_supported(int(t.shape[1]), L):
            return False
    return True


def compute_hungarian_loss(end_points, num_decoder_layers, set_criterion, query_points_obj_topk=5):
    from . import losses as LT
    prefixes = ["proposal_", "last_"] + [f"{i}head_" for i in range(num_decoder_layers - 1)]
    P = len(prefixes)
    crit = set_criterion
    nt, valid_u8, tgt_boxes, tgt_labels, mp, nb = compact_targets(
        end_points["box_label_mask"], end_points["center_label"], end_points["size_gts"], end_points["sem_cls_label"],
        [end_points[k] for k in _KEYS])
    given = end_points.get("num_boxes_global")
    if given is not None:                                     # losses.global_box_count(): formed outside the (captured) step
        nb = given.to(torch.float32).reshape(1)
    elif dist.is_available() and dist.is_initialized():
        dist.all_reduce(nb)                                   # losses.py:630-636: boxes of the global batch
    maps = dict(zip(_KEYS, mp))
    B, G = tgt_boxes.shape[:2]
    stack = lambda name: torch.cat([end_points[f"{p}{name}"] for p in prefixes], dim=0)         # noqa: E731
    logits = stack("sem_cls_scores")
    pred_boxes = torch.cat([stack("center"), stack("pred_size")], dim=-1)
    PB, Q, C = logits.shape
    with torch.no_grad():
        cost = match_cost(logits.detach(), pred_boxes.detach(), tgt_boxes, nt, crit.matcher, maps["positive_map"], tgt_labels)
        assign = LT.solve_assignment(cost, nt.repeat(P))
        tq = match_slots(assign, nt, Q)
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
    for i, prefix in enumerate(prefixes):
        end_points[f"{prefix}assign"] = assign[i * B:(i + 1) * B]
    obj, qp_torch = None, None
    if "seeds_obj_cls_logits" in end_points:
        if seed_objectness_usable(end_points, query_points_obj_topk):
            obj = seed_objectness_per_scene(end_points, query_points_obj_topk)
        else:
            qp_torch = LT.compute_points_obj_cls_loss_hard_topk(end_points, query_points_obj_topk)
    names = ["loss_ce", "loss_bbox", "loss_giou", "loss_sem_align"]
    weight = 0.5 if end_points["language_dataset"][0] == "scanrefer" else 1
    tot = {n: 0 for n in names}
    qp = 0.0
    if rows or obj is not None:
        loss, per_head, totals = _Combine.apply(P, B, (weight, 5.0, 1.0, weight), 1.0 / (num_decoder_layers + 1), 8.0, obj,
                                                *[rows.get(n) for n in names])
        for j, n in enumerate(names):
            if n in rows:
                tot[n] = totals[j]
                for i, prefix in enumerate(prefixes):
                    end_points[f"{prefix}_{n}"] = per_head[j, i]
        if obj is not None:
            qp = totals[4]
    else:
        loss = 0.0
    if qp_torch is not None:
        qp = qp_torch
        loss = loss + 8 * qp
    end_points.update(loss_ce=tot["loss_ce"], loss_bbox=tot["loss_bbox"], loss_giou=tot["loss_giou"],
                      query_points_generation_loss=qp, loss_sem_align=tot["loss_sem_align"], loss=loss)
    return loss, end_points
