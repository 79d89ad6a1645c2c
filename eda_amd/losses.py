"""The training loss of EDA (reference models/losses.py) with the Hungarian matching ON THE DEVICE.

SURVEY.md §8f rank 1.  Same public names as the reference -- ``HungarianMatcher``,
``SetCriterion``, ``compute_hungarian_loss``, ``compute_points_obj_cls_loss_hard_topk``,
``generalized_box_iou3d``, ``box_cxcyczwhd_to_xyzxyz``, ``SigmoidFocalClassificationLoss`` -- and
the same numbers (tests/test_losses.py against goldens produced by the reference itself), but a
different organisation:

* the reference turns the padded ground truth of a batch into per-scene Python lists by boolean
  indexing (a device->host synchronisation per tensor), builds one (B*Q, sum T) cost matrix,
  copies it to the host and calls scipy's ``linear_sum_assignment`` per scene, for each of the 7
  prediction heads and twice (a second, auxiliary matching whose result no loss reads):
  losses.py:262-336, 617-628, 666-680.  Here targets stay PADDED (B, G, ...) with a count per
  scene, the (B, Q, G) cost is one batched computation, and the assignment is solved by
  csrc/lsa.hip for all scenes in one launch: nothing leaves the device, so the whole loss can
  live inside the captured training step.  The unused auxiliary matching is not computed.
* the three losses are written against that padded assignment (gather / scatter with masks
  instead of ``tensor[src_idx] = ...`` on concatenated lists).
* no ``torch.zeros`` / ``zeros_like`` / ``one_hot`` on the training path: they are memset nodes in
  a captured HIP graph, which ROCm 7.2 does not order reliably against the kernels around them
  (DESIGN.md); masks are built from comparisons and fill kernels instead.

``HungarianMatcher.forward(outputs, targets)`` keeps the reference's list-of-dicts signature and
result format for drop-in use and for the parity tests.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from . import _lib


# ----------------------------------------------------------------------------- boxes
def box_cxcyczwhd_to_xyzxyz(x):
    """(..., 6) centre + size -> (..., 6) min corner + max corner; sizes clamped at 1e-6 (losses.py:33-43)."""
    half = 0.5 * x[..., 3:].clamp(min=1e-6)
    return torch.cat([x[..., :3] - half, x[..., :3] + half], dim=-1)


def _volume(b):
    return (b[..., 3] - b[..., 0]) * (b[..., 4] - b[..., 1]) * (b[..., 5] - b[..., 2])


def _giou3d(a, b):
    """Generalised IoU of broadcastable corner boxes a, b (..., 6) (losses.py:46-97)."""
    lo = torch.maximum(a[..., :3], b[..., :3])
    hi = torch.minimum(a[..., 3:], b[..., 3:])
    e = (hi - lo).clamp(min=0)
    inter = e[..., 0] * e[..., 1] * e[..., 2]
    union = _volume(a) + _volume(b) - inter
    hull = (torch.maximum(a[..., 3:], b[..., 3:]) - torch.minimum(a[..., :3], b[..., :3])).clamp(min=0)
    hv = hull[..., 0] * hull[..., 1] * hull[..., 2]
    return inter / union - (hv - union) / hv


def generalized_box_iou3d(boxes1, boxes2):
    """Pairwise (N, M) generalised IoU of corner boxes (N, 6), (M, 6)."""
    return _giou3d(boxes1[:, None, :], boxes2[None, :, :])


# ------------------------------------------------------------------ seed objectness
class SigmoidFocalClassificationLoss(nn.Module):
    """Sigmoid focal loss (losses.py:100-164): alpha-balanced, (1-p_t)^gamma modulated BCE."""

    def __init__(self, gamma=2.0, alpha=0.25):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    def forward(self, input, target, weights):
        p = torch.sigmoid(input)
        alpha_w = target * self.alpha + (1 - target) * (1 - self.alpha)
        pt = target * (1.0 - p) + (1.0 - target) * p
        bce = torch.clamp(input, min=0) - input * target + torch.log1p(torch.exp(-torch.abs(input)))
        return (alpha_w * torch.pow(pt, self.gamma) * bce).squeeze(-1) * weights


def compute_points_obj_cls_loss_hard_topk(end_points, topk):
    """Focal loss on the seed-objectness logits: the `topk` seeds closest (in box-normalised
    distance) to each real ground-truth centre among the seeds of that instance are positives
    (losses.py:166-228)."""
    mask = end_points["box_label_mask"]
    seed_inds = end_points["seed_inds"].long()
    seed_xyz = end_points["seed_xyz"]
    logits = end_points["seeds_obj_cls_logits"]
    centre = end_points["center_label"][:, :, :3]
    size = end_points["size_gts"][:, :, :3]
    B, K, G = centre.shape[0], seed_xyz.shape[1], centre.shape[1]
    inst = torch.gather(end_points["point_instance_label"], 1, seed_inds)            # (B,K), <0 = background
    owner = torch.where(inst < 0, torch.full_like(inst, G - 1), inst)
    own = (owner[..., None] == torch.arange(G, device=owner.device)).to(seed_xyz.dtype)   # one-hot (B,K,G)
    d = (seed_xyz[:, :, None, :] - centre[:, None, :, :]) / (size[:, None, :, :] + 1e-6)
    dist_ = torch.sqrt((d ** 2).sum(-1) + 1e-6)
    dist_ = (dist_ * own + 100 * (1 - own)).transpose(1, 2).contiguous()             # (B,G,K)
    near = torch.topk(dist_, topk, largest=False)[1]                                 # (B,G,topk)
    m = mask[:, :, None]
    near = (near * m + (m - 1)).long().view(B, -1)            # padded slots -> -1 == the extra column K
    label = torch.full((B, K + 1), 0, dtype=torch.long, device=seed_xyz.device)      # (fill kernel, see module doc)
    label.scatter_(1, torch.where(near < 0, torch.full_like(near, K), near), 1)
    label = label[:, :K]
    label = label * (inst >= 0).to(label.dtype)
    w = torch.full((B, K), 1.0 / max(K, 1), dtype=logits.dtype, device=logits.device)   # all K seeds count
    loss = SigmoidFocalClassificationLoss()(logits.reshape(B, K, 1), label.unsqueeze(-1).to(logits.dtype), w)
    return loss.sum() / B


# ---------------------------------------------------------------------------- matcher
def solve_assignment(cost, ntargets):
    """cost (B, Q, G) fp32 on the GPU, ntargets (B,) int32: assign (B, G) int32, the query matched
    to each of the first ntargets[b] target slots (csrc/lsa.hip), -1 beyond.  No host sync."""
    if not cost.is_cuda:
        raise RuntimeError("CPU not supported: the Hungarian matching runs on the HIP library only")
    cost = cost.float()
    B, Q, G = cost.shape
    nt = ntargets.to(device=cost.device, dtype=torch.int32).contiguous()
    out = torch.empty((B, G), dtype=torch.int32, device=cost.device)
    with torch.cuda.device(cost.device):
        rc = _lib.lib().eda_lsa_f32(cost.data_ptr(), cost.stride(0), cost.stride(1), cost.stride(2), B, Q, G,
                                    nt.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_lsa_f32")
    return out


def compact_targets(mask, *tensors):
    """Move the valid target slots (mask (B,G) != 0) to the front, order kept (what the
    reference's boolean indexing does per scene).  Returns (ntargets (B,) int32, valid (B,G) bool,
    [tensors gathered the same way])."""
    valid = mask > 0
    order = torch.argsort((~valid).to(torch.int8), dim=1, stable=True)
    nt = valid.sum(1).to(torch.int32)
    G = mask.shape[1]
    front = torch.arange(G, device=mask.device)[None, :] < nt[:, None]
    out = []
    for t in tensors:
        idx = order.reshape(order.shape + (1,) * (t.dim() - 2)).expand(-1, -1, *t.shape[2:])
        out.append(torch.gather(t, 1, idx))
    return nt, front, out


class HungarianMatcher(nn.Module):
    """Minimum-cost one-to-one matching of targets to queries (losses.py:231-336), cost =
    cost_class * (-p[target tokens]) + cost_bbox * L1 + cost_giou * (-GIoU)."""

    def __init__(self, cost_class=1, cost_bbox=5, cost_giou=2, soft_token=False):
        super().__init__()
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0
        self.cost_class, self.cost_bbox, self.cost_giou, self.soft_token = cost_class, cost_bbox, cost_giou, soft_token

    @torch.no_grad()
    def cost_matrix(self, pred_logits, pred_boxes, tgt_boxes, tgt_pmap=None, tgt_labels=None):
        """(B,Q,C) logits, (B,Q,6) boxes, padded targets (B,G,6) [+ (B,G,>=C) token maps or (B,G)
        labels] -> (B,Q,G) cost."""
        prob = pred_logits.softmax(-1)
        if self.soft_token:
            cls = -torch.bmm(prob, tgt_pmap[..., :prob.shape[-1]].transpose(1, 2))
        else:
            cls = -torch.gather(prob, 2, tgt_labels[:, None, :].expand(-1, prob.shape[1], -1))
        l1 = (pred_boxes[:, :, None, :] - tgt_boxes[:, None, :, :]).abs().sum(-1)
        giou = -_giou3d(box_cxcyczwhd_to_xyzxyz(pred_boxes)[:, :, None, :],
                        box_cxcyczwhd_to_xyzxyz(tgt_boxes)[:, None, :, :])
        return self.cost_bbox * l1 + self.cost_class * cls + self.cost_giou * giou

    @torch.no_grad()
    def match_padded(self, pred_logits, pred_boxes, tgt_boxes, ntargets, tgt_pmap=None, tgt_labels=None):
        cost = self.cost_matrix(pred_logits, pred_boxes, tgt_boxes, tgt_pmap, tgt_labels)
        # padded slots may hold anything (degenerate boxes -> nan): they are never read by the solver,
        # but keep the matrix finite for tools that print it
        return solve_assignment(torch.nan_to_num(cost, nan=0.0, posinf=1e30, neginf=-1e30), ntargets)

    @torch.no_grad()
    def forward(self, outputs, targets):
        """Reference signature: targets = list (one per scene) of {"labels", "boxes", "positive_map"};
        returns [(query indices ascending, matching target indices)] as int64 CPU tensors."""
        B, Q = outputs["pred_logits"].shape[:2]
        dev = outputs["pred_logits"].device
        sizes = [len(t["boxes"]) for t in targets]
        G = max(max(sizes), 1)
        boxes = torch.zeros((B, G, 6), device=dev)
        boxes[..., 3:] = 1.0
        C = outputs["pred_logits"].shape[-1]
        pmap = torch.zeros((B, G, C), device=dev)
        labels = torch.zeros((B, G), dtype=torch.long, device=dev)
        for b, t in enumerate(targets):
            n = sizes[b]
            boxes[b, :n] = t["boxes"]
            labels[b, :n] = t["labels"]
            if self.soft_token:
                pm = t["positive_map"][..., :C]
                pmap[b, :n, :pm.shape[-1]] = pm
        nt = torch.tensor(sizes, dtype=torch.int32, device=dev)
        assign = self.match_padded(outputs["pred_logits"], outputs["pred_boxes"], boxes, nt, pmap, labels).cpu()
        out = []
        for b, n in enumerate(sizes):
            q = assign[b, :n].long()
            order = torch.argsort(q)
            out.append((q[order], order))
        return out


# -------------------------------------------------------------------------- criterion
def _per_query(tq, matched, table):
    """Row of `table` (B,G,W) for the target matched to each query (B,Q), zeros if unmatched."""
    idx = tq.clamp(min=0)[..., None].expand(-1, -1, table.shape[-1])
    return torch.gather(table, 1, idx) * matched[..., None].to(table.dtype)


def count_boxes(ntargets):
    """Number of real target boxes of the (global) batch as a float tensor (losses.py:630-636)."""
    num_boxes = ntargets.sum().to(torch.float32).reshape(1)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(num_boxes)
    return num_boxes


def global_box_count(box_label_mask):
    """The loss's normaliser -- real target boxes of the GLOBAL batch (losses.py:630-636: sum over the scenes, all-reduced over
    the ranks) -- from a batch's `box_label_mask` alone, as a (1,) float32 tensor.  It depends on the targets only, so a data
    loader can form it when the batch arrives and hand it over as end_points["num_boxes_global"]: compute_hungarian_loss then runs
    NO collective, which is what lets the loss live inside a captured HIP graph at N > 1 (a collective captured in the step is
    queried by RCCL's watchdog thread while the stream captures: the process aborts, bench.py --force-dist --loss hungarian did
    in 2 of 3 runs).  Same number as the reference's in-loss all-reduce."""
    nb = (box_label_mask > 0).sum().to(torch.float32).reshape(1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(nb)
    return nb


class SetCriterion(nn.Module):
    """Position-aligned cross entropy, box L1 + GIoU and semantic-alignment contrastive losses
    (losses.py:339-647) on padded targets."""

    def __init__(self, matcher, losses={}, eos_coef=0.1, temperature=0.07):
        super().__init__()
        self.matcher, self.losses, self.eos_coef, self.temperature = matcher, losses, eos_coef, temperature

    # -- the three losses; tq (B,Q) = matched target slot or -1, assign (B,G), valid (B,G).
    #    Each returns PER-SCENE sums (B,) already divided by num_boxes: compute_hungarian_loss
    #    stacks all prediction heads into the batch dimension and splits the result per head. -----
    def loss_pos_align(self, outputs, tgt, tq, assign, valid, num_boxes):
        logp = outputs["pred_logits"].log_softmax(-1)
        C = logp.shape[-1]
        matched = tq >= 0
        if outputs["language_dataset"][0] == "sr3d":
            w = (0.625, 0.125, 0.125, 0.125)
        else:
            w = (0.6, 0.2, 0.2, 0.1)
        weight_pos = (tgt["positive_map"] * w[0] + tgt["modify_positive_map"] * w[1]
                      + tgt["pron_positive_map"] * w[2] + tgt["rel_positive_map"] * w[3])[..., :C]
        sim = _per_query(tq, matched, weight_pos)
        no_obj = (torch.arange(C, device=sim.device) == C - 1).to(sim.dtype)          # one-hot on "no object"
        sim = torch.where(matched[..., None], sim, no_obj)
        ce = (torch.log(sim + 1e-6) * sim - logp * sim).sum(-1)
        ce = ce * (matched.to(ce.dtype) * (1 - self.eos_coef) + self.eos_coef)
        return {"loss_ce": ce.sum(1) / num_boxes}

    def loss_boxes(self, outputs, tgt, tq, assign, valid, num_boxes):
        idx = assign.clamp(min=0).long()[..., None].expand(-1, -1, 6)
        src = torch.gather(outputs["pred_boxes"], 1, idx)
        ref_box = (torch.arange(6, device=src.device) >= 3).to(src.dtype)             # (0,0,0,1,1,1)
        src = torch.where(valid[..., None], src, ref_box)             # padded slots: harmless unit boxes
        tb = torch.where(valid[..., None], tgt["boxes"], ref_box)
        l1 = (src[..., :3] - tb[..., :3]).abs() + 0.2 * (src[..., 3:] - tb[..., 3:]).abs()
        giou = 1 - _giou3d(box_cxcyczwhd_to_xyzxyz(src), box_cxcyczwhd_to_xyzxyz(tb))
        v = valid.to(l1.dtype)
        return {"loss_bbox": (l1 * v[..., None]).sum((1, 2)) / num_boxes, "loss_giou": (giou * v).sum(1) / num_boxes}

    def loss_sem_align(self, outputs, tgt, tq, assign, valid, num_boxes):
        logits = torch.matmul(outputs["proj_queries"], outputs["proj_tokens"].transpose(-1, -2)) / self.temperature
        B, Q, L = logits.shape
        matched = tq >= 0
        am = outputs["tokenized"]["attention_mask"]
        last = (am.sum(1) - 1) % L                                   # "not mentioned" token (python-style wrap)
        prev = (am.sum(1) - 2) % L
        pos_tok = torch.arange(L, device=logits.device)[None, :]
        nm = ((pos_tok == last[:, None]) | (pos_tok == prev[:, None])).to(logits.dtype) * 0.5     # (B,L)
        pmap = torch.where(matched[..., None], _per_query(tq, matched, tgt["positive_map"][..., :L]),
                           nm[:, None, :].expand(-1, Q, -1)) > 0
        modi = _per_query(tq, matched, tgt["modify_positive_map"][..., :L])
        pron = _per_query(tq, matched, tgt["pron_positive_map"][..., :L])
        other = _per_query(tq, matched, tgt["other_entity_map"][..., :L])
        rel = _per_query(tq, matched, tgt["rel_positive_map"][..., :L])
        modi_b, pron_b, rel_b = modi > 0, pron > 0, rel > 0
        qmask = matched.to(logits.dtype) * (1 - self.eos_coef) + self.eos_coef
        neg_logits = -logits
        pos_l = neg_logits * pmap.to(logits.dtype)
        modi_l, pron_l, rel_l = (neg_logits * m.to(logits.dtype) for m in (modi_b, pron_b, rel_b))
        # object -> text
        neg = (logits + logits * (other > 0).to(logits.dtype)).logsumexp(2)
        b2t = (pos_l.sum(2) / (pmap.sum(2) + 1e-6) + 0.2 * modi_l.sum(2) / (modi_b.sum(2) + 1e-6)
               + 0.2 * pron_l.sum(2) / (pron_b.sum(2) + 1e-6) + 0.1 * rel_l.sum(2) / (rel_b.sum(2) + 1e-6) + neg)
        b2t = (b2t * pmap.any(2).to(b2t.dtype) * qmask).sum(1)
        # text -> object; the token weights are overwritten in this order (losses.py:550-556)
        tmask = torch.full((B, L), self.eos_coef, dtype=logits.dtype, device=logits.device)
        tmask = torch.where(pos_tok == last[:, None], torch.full_like(tmask, 1.0), tmask)
        tmask = torch.where(pmap.any(1), torch.full_like(tmask, 1.0), tmask)
        tmask = torch.where(modi_b.any(1), torch.full_like(tmask, 0.2), tmask)
        tmask = torch.where(pron_b.any(1), torch.full_like(tmask, 0.2), tmask)
        tmask = torch.where(rel_b.any(1), torch.full_like(tmask, 0.1), tmask)
        tmask = torch.where(pos_tok == prev[:, None], torch.full_like(tmask, 0.1), tmask)
        with_pos = (pmap | modi_b | pron_b | rel_b).any(1)
        pos_t = pos_l.sum(1) + modi_l.sum(1) + pron_l.sum(1) + rel_l.sum(1)
        nb = pmap.sum(1) + modi.sum(1) + pron.sum(1) + rel.sum(1) + 1e-6     # counts + map VALUES, as the reference
        t2b = -torch.log(nb + 1e-6) / nb + pos_t / nb + logits.logsumexp(1)
        t2b = (t2b * with_pos.to(t2b.dtype) * tmask).sum(1)
        return {"loss_sem_align": (b2t + t2b) / 2 / num_boxes}

    _LOSSES = {"boxes": "loss_boxes", "labels": "loss_pos_align", "contrastive_align": "loss_sem_align"}

    def forward_padded(self, outputs, tgt, ntargets, valid, assign=None, num_boxes=None, per_scene=False):
        """tgt: compacted padded targets {"boxes" (B,G,6), "labels" (B,G), five "*_map" (B,G,256)};
        ntargets (B,) int32, valid (B,G) bool.  assign (B,G): given, or solved on the device.
        num_boxes: normaliser (default: sum of ntargets, all-reduced over ranks like the reference);
        per_scene: return (B,) vectors instead of their sums."""
        if assign is None:
            assign = self.matcher.match_padded(outputs["pred_logits"].detach(), outputs["pred_boxes"].detach(),
                                               tgt["boxes"], ntargets, tgt["positive_map"], tgt["labels"])
        B, Q = outputs["pred_logits"].shape[:2]
        G = assign.shape[1]
        # target slot matched to each query (or -1): scatter the slot index to its query
        slot = torch.arange(G, device=assign.device)[None, :].expand(B, -1)
        a = torch.where(valid, assign.long(), torch.full_like(assign.long(), Q))      # padded -> dummy column Q
        tq = torch.full((B, Q + 1), -1, dtype=torch.long, device=assign.device).scatter_(1, a, slot)[:, :Q]
        if num_boxes is None:
            num_boxes = count_boxes(ntargets)
        losses = {}
        for name in self.losses:
            assert name in self._LOSSES, f"do you really want to compute {name} loss?"
            losses.update(getattr(self, self._LOSSES[name])(outputs, tgt, tq, assign, valid, num_boxes))
        if not per_scene:
            losses = {k: v.sum() for k, v in losses.items()}
        return losses, assign

    def forward(self, outputs, targets):
        """Reference signature (list of per-scene dicts); returns (losses, assign (B,G))."""
        B = outputs["pred_boxes"].shape[0]
        dev = outputs["pred_boxes"].device
        sizes = [len(t["boxes"]) for t in targets]
        G = max(max(sizes), 1)
        keys = ["positive_map", "modify_positive_map", "pron_positive_map", "other_entity_map", "rel_positive_map"]
        tgt = {"boxes": torch.zeros((B, G, 6), device=dev), "labels": torch.zeros((B, G), dtype=torch.long, device=dev)}
        for k in keys:
            tgt[k] = torch.zeros((B, G, targets[0][k].shape[-1]), device=dev)
        for b, t in enumerate(targets):
            n = sizes[b]
            tgt["boxes"][b, :n] = t["boxes"]
            tgt["labels"][b, :n] = t["labels"]
            for k in keys:
                tgt[k][b, :n] = t[k]
        nt = torch.tensor(sizes, dtype=torch.int32, device=dev)
        valid = torch.arange(G, device=dev)[None, :] < nt[:, None]
        return self.forward_padded(outputs, tgt, nt, valid)


def compute_hungarian_loss(end_points, num_decoder_layers, set_criterion, query_points_obj_topk=5, assign=None):
    """Total loss over the proposal head and the decoder heads (losses.py:650-738); writes the
    same keys into end_points.  The reference runs matcher + criterion once per head; here the P
    heads are stacked into the batch dimension -- one (P*B, Q, G) cost, ONE assignment launch,
    one pass of each loss -- and the per-head values are read off the per-scene sums.
    `assign`: optional {prefix: (B,G) assignment} (tests on CPU).  end_points["num_boxes_global"] (optional, (1,) float:
    global_box_count of this batch) replaces the in-loss all-reduce of the box count.  On CUDA tensors the same arithmetic runs as a handful
    of fused launches (losses_fused.py / csrc/loss.hip) unless EDA_FUSED_LOSS=0."""
    from . import losses_fused
    if losses_fused.usable(end_points, set_criterion, assign):
        return losses_fused.compute_hungarian_loss(end_points, num_decoder_layers, set_criterion, query_points_obj_topk)
    prefixes = ["proposal_", "last_"] + [f"{i}head_" for i in range(num_decoder_layers - 1)]
    P = len(prefixes)
    gt_box = torch.cat([end_points["center_label"][:, :, 0:3], end_points["size_gts"]], dim=-1)
    keys = ["positive_map", "modify_positive_map", "pron_positive_map", "other_entity_map", "rel_positive_map"]
    nt, valid, packed = compact_targets(end_points["box_label_mask"], gt_box, end_points["sem_cls_label"],
                                        *[end_points[k] for k in keys])
    B = nt.shape[0]
    rep = lambda t: t.unsqueeze(0).expand(P, *t.shape).reshape(P * t.shape[0], *t.shape[1:])    # heads x scenes
    tgt = {"boxes": rep(packed[0]), "labels": rep(packed[1])}
    tgt.update({k: rep(packed[2 + i]) for i, k in enumerate(keys)})
    stack = lambda name: torch.cat([end_points[f"{p}{name}"] for p in prefixes], dim=0)
    out = {"pred_logits": stack("sem_cls_scores"),
           "pred_boxes": torch.cat([stack("center"), stack("pred_size")], dim=-1),
           "language_dataset": end_points["language_dataset"]}
    if "proj_tokens" in end_points:
        out["proj_tokens"] = rep(end_points["proj_tokens"])
        out["proj_queries"] = stack("proj_queries")
        out["tokenized"] = {"attention_mask": rep(end_points["tokenized"]["attention_mask"])}
    a_in = None if assign is None else torch.cat([assign[p] for p in prefixes], dim=0)
    nb = end_points.get("num_boxes_global")          # (the caller's: global_box_count() outside the step) or counted here
    losses, a = set_criterion.forward_padded(out, tgt, rep(nt), rep(valid), a_in,
                                             num_boxes=count_boxes(nt) if nb is None else nb.to(torch.float32).reshape(1),
                                             per_scene=True)
    tot = {"loss_ce": 0, "loss_bbox": 0, "loss_giou": 0, "loss_sem_align": 0}
    for k, v in losses.items():
        per_head = v.reshape(P, B).sum(1)
        tot[k] = per_head.sum()
        for i, prefix in enumerate(prefixes):
            end_points[f"{prefix}_{k}"] = per_head[i]
    for i, prefix in enumerate(prefixes):
        end_points[f"{prefix}assign"] = a[i * B:(i + 1) * B]
    qp = (compute_points_obj_cls_loss_hard_topk(end_points, query_points_obj_topk)
          if "seeds_obj_cls_logits" in end_points else 0.0)
    weight = 0.5 if end_points["language_dataset"][0] == "scanrefer" else 1
    loss = 8 * qp + 1.0 / (num_decoder_layers + 1) * (
        weight * tot["loss_ce"] + 5 * tot["loss_bbox"] + tot["loss_giou"] + weight * tot["loss_sem_align"])
    end_points.update(loss_ce=tot["loss_ce"], loss_bbox=tot["loss_bbox"], loss_giou=tot["loss_giou"],
                      query_points_generation_loss=qp, loss_sem_align=tot["loss_sem_align"], loss=loss)
    return loss, end_points
