"""Evaluation post-processing of EDA: language-grounding accuracies.

Same public surface and the same numbers as the reference's `src/grounding_evaluator.py:30-394`
(`GroundingEvaluator(only_root, thresholds, topks, prefixes, filter_non_gt_boxes)`, `.evaluate(end_points, prefix)`,
`.evaluate_bbox_by_pos_align`, `.evaluate_bbox_by_sem_align`, `.dets` / `.gts` counters with the reference's keys,
`.reset()`, `.print_stats()`, `.synchronize_between_processes()`), organised for the device: the reference walks the
batch sample by sample on the host (token-score products, `argsort`, IoU, `.item()` per threshold and k); here a batch is
one set of batched tensor operations and ONE device-to-host copy of all counters per `evaluate_*` call.

    position alignment  scores = softmax(sem_cls_scores) . token maps      (grounding_evaluator.py:133-224)
    semantic alignment  scores = softmax(proj_queries proj_tokens^T / 0.07) . token maps   (:226-372)
    score of a query for ground-truth object o: main(o) + modifier + pronoun + relation - other-entity maps (the four
    auxiliary maps are the FIRST object's, as in the reference), top-10 queries by score, 3D IoU of their boxes with
    the object's box, Acc@t for k in topks; on the `last_` prefix the semantic branch also fills the
    view-dependent / hard / unique break-downs from the first object.

Parity: tests/test_evaluator.py compares every counter with goldens produced by RUNNING the reference's evaluator in
the build container (tools/gen_golden_eval.py).
"""
import torch
import torch.distributed as dist

from .losses import box_cxcyczwhd_to_xyzxyz


def _iou3d_pairs(a, b):
    """IoU of corner boxes a (..., 6) with b (..., 6), broadcast (losses.py:46-74)."""
    lo = torch.maximum(a[..., :3], b[..., :3])
    hi = torch.minimum(a[..., 3:], b[..., 3:])
    e = (hi - lo).clamp(min=0)
    inter = e[..., 0] * e[..., 1] * e[..., 2]
    va = (a[..., 3] - a[..., 0]) * (a[..., 4] - a[..., 1]) * (a[..., 5] - a[..., 2])
    vb = (b[..., 3] - b[..., 0]) * (b[..., 4] - b[..., 1]) * (b[..., 5] - b[..., 2])
    return inter / (va + vb - inter)


class GroundingEvaluator:
    """Evaluate language grounding (reference: src/grounding_evaluator.py:30-49).

    only_root: detect only the root noun; thresholds: IoU thresholds; topks: k of top-k accuracy; prefixes: names of
    the prediction heads to evaluate; filter_non_gt_boxes: zero the score of queries that overlap no detected box."""

    ANALYSIS = ("vd", "vid", "hard", "easy", "multi", "unique")

    def __init__(self, only_root=True, thresholds=(0.25, 0.5), topks=(1, 5, 10), prefixes=(), filter_non_gt_boxes=False):
        self.only_root = only_root
        self.thresholds = list(thresholds)
        self.topks = list(topks)
        self.prefixes = list(prefixes)
        self.filter_non_gt_boxes = filter_non_gt_boxes
        self.reset()

    def reset(self):
        """Reset accumulators (same keys and initial values as the reference, :51-76)."""
        self.dets = {(p, t, k, m): 0 for p in self.prefixes for t in self.thresholds for k in self.topks
                     for m in ("bbs", "bbf")}
        self.gts = dict(self.dets)
        for suffix in ("", "50"):
            for f in self.ANALYSIS:
                self.dets[f + suffix] = 0
                self.gts[f + suffix] = 1e-14

    def print_stats(self):
        mode_str = {"bbs": "position alignment", "bbf": "semantic alignment"}
        for prefix in self.prefixes:
            for mode in ("bbs", "bbf"):
                for t in self.thresholds:
                    print(prefix, mode_str[mode], "Acc%.2f:" % t, ", ".join(
                        "Top-%d: %.5f" % (k, self.dets[(prefix, t, k, mode)] / max(self.gts[(prefix, t, k, mode)], 1))
                        for k in self.topks))
        print("\nAnalysis")
        for title, suffix in (("iou@0.25", ""), ("iou@0.50", "50")):
            print(title)
            for f in ("easy", "hard", "vd", "vid", "unique", "multi"):
                print(f + suffix, self.dets[f + suffix] / self.gts[f + suffix])

    def synchronize_between_processes(self):
        """Sum the counters over the ranks (the reference gathers pickled dicts and merges on rank 0, :106-124; here one
        all-reduce of a flat tensor, every rank ends with the totals)."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        keys = sorted(self.dets, key=str)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([[float(self.dets[k]), float(self.gts[k])] for k in keys], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        for k, (d, g) in zip(keys, t.tolist()):
            self.dets[k] = int(round(d)) if float(d).is_integer() else d
            self.gts[k] = g if k in self._analysis_keys() else int(round(g))

    def _analysis_keys(self):
        return {f + s for f in self.ANALYSIS for s in ("", "50")}

    # ------------------------------------------------------------------------------------------------ evaluation
    def evaluate(self, end_points, prefix):
        """Both alignments (:126-137)."""
        self.evaluate_bbox_by_pos_align(end_points, prefix)
        self.evaluate_bbox_by_sem_align(end_points, prefix)

    def evaluate_bbox_by_pos_align(self, end_points, prefix):
        """Score = softmax of the per-token classification scores (:139-224)."""
        sem = end_points[f"{prefix}sem_cls_scores"].softmax(-1)
        T = end_points["positive_map"].shape[-1]
        if sem.shape[-1] != T:
            pad = sem.new_zeros(sem.shape[0], sem.shape[1], T)
            pad[:, :, :sem.shape[-1]] = sem
            sem = pad
        self._accumulate(end_points, prefix, sem, "bbs")

    def evaluate_bbox_by_sem_align(self, end_points, prefix):
        """Score = softmax of the query / token similarity at temperature 0.07, padded to 256 token slots (:226-372)."""
        sim = torch.matmul(end_points[f"{prefix}proj_queries"], end_points["proj_tokens"].transpose(-1, -2))
        sm = (sim / 0.07).softmax(-1)
        sem = sm.new_zeros(sm.shape[0], sm.shape[1], 256)
        sem[:, :, :sm.shape[2]] = sm
        self._accumulate(end_points, prefix, sem, "bbf")

    def _accumulate(self, ep, prefix, sem, mode):
        B, Q, T = sem.shape
        dev = sem.device
        pmap = (ep["positive_map"] > 0).to(sem.dtype)                     # (B, G, T): 1 on the object's tokens (:376-384)
        gt = torch.cat([ep["center_label"][:, :, 0:3], ep["size_gts"]], dim=-1)
        if self.only_root:
            pmap, gt = pmap[:, :1], gt[:, :1]
        G = pmap.shape[1]
        nobj = ep["box_label_mask"].sum(1).long().clamp(max=G)            # annotated objects per scene
        valid = torch.arange(G, device=dev)[None, :] < nobj[:, None]       # (B, G)
        pred_size = ep[f"{prefix}pred_size"]
        assert (pred_size < 0).sum() == 0
        pred = torch.cat([ep[f"{prefix}center"], pred_size], dim=-1)       # (B, Q, 6)
        # auxiliary components: the first object's maps, for every object of the scene
        extra = (ep["modify_positive_map"][:, 0] + ep["pron_positive_map"][:, 0] + ep["rel_positive_map"][:, 0]
                 - ep["other_entity_map"][:, 0]).to(sem.dtype)            # (B, T)
        scores = torch.einsum("bqt,bot->boq", sem, pmap) + torch.einsum("bqt,bt->bq", sem, extra)[:, None, :]
        pred_c = box_cxcyczwhd_to_xyzxyz(pred)
        if self.filter_non_gt_boxes:
            det = box_cxcyczwhd_to_xyzxyz(ep["all_detected_boxes"])       # (B, D, 6)
            dmask = ep["all_detected_bbox_label_mask"].bool()
            iou_d = _iou3d_pairs(det[:, :, None, :], pred_c[:, None, :, :])            # (B, D, Q)
            iou_d = torch.where(dmask[:, :, None], iou_d, iou_d.new_full((), -1.0))
            scores = scores * (iou_d.max(1)[0] > 0.25).to(scores.dtype)[:, None, :]
        top = scores.argsort(-1, descending=True)[..., :10]                # (B, G, 10)
        pbox = torch.gather(pred_c[:, None].expand(B, G, Q, 6), 2, top[..., None].expand(B, G, top.shape[-1], 6))
        ious = _iou3d_pairs(box_cxcyczwhd_to_xyzxyz(gt)[:, :, None, :], pbox)          # (B, G, 10)
        rows = []
        for t in self.thresholds:
            hit = ious > t
            for k in self.topks:
                found = hit[..., :k].any(-1) & valid                       # (B, G)
                rows.append(found.sum())
        n_valid = valid.sum()
        analysis = None
        if mode == "bbf" and prefix == "last_":
            # break-downs of the FIRST object at top-1 (:330-372); a scene without annotated object would fail in the reference
            f25 = (ious[:, 0, :1] > self.thresholds[0]).any(-1)
            f50 = (ious[:, 0, :1] > self.thresholds[1]).any(-1) if len(self.thresholds) > 1 else None
            flags = [torch.as_tensor(ep[name], device=dev).bool().reshape(B) for name in ("is_view_dep", "is_hard", "is_unique")]
            analysis = torch.stack([f25.long()] + ([f50.long()] if f50 is not None else []) + [f.long() for f in flags])
        packed = torch.stack(rows + [n_valid]).cpu().tolist()              # the one device-to-host copy of this call
        i = 0
        for t in self.thresholds:
            for k in self.topks:
                self.dets[(prefix, t, k, mode)] += int(packed[i])
                self.gts[(prefix, t, k, mode)] += int(packed[-1])
                i += 1
        if analysis is not None:
            a = analysis.cpu().tolist()
            has50 = len(self.thresholds) > 1
            f25, f50 = a[0], (a[1] if has50 else None)
            vd, hard, uniq = a[-3], a[-2], a[-1]
            for b in range(B):
                for suffix, found in (("", f25[b]),) + ((("50", f50[b]),) if has50 else ()):
                    for on, name_on, name_off in ((vd[b], "vd", "vid"), (hard[b], "hard", "easy"), (uniq[b], "unique", "multi")):
                        key = (name_on if on else name_off) + suffix
                        self.gts[key] += 1
                        self.dets[key] += found
