"""Seeded inputs of the evaluation post-processing parity test (shared by tools/gen_golden_eval.py, which feeds them to
the REFERENCE's GroundingEvaluator in the build container, and tests/test_evaluator.py)."""
import numpy as np
import torch

import loss_fixtures as LF

PREFIXES = list(LF.PREFIXES)
# case -> (seed, only_root, filter_non_gt_boxes)
CASES = {"root": (21, True, False), "all_objects": (22, False, False), "root_filtered": (23, True, True)}


def make_end_points(seed):
    """loss_fixtures' end_points (predictions for three heads, padded targets with the five token maps) made to look like
    a half-trained model -- some queries sit near the annotated boxes, their token scores lean towards the object's
    tokens -- plus what only the evaluator reads: the analysis flags and the detected boxes of the filter option."""
    ep = LF.make_end_points(seed, B=8, Q=48, L=24)
    rng = np.random.default_rng(1000 + seed)
    B, Q = ep["last_center"].shape[:2]
    nobj = ep["box_label_mask"].sum(1).long()
    for p in PREFIXES:
        for b in range(B):
            for o in range(int(nobj[b])):
                for q in rng.choice(Q, 4, replace=False):
                    # (a mix of hits and misses at both IoU thresholds and at every k)
                    jitter = torch.from_numpy(rng.normal(0, 0.22, 3).astype(np.float32))
                    ep[f"{p}center"][b, q] = ep["center_label"][b, o] + jitter
                    ep[f"{p}pred_size"][b, q] = ep["size_gts"][b, o] * float(rng.uniform(0.7, 1.4))
                    tok = (ep["positive_map"][b, o] > 0).float()
                    ep[f"{p}sem_cls_scores"][b, q] += float(rng.uniform(0.5, 2.5)) * tok
                    t64 = (tok[:ep["proj_tokens"].shape[1], None] * ep["proj_tokens"][b]).sum(0)
                    v = ep[f"{p}proj_queries"][b, q] + float(rng.uniform(0.1, 0.6)) * t64
                    ep[f"{p}proj_queries"][b, q] = v / v.norm()
    ep["is_view_dep"] = torch.from_numpy(rng.integers(0, 2, B).astype(bool))
    ep["is_hard"] = torch.from_numpy(rng.integers(0, 2, B).astype(bool))
    ep["is_unique"] = torch.from_numpy(rng.integers(0, 2, B).astype(bool))
    D = 16
    det = torch.cat([torch.from_numpy(rng.uniform(-2, 2, (B, D, 3)).astype(np.float32)),
                     torch.from_numpy(rng.uniform(0.3, 1.5, (B, D, 3)).astype(np.float32))], -1)
    for b in range(B):                                   # the annotated objects are among the detections
        for o in range(int(nobj[b])):
            det[b, o, :3] = ep["center_label"][b, o]
            det[b, o, 3:] = ep["size_gts"][b, o]
    ep["all_detected_boxes"] = det
    m = torch.zeros(B, D, dtype=torch.bool)
    for b in range(B):
        m[b, :int(rng.integers(4, D + 1))] = True
    ep["all_detected_bbox_label_mask"] = m
    return ep


def counter_keys(ev):
    return sorted(ev.dets, key=str)
