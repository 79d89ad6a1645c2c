"""tests/golden/loss_*.npz: outputs of the REFERENCE's models/losses.py (imported from
/root/reference in this build container only; it needs just scipy + torch) on the seeded inputs
of tests/loss_fixtures.py: the total loss, its parts per prefix, the matcher's indices and the
gradients w.r.t. every prediction tensor.  Only arrays are stored.
    python tools/gen_golden_loss.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import loss_fixtures as LF  # noqa: E402


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_losses", "/root/reference/models/losses.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    for name, seed, dataset in (("scanrefer", 11, "scanrefer"), ("sr3d", 12, "sr3d")):
        ep = LF.make_end_points(seed, dataset=dataset)
        for k in LF.GRAD_KEYS:
            ep[k].requires_grad_(True)
        matcher = ref.HungarianMatcher(1, 0, 2, True)                  # train_dist_mod.py / main_utils defaults
        crit = ref.SetCriterion(matcher, losses=["boxes", "labels", "contrastive_align"], eos_coef=0.1,
                                temperature=0.07)
        out = {}
        # matcher indices per prefix (recorded before the loss call mutates nothing)
        ep2 = dict(ep)
        ref.compute_hungarian_loss(ep2, 2, crit, query_points_obj_topk=5)
        loss = ep2["loss"]
        loss.backward()
        out["loss"] = loss.detach().numpy()
        for k in ("loss_ce", "loss_bbox", "loss_giou", "loss_sem_align", "query_points_generation_loss"):
            out[k] = torch.as_tensor(ep2[k]).detach().numpy()
        for p in LF.PREFIXES:
            for k in ("loss_ce", "loss_bbox", "loss_giou", "loss_sem_align"):
                out[f"{p}_{k}"] = ep2[f"{p}_{k}"].detach().numpy()
        for k in LF.GRAD_KEYS:
            out["grad_" + k] = ep[k].grad.numpy()
        # the matcher alone, on the 'last_' prediction: scene-wise (query index, target index) pairs
        B = ep["box_label_mask"].shape[0]
        tgt = [{"labels": ep["sem_cls_label"][b, ep["box_label_mask"][b].bool()],
                "boxes": torch.cat([ep["center_label"], ep["size_gts"]], -1)[b, ep["box_label_mask"][b].bool()],
                "positive_map": ep["positive_map"][b, ep["box_label_mask"][b].bool()]} for b in range(B)]
        o = {"pred_logits": ep["last_sem_cls_scores"].detach(),
             "pred_boxes": torch.cat([ep["last_center"], ep["last_pred_size"]], -1).detach()}
        ind = matcher(o, tgt)
        pairs = np.full((B, LF.G, 2), -1, np.int64)
        for b, (i, j) in enumerate(ind):
            pairs[b, :len(i), 0] = i.numpy(); pairs[b, :len(i), 1] = j.numpy()
        out["last_match_pairs"] = pairs
        path = os.path.join(ROOT, "tests", "golden", f"loss_{name}.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path) // 1024, "KiB; loss", float(np.asarray(out["loss"]).reshape(-1)[0]))


if __name__ == "__main__":
    main()
