"""Device-kernel launches of the training loss (eda_amd/losses.py) by part, forward + backward, at the bench shapes
(7 heads x 8 scenes, 256 queries, 132 target slots, 256 token classes, 80 tokens).  usage: python tools/loss_census.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from eda_amd import losses as L  # noqa: E402


def count(fn):
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    if "--list" in sys.argv:
        for e in ev:
            print("    %7.1f us  %s" % (e.device_time, e.name[:150]))
    return len(ev), sum(e.device_time for e in ev)


def main():
    dev = torch.device("cuda", 0)
    B, Q, G, C, Lt, P = 8, 256, 132, 256, 80, 7
    inputs = bench.make_inputs(0, B, dev, 50000, Lt)
    tg = bench.make_targets(0, B, dev, inputs)
    torch.manual_seed(0)
    prefixes = ["proposal_", "last_"] + [f"{i}head_" for i in range(5)]
    ep = dict(tg)
    ep["language_dataset"] = ["scanrefer"] * B
    ep["tokenized"] = inputs["tokenized"]
    ep["proj_tokens"] = torch.nn.functional.normalize(torch.randn(B, Lt, 64, device=dev), dim=-1).requires_grad_(True)
    ep["seed_inds"] = torch.randint(0, 50000, (B, 1024), device=dev, dtype=torch.int32)
    ep["seed_xyz"] = torch.gather(inputs["point_clouds"][..., :3], 1, ep["seed_inds"].long()[..., None].expand(-1, -1, 3))
    ep["seeds_obj_cls_logits"] = torch.randn(B, 1, 1024, device=dev, requires_grad=True)
    for p in prefixes:
        ep[p + "center"] = torch.randn(B, Q, 3, device=dev, requires_grad=True)
        ep[p + "pred_size"] = (torch.rand(B, Q, 3, device=dev) + 0.2).requires_grad_(True)
        ep[p + "sem_cls_scores"] = torch.randn(B, Q, C, device=dev, requires_grad=True)
        ep[p + "proj_queries"] = torch.nn.functional.normalize(torch.randn(B, Q, 64, device=dev), dim=-1).requires_grad_(True)
    crit = L.SetCriterion(L.HungarianMatcher(1, 0, 2, True), losses=["boxes", "labels", "contrastive_align"], eos_coef=0.1,
                          temperature=0.07)

    def whole():
        e = dict(ep)
        loss = L.compute_hungarian_loss(e, 6, crit, query_points_obj_topk=4)[0]
        loss.backward()
    n, t = count(whole)
    print("whole loss forward + backward: %d launches, %.1f us of kernels" % (n, t))
    if "--list" in sys.argv:
        return
    for parts in (["boxes"], ["labels"], ["contrastive_align"], []):
        c2 = L.SetCriterion(L.HungarianMatcher(1, 0, 2, True), losses=parts, eos_coef=0.1, temperature=0.07)

        def part():
            e = dict(ep)
            if not parts:
                e.pop("seeds_obj_cls_logits")
            loss = L.compute_hungarian_loss(e, 6, c2, query_points_obj_topk=4)[0]
            if torch.is_tensor(loss) and loss.requires_grad:
                loss.backward()
        n, t = count(part)
        print("%-22s (+ matching, stacking%s): %d launches, %.1f us" % (parts or "none", ", objectness" if parts else "", n, t))


if __name__ == "__main__":
    main()
