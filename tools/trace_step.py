"""Export a chrome trace of ONE eager training step (gzip) for offline analysis of which
aten / autograd op launched each small kernel:  python tools/trace_step.py out.json.gz"""
import gzip
import os
import shutil
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from eda_amd.bdetr import BeaUTyDETR  # noqa: E402
from eda_amd.parallel import FlatParams  # noqa: E402


def main():
    out = sys.argv[1]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = BeaUTyDETR().to(dev).train()
    model.text_encoder.eval()
    grads = FlatParams(model)
    opt = torch.optim.AdamW(list(grads.groups.values()), lr=1e-4, weight_decay=5e-4, fused=True)
    inputs = bench.make_inputs(0, 8, dev, 50000, 80)
    loss_fn = bench.synthetic_loss
    if os.environ.get("LOSS") == "hungarian":
        from eda_amd import losses as L
        targets = bench.make_targets(0, 8, dev, inputs)
        crit = L.SetCriterion(L.HungarianMatcher(1, 0, 2, True), losses=["boxes", "labels", "contrastive_align"])

        def loss_fn(ep):
            ep.update(targets)
            ep["language_dataset"] = ["scanrefer"] * 8
            return L.compute_hungarian_loss(ep, 6, crit, query_points_obj_topk=4)[0]

    def step():
        loss = loss_fn(model(inputs))
        with grads.deferred_wgrad():
            loss.backward()
        grads.collect_grads()
        grads.clip_grad_norm_(0.1)
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    tmp = out + ".tmp.json"
    prof.export_chrome_trace(tmp)
    with open(tmp, "rb") as f, gzip.open(out, "wb") as g:
        shutil.copyfileobj(f, g)
    os.remove(tmp)


if __name__ == "__main__":
    main()
