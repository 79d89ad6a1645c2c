"""Where does a training step spend its device time?  Segment timing (HIP events)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from eda_amd.bdetr import BeaUTyDETR  # noqa: E402
from eda_amd.parallel import FlatParams  # noqa: E402

marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


def hook_module(mod, name):
    mod.register_forward_pre_hook(lambda m, i: mark(f"> {name}"))
    mod.register_forward_hook(lambda m, i, o: mark(f"< {name}"))


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = BeaUTyDETR().to(dev).train()
    model.text_encoder.eval()
    grads = FlatParams(model)
    opt = torch.optim.AdamW(list(grads.groups.values()), lr=1e-4, weight_decay=5e-4, fused=True)
    inputs = bench.make_inputs(0, 8, dev, 50000, 80)
    bb = model.backbone_net
    for n in ["sa1", "sa2", "sa3", "sa4", "fp1", "fp2"]:
        hook_module(getattr(bb, n), n)
    hook_module(model.text_encoder, "roberta")
    hook_module(model.cross_encoder, "cross_encoder")
    for i, d in enumerate(model.decoder):
        hook_module(d, f"decoder{i}")
    for i, d in enumerate(model.prediction_heads):
        hook_module(d, f"head{i}")

    def step():
        marks.clear()
        mark("start")
        ep = model(inputs)
        mark("fwd_end")
        loss = bench.synthetic_loss(ep)
        mark("loss_end")
        loss.backward()
        mark("bwd_end")
        grads.collect_grads()
        grads.clip_grad_norm_(0.1)
        opt.step()
        mark("opt_end")

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    step()
    torch.cuda.synchronize()
    t0 = marks[0][1]
    prev = t0
    agg = {}
    open_t = {}
    for name, e in marks[1:]:
        if name.startswith("> "):
            open_t[name[2:]] = e
        elif name.startswith("< "):
            k = name[2:]
            kk = "decoder(6)" if k.startswith("decoder") else "heads(6)" if k.startswith("head") else k
            agg[kk] = agg.get(kk, 0) + open_t[k].elapsed_time(e)
        else:
            print(f"{name:10s} at {t0.elapsed_time(e):8.2f} ms")
    for k, v in agg.items():
        print(f"  fwd {k:14s} {v:7.2f} ms")


if __name__ == "__main__":
    main()
