"""Which python lines / autograd shapes issue the small torch ops of one training step.
A TorchDispatchMode logs every aten call of one step: forward calls are keyed by the nearest
frame inside this repo, backward calls (issued by the autograd engine) by op + argument shapes.
    OPS=copy_,add,add_,sum,div,mul,fill_,zeros python tools/op_census.py
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from eda_amd.bdetr import BeaUTyDETR  # noqa: E402
from eda_amd.parallel import FlatParams  # noqa: E402


class Census(TorchDispatchMode):
    def __init__(self, want):
        super().__init__()
        self.want, self.phase = want, "fwd"
        self.agg = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in self.want or ("*" in self.want and name not in ("detach", "view", "_unsafe_view", "t", "transpose", "expand", "slice", "select", "unsqueeze", "squeeze", "permute", "alias", "as_strided", "unbind", "split", "split_with_sizes", "reshape", "empty", "empty_like", "empty_strided", "new_empty", "_reshape_alias", "is_same_size", "lift_fresh", "unfold", "size", "stride")):
            shapes = ",".join(str(tuple(a.shape)) + ("" if a.is_contiguous() else "nc")
                              for a in args if isinstance(a, torch.Tensor))
            where = ""
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "/eda_amd/" in fr.filename or fr.filename.endswith("bench.py"):
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            self.agg[(self.phase, name, where, shapes)] += 1
        return func(*args, **(kwargs or {}))


def main():
    want = set(os.environ.get("OPS", "copy_,add,add_,sum,div,mul,fill_,zeros,zeros_like,mean,cat,clone").split(","))
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = BeaUTyDETR().to(dev).train()
    model.text_encoder.eval()
    grads = FlatParams(model)
    inputs = bench.make_inputs(0, 8, dev, 50000, 80)
    for _ in range(2):
        with grads.deferred_wgrad():
            bench.synthetic_loss(model(inputs)).backward()
        grads.collect_grads()
    c = Census(want)
    with c:
        loss = bench.synthetic_loss(model(inputs))
        c.phase = "bwd"
        with grads.deferred_wgrad():                    # (the step bench.py runs: weight gradients queued)
            loss.backward()
        c.phase = "opt"
        grads.collect_grads()
        grads.clip_grad_norm_(0.1)
    tot = collections.Counter()
    for (ph, name, where, shapes), n in c.agg.items():
        tot[name] += n
    print("totals:", dict(tot))
    for (ph, name, where, shapes), n in sorted(c.agg.items(), key=lambda kv: (kv[0][1], -kv[1])):
        if n >= int(os.environ.get("MIN", 2)):
            print(f"{n:4d} {ph} {name:10s} {where:32s} {shapes[:110]}")


if __name__ == "__main__":
    main()
