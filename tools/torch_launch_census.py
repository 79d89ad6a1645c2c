"""Where the training step's NON-library launches come from: one eager forward + synthetic loss + backward of the bench model under
torch.profiler with Python stacks; every device kernel that is not libeda_hip.so's is attributed to (forward: the innermost eda_amd /
bench source line; backward: the autograd node whose evaluation launched it) and counted.
usage: python tools/torch_launch_census.py [--loss hungarian] [--top 60]"""
import collections
import os
import re
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from step_sequence import NATIVE, short  # noqa: E402


def main():
    from eda_amd.bdetr import BeaUTyDETR
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    from eda_amd.parallel import FlatParams, reference_lr_groups
    model = BeaUTyDETR(num_queries=256, butd=True).to(dev).train()
    model.text_encoder.eval()                  # frozen (bdetr.py:78-80), as bench.py runs it
    flat = FlatParams(model, reference_lr_groups)
    inputs = bench.make_inputs(0, 8, dev, 50000, 80)
    hung = "--loss" in sys.argv and sys.argv[sys.argv.index("--loss") + 1] == "hungarian"
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 60
    if hung:
        from eda_amd import losses as L
        crit = L.SetCriterion(L.HungarianMatcher(1, 0, 2, True), losses=["boxes", "labels", "contrastive_align"], eos_coef=0.1,
                              temperature=0.07)
        tg = bench.make_targets(0, 8, dev, inputs)

    def step():
        ep = model(inputs)
        if hung:
            ep.update(tg)
            loss = L.compute_hungarian_loss(ep, 6, crit, query_points_obj_topk=4)[0]
        else:
            loss = bench.synthetic_loss(ep)
        with flat.deferred_wgrad():            # weight gradients: one grouped kernel after the backward (bench.py)
            loss.backward()
        flat.collect_grads()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    if "--sources" in sys.argv:
        # which source lines of eda_amd / bench.py call the torch functions that launch (forward and backward python code)
        import traceback
        from torch.overrides import TorchFunctionMode, resolve_name
        names = ("add", "cat", "contiguous", "stack", "mul", "sum", "to", "clone", "copy_", "index", "gather", "full", "fill_", "div",
                 "sub", "float", "long", "reshape", "repeat", "expand_as", "where", "sqrt", "reciprocal", "ne", "eq", "topk", "matmul",
                 "bmm", "dot", "pow", "mean", "zeros", "ones", "full_like", "zeros_like", "ones_like", "arange", "cumsum", "sigmoid")
        calls = collections.Counter()

        class Mode(TorchFunctionMode):
            def __torch_function__(self, func, types, args=(), kwargs=None):
                n = (resolve_name(func) or getattr(func, "__name__", "?")).split(".")[-1].strip("_")
                n = {"radd": "add", "iadd": "add", "rmul": "mul", "imul": "mul", "truediv": "div", "getitem": "index"}.get(n, n)
                if n in names:
                    for fr in reversed(traceback.extract_stack()[:-1]):
                        if "/eda_amd/" in fr.filename or fr.filename.endswith("bench.py"):
                            calls[(n, re.sub(r".*/(eda_amd/|bench\.py)", r"\1", fr.filename) + ":%d" % fr.lineno, fr.line[:90])] += 1
                            break
                return func(*args, **(kwargs or {}))

        with Mode():
            step()
        torch.cuda.synchronize()
        for (n, where, line), c in sorted(calls.items(), key=lambda kv: (-kv[1], kv[0])):
            print("%3d x %-10s %-44s %s" % (c, n, where, line))
        return
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    ev = prof.events()
    cpu = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
    cnt, tt = collections.Counter(), collections.Counter()
    for c in cpu:
        if not c.kernels or any(ch.kernels for ch in c.cpu_children):
            continue                                      # (the innermost op that owns the launches)
        chain, p = [], c
        while p is not None:
            chain.append(p)
            p = p.cpu_parent
        node = next((x.name for x in chain if x.name.startswith("autograd::engine::evaluate_function")), None)
        if node:
            where = "bwd " + node.split(": ", 1)[-1]
        else:
            frames = []
            for x in chain:
                for fr in (x.stack or []):
                    if "/eda_amd/" in fr or "bench.py" in fr:
                        frames.append(re.sub(r".*/(eda_amd/|bench\.py)", r"\1", fr))
                if frames:
                    break
            where = "fwd " + (frames[0] if frames else chain[-1].name)
        for k in c.kernels:
            name = short(k.name)
            if NATIVE.match(name) or name.startswith("Memcpy") or name.startswith("Memset"):
                continue
            key = (where[:110], c.name[:40], re.sub(r"<.*", "", name)[:40])
            cnt[key] += 1
            tt[key] += k.duration
    print("non-library launches of one eager step: %d, %.1f us of kernels" % (sum(cnt.values()), sum(tt.values())))
    for k, c in cnt.most_common(top):
        print("%3d x %7.1f us  %-70s %-28s %s" % (c, tt[k], k[0], k[1], k[2]))


if __name__ == "__main__":
    main()
