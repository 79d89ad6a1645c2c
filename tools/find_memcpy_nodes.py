"""Which aten::copy_ calls of one eager training step become device-to-device memcpy launches (in a captured step: memcpy
nodes, ~7 us of queue time each) -- torch.profiler with stacks.    python tools/find_memcpy_nodes.py"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from eda_amd.bdetr import BeaUTyDETR
from eda_amd.parallel import FlatParams
from torch.profiler import ProfilerActivity, profile
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = BeaUTyDETR().to(dev).train(); model.text_encoder.eval()
grads = FlatParams(model)
opt = torch.optim.AdamW(list(grads.groups.values()), lr=1e-4, weight_decay=5e-4, fused=True)
inputs = bench.make_inputs(0, 8, dev, 50000, 80)
def step():
    loss = bench.synthetic_loss(model(inputs))
    with grads.deferred_wgrad():
        loss.backward()
    grads.collect_grads(); grads.clip_grad_norm_(0.1); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ev = prof.events()
agg = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and ("copy_" in e.name or "clone" in e.name or "contiguous" in e.name):
        ks = [k.name for k in e.kernels] if hasattr(e, "kernels") else []
        if any("Memcpy" in k or "copyBuffer" in k for k in ks):
            st = [s for s in (e.stack or []) if "/eda_amd/" in s or "bench.py" in s]
            agg[(e.name, st[0][-70:] if st else "?", tuple(e.input_shapes[0]) if e.input_shapes else None)] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]): print(v, k)
print("all memcpy-like kernels:", collections.Counter(e.name[:40] for e in ev if e.device_type == torch.autograd.DeviceType.CUDA and ("emcpy" in e.name or "copyBuffer" in e.name)))
