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
inputs = bench.make_inputs(0, 8, dev, 50000, 80)
for _ in range(2):
    with grads.deferred_wgrad():
        bench.synthetic_loss(model(inputs)).backward()
    grads.collect_grads()
with grads.deferred_wgrad():
    bench.synthetic_loss(model(inputs)).backward()
have = [(v, p.grad, n) for (n, p), v in zip([(n, p) for n, p in model.named_parameters() if p.requires_grad], grads._grad_views) if p.grad is not None] if len(grads._grad_views) == len([p for p in model.parameters() if p.requires_grad]) else None
pg = [(p.grad, v) for v, p in zip(grads._grad_views, grads.params) if p.grad is not None]
print("params with autograd grads:", len(pg))
bad = [(tuple(g.shape), g.stride(), tuple(v.shape), v.stride(), g.dtype) for g, v in pg if not (g.is_contiguous() and g.shape == v.shape and g.stride() == v.stride())]
print("not matching:", len(bad), bad[:10])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    grads.collect_grads()
    torch.cuda.synchronize()
c = collections.Counter(e.name[:60] for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
print(c)
