"""The device-copy kernel (achievable HBM rate of bench.py) next to torch copy_ on 2 x 1 GiB."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import ext  # noqa: E402

n = 1 << 28
src = torch.full((n,), 1.0, device="cuda")
dst = torch.empty_like(src)
for _ in range(2):
    ext.device_copy(src, dst)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ext.device_copy(src, dst)
e1.record()
torch.cuda.synchronize()
print("eda_device_copy_f32",
      round(2.0 * 4 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1), "GB/s", flush=True)
e0.record()
for _ in range(10):
    dst.copy_(src)
e1.record()
torch.cuda.synchronize()
print("torch copy_", round(2.0 * 4 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1), "GB/s")
