"""Phase timeline of csrc/loss.hip's alignment kernel at the bench shapes: builds the -DEDA_LOSS_PROFILE variant, runs the loss a few
times and prints, over the workgroups of the LAST launch, when thread 0 passed each phase boundary relative to the earliest start
(wall-clock stamps, 10 ns ticks).  usage: python tools/loss_phase_profile.py [tokens]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eda_amd import build  # noqa: E402

lib = build.build_variant("lossprof", ["-DEDA_LOSS_PROFILE"])
os.environ["EDA_HIP_LIB"] = lib
import numpy as np  # noqa: E402
import torch  # noqa: E402
from eda_amd import losses_fused as F  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 80
P, B, Q, G = 7, 8, 256, 132
torch.manual_seed(0)
dev = "cuda"
sim = torch.randn(P * B, Q, L, device=dev)
nt = torch.randint(1, 9, (B,), device=dev, dtype=torch.int32)
tq = torch.full((P * B, Q), -1, dtype=torch.long, device=dev)
for pb in range(P * B):
    n = int(nt[pb % B])
    tq[pb, torch.randperm(Q, device=dev)[:n]] = torch.arange(n, device=dev)
maps = [(torch.rand(B, G, 256, device=dev) < 0.05).float() for _ in range(5)]
am = torch.ones(B, L, dtype=torch.long, device=dev)
nb = nt.sum().float().reshape(1)
for _ in range(5):
    F._SemAlign.apply(sim, tq, nb, 0.1, am, *maps)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 8))()
assert ctypes.CDLL(lib).eda_loss_profile_read(buf, 256 * 8) == 0
t = np.array(buf[:], dtype=np.float64).reshape(256, 8)[:P * B]
t0 = t[:, 0].min()
for i, n in enumerate(["entry", "logits in LDS, slots, counts", "phase 1 (rows) done by wave 0", "phase 2a (column lse) done by wave 0",
                       "phase 2b (column sums) done", "phase 3 (gradient) done by wave 0", "exit"]):
    col = t[:, i]
    print("%-40s min %7.2f  median %7.2f  max %7.2f us" % (n, (col.min() - t0) / 100, (np.median(col) - t0) / 100, (col.max() - t0) / 100))
