"""Phase timeline of csrc/mha4.hip (keys-per-wave attention forward): builds the -DEDA_MHA4_PROFILE variant, runs one shape a
few times and prints, over the workgroups of the LAST launch, when each phase boundary was passed relative to the earliest
workgroup start (wall-clock stamps, 10 ns ticks).  usage: python tools/mha4_phase_profile.py [Lq Lk]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eda_amd import build  # noqa: E402

extra = os.environ.get("MHA4_DEFS", "").split()
lib = build.build_variant("mha4prof" + "".join(x.replace("-D", "_").replace("=", "") for x in extra), ["-DEDA_MHA4_PROFILE"] + extra)
os.environ["EDA_HIP_LIB"] = lib
import torch  # noqa: E402
from eda_amd import _lib, attention  # noqa: E402

Lq, Lk = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (80, 1024)
B = 8
q, k, v = (torch.randn(B, n, 288, device="cuda") for n in (Lq, Lk, Lk))
for _ in range(5):
    attention.attention_core(q, k, v, None, 8, 0.1, 1)
torch.cuda.synchronize()
L = ctypes.CDLL(lib)
nwg = B * 8 * ((Lk + 255) // 256)
buf = (ctypes.c_ulonglong * (1024 * 8))()
assert L.eda_mha4_profile_read(buf, 1024 * 8) == 0
import numpy as np  # noqa: E402
t = np.array(buf[:], dtype=np.float64).reshape(1024, 8)[:min(nwg, 1024)]
t0 = t[:, 0].min()
names = ["entry", "operand loads issued, Q staged", "after first barrier", "after compute + LDS merge + publish", "stores drained",
         "ticket taken", "last arriver done", "wave 0: its tiles of the (last) batch computed"]
for i, n in enumerate(names):
    col = t[:, i]
    col = col[col >= t0]            # (slot 6: only the last arrivers of THIS launch wrote it)
    col = col[col < t0 + 1e5]
    if len(col):
        print("%-40s n %4d  min %7.2f  median %7.2f  max %7.2f us" % (n, len(col), (col.min() - t0) / 100, (np.median(col) - t0) / 100,
                                                                  (col.max() - t0) / 100))
