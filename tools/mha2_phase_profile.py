"""In-kernel phase timing of the one-pass attention backward (csrc/mha2.hip built with -DEDA_MHA2_PROFILE:
`python -c "from eda_amd import build; build.build_variant('prof', ['-DEDA_MHA2_PROFILE'])"`, then run with
EDA_HIP_LIB=eda_amd/csrc/libeda_hip_prof.so).  Prints s_memtime cycles per wave and phase, averaged over all waves."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eda_amd import _lib, attention  # noqa: E402

NAMES = ["prologue (K rows, first chunk staged)", "stage next chunk (lse, delta, DMA issue)", "phase A (S, dP, dS, dV, dK)",
         "barrier 1", "phase B (dQ)", "barrier 2", "epilogue"]


def main():
    L = _lib.lib()
    fn = L.eda_mha2_profile_read
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
    buf = (ctypes.c_ulonglong * 16)()
    shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ.get("SHAPES", "1024x1024 256x1024 1024x80 1024x132 256x256").split()]
    p = float(os.environ.get("PDROP", 0.1))
    for lq, lk in shapes:
        q = torch.randn(8, lq, 288, device="cuda", requires_grad=True)
        k = torch.randn(8, lk, 288, device="cuda", requires_grad=True)
        v = torch.randn(8, lk, 288, device="cuda", requires_grad=True)
        w = torch.randn(8, lq, 288, device="cuda")
        o = attention.attention_core(q, k, v, None, 8, p, 1)
        o.backward(w, retain_graph=True)
        torch.cuda.synchronize()
        fn(buf)
        for _ in range(3):
            o.backward(w, retain_graph=True)
        torch.cuda.synchronize()
        fn(buf)
        waves = buf[7]
        tot = sum(buf[i] for i in range(7))
        print(f"Lq={lq} Lk={lk}: {waves // 3} waves per launch, {tot / waves:.0f} cycles per wave")
        for i, n in enumerate(NAMES):
            print(f"   {n:45s} {buf[i] / waves:10.0f}  {100.0 * buf[i] / tot:5.1f} %")


if __name__ == "__main__":
    main()
