"""A second process that keeps the GPU's CUs, LDS and HBM busy for `seconds`: LDS-heavy products (the DMA-staged GEMM, attention),
HBM-heavy copies.  Run the GPU test-suite next to it to screen kernels for races that an idle GPU hides
(profiles/r05_lds_dma_races.md):   python tools/background_load.py 240 &  python -m pytest tests -m gpu -q"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import attention, gemm  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    dev = torch.device("cuda", 0)
    x = torch.randn(8192, 288, device=dev); w = torch.randn(288, 288, device=dev)
    x2 = torch.randn(640, 3072, device=dev); w2 = torch.randn(768, 3072, device=dev)
    q = torch.randn(8, 1024, 288, device=dev); k = torch.randn(8, 1024, 288, device=dev)
    big = torch.empty(1 << 26, device=dev); big2 = torch.empty_like(big)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            gemm.linear_fwd(x, w); gemm.linear_fwd(x2, w2); attention.mha_core_fwd(q, k, k, 8) if hasattr(attention, "mha_core_fwd") else None
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                gemm.linear_fwd(x, w); gemm.linear_fwd(x2, w2); big2.copy_(big)
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        with torch.cuda.stream(s):
            g.replay()
        n += 1
        if n % 50 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("background load: %d replays in %.0f s" % (n, time.time() - t0))


if __name__ == "__main__":
    main()
