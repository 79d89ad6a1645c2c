"""The text encoder's four linear-layer shapes (640 rows): bf16 x 3 planes of the frozen weight (csrc/gemm_frozen.hip) against the
fp32-MFMA row product (csrc/gemm.hip) and the library (torch.matmul -> hipBLASLt), us per launch: 48 launches over 4 rotating
operand sets in a replayed hipGraph.  usage: python tools/bench_gemm_frozen.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eda_amd import gemm  # noqa: E402


def timed(fn, n=48):
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s), torch.no_grad():
        for i in range(4):
            fn(i)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                fn(i)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (10 * n) * 1e3


def main():
    dev = "cuda"
    print("R x K x N            bf16x3 planes   fp32 MFMA   hipBLASLt   (us per launch)")
    for R, K, N, act in [(640, 768, 2304, 0), (640, 768, 768, 0), (640, 768, 3072, 2), (640, 3072, 768, 0), (1040, 768, 2304, 0),
                         (1040, 3072, 768, 0)]:
        xs = [torch.randn(R, K, device=dev) for _ in range(4)]
        ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(4)]
        b = torch.randn(N, device=dev)
        pls = [gemm.frozen_planes(w) for w in ws]
        ys = [torch.empty(R, N, device=dev) for _ in range(4)]
        t_b3 = timed(lambda i: gemm.linear_frozen(xs[i % 4], pls[i % 4], b, act=act, out=ys[i % 4]))
        t_32 = timed(lambda i: gemm.linear_fwd(xs[i % 4], ws[i % 4], b, relu=act, out=ys[i % 4]))
        t_lib = timed(lambda i: torch.addmm(b, xs[i % 4], ws[i % 4].t(), out=ys[i % 4]))
        fl = 2.0 * R * K * N
        print("%5d x %4d x %4d   %8.1f (%5.1f TF)  %8.1f   %8.1f" % (R, K, N, t_b3, fl / t_b3 / 1e6, t_32, t_lib))


if __name__ == "__main__":
    main()
