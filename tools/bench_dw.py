"""GPU time of the weight-gradient GEMMs dW = dY^T X (small output, long K) as torch issues
them, timed inside a HIP graph (no launch gaps):  python tools/bench_dw.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd.nn_utils import colsum, wgrad  # noqa: E402

SHAPES = [  # (K rows, M = Cout, N = Cin)
    (2048, 288, 288), (8192, 288, 288), (640, 288, 288), (2048, 864, 288), (8192, 864, 288),
    (8192, 576, 288), (2048, 576, 288), (2048, 256, 288), (2048, 288, 256), (8192, 256, 288),
    (1056, 576, 288), (2048, 64, 288), (2048, 3, 288),
]


def timed(fn, reps=50):
    fn()
    torch.cuda.synchronize()
    if os.environ.get("EAGER"):          # for rocprofv3 runs (no graph replay under the profiler)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3     # us


def main():
    dev = torch.device("cuda", 0)
    shapes = SHAPES
    if os.environ.get("ONLY"):
        shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["ONLY"].split(",")]
    for K, M, N in shapes:
        dy = torch.randn(K, M, device=dev)
        x = torch.randn(K, N, device=dev)
        out = torch.empty(M, N, device=dev)
        t_mm = timed(lambda: torch.mm(dy.t(), x, out=out))
        outT = torch.empty(N, M, device=dev)
        t_mmT = timed(lambda: torch.mm(x.t(), dy, out=outT))
        t_sum = timed(lambda: dy.sum(0))
        db = torch.empty(M, device=dev)
        t_wg = timed(lambda: wgrad(dy, x, dW=out, db=db)) if M >= 32 else float("nan")
        t_cs = timed(lambda: colsum(dy, out=db))
        res = [f"wgrad+db {t_wg:6.1f}us ({2e-6 * K * M * N / t_wg:5.1f} TF/s)", f"colsum {t_cs:5.1f}us", f"mm {t_mm:6.1f}us ({2e-6 * K * M * N / t_mm:5.1f} TF/s)", f"mmT {t_mmT:6.1f}us", f"sum {t_sum:5.1f}us"]
        for s in ():
            if K % s == 0 and K // s >= 128:
                t = timed(lambda: torch.bmm(dy.view(s, K // s, M).transpose(1, 2), x.view(s, K // s, N)).sum(0))
                res.append(f"splitK{s} {t:6.1f}us")
        print(f"K={K:5d} M={M:4d} N={N:4d}: " + "  ".join(res), flush=True)


if __name__ == "__main__":
    main()
