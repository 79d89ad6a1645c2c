"""What a second busy queue costs the kernels of the first: a replayed graph of 300 small products on stream A, alone and
next to (b) ONE long single-workgroup kernel, (c) a long furthest-point-sampling launch (104 spinning workgroups),
(d) a stream of short element-wise kernels on stream B.  us per launch of the graph on A."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import gemm, pointnet2_utils  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    A, B = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
    x = torch.randn(2048, 288, device=dev); w = torch.randn(288, 288, device=dev); y = torch.empty(2048, 288, device=dev)
    xyz = torch.rand(8, 50000, 3, device=dev)
    small = torch.randn(1 << 16, device=dev)
    n = 300
    with torch.cuda.stream(A):
        gemm.linear_fwd(x, w, out=y); torch.cuda.synchronize()
        gA = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gA, stream=A):
            for _ in range(n):
                gemm.linear_fwd(x, w, out=y)
    with torch.cuda.stream(B):
        pointnet2_utils.furthest_point_sample(xyz, 2048); torch.cuda.synchronize()
        g_fps = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_fps, stream=B):
            pointnet2_utils.furthest_point_sample(xyz, 2048)
        g_small = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_small, stream=B):
            for _ in range(600):
                small.add_(1.0)
    torch.cuda.synchronize()

    def side_sleep():
        torch.cuda._sleep(int(2.4e9 * 0.004))       # ~4 ms, one workgroup

    def run(side):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if side is not None:
            with torch.cuda.stream(B):
                s0.record(B); side(); s1.record(B)
        with torch.cuda.stream(A):
            e0.record(A); gA.replay(); e1.record(A)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n, (s0.elapsed_time(s1) if side is not None else 0.0)

    for name, side in (("alone", None), ("one sleeping workgroup", side_sleep), ("fps 8 x 50000 -> 2048", g_fps.replay),
                       ("600 short element-wise kernels", g_small.replay)):
        r = [run(side) for _ in range(5)][1:]
        print("%-34s A: %.2f us / launch (min %.2f)   B busy %.2f ms" % (name, sum(t for t, _ in r) / len(r), min(t for t, _ in r), r[-1][1]))


if __name__ == "__main__":
    main()
