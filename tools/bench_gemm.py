"""Time the repo's row GEMMs against torch (hipBLASLt) on the path's shapes.  GPU only."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import gemm  # noqa: E402

SMALL = [(2048, 288, 288), (2048, 288, 864), (2048, 288, 576), (2048, 288, 256), (2048, 256, 288), (640, 288, 576), (640, 288, 288), (1056, 288, 576),
         (8192, 288, 288), (8192, 288, 576), (8192, 288, 864), (8192, 288, 256), (8192, 256, 288), (2048, 288, 64), (640, 768, 288)]
SHAPES = SMALL if os.environ.get('EDA_BENCH_SMALL') else [(1048576, 64, 64), (1048576, 64, 128), (262144, 128, 128), (262144, 128, 256), (65536, 256, 128),
          (65536, 128, 256), (8192, 288, 864), (8192, 288, 288), (8192, 288, 256), (8192, 256, 288),
          (2048, 288, 288), (2048, 288, 864), (640, 288, 576), (8192, 512, 256), (4096, 512, 256)]


def timeit(fn, n=20):
    """us per call, launches replayed from a HIP graph (no host launch overhead in the number)."""
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * n) * 1e3


for R, K, N in SHAPES:
    x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda"); dy = torch.randn(R, N, device="cuda")
    y = torch.empty(R, N, device="cuda"); dx = torch.empty(R, K, device="cuda")
    fl = 2.0 * R * K * N
    t1 = timeit(lambda: gemm.linear_fwd(x, w, out=y))
    t2 = timeit(lambda: torch.mm(x, w.t(), out=y))
    t3 = timeit(lambda: gemm.linear_dgrad(dy, w, out=dx))
    t4 = timeit(lambda: torch.mm(dy, w, out=dx))
    print(f"R={R:8d} K={K:4d} N={N:4d} | fwd own {t1:8.1f} us {fl/t1/1e6:6.1f} TF  lib {t2:8.1f} us {fl/t2/1e6:6.1f} TF |"
          f" dgrad own {t3:8.1f} us {fl/t3/1e6:6.1f} TF  lib {t4:8.1f} us {fl/t4/1e6:6.1f} TF", flush=True)
