"""Time the repo's row GEMMs against torch (hipBLASLt) on the path's shapes.  GPU only."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import gemm  # noqa: E402

SHAPES = [(1048576, 64, 64), (1048576, 64, 128), (262144, 128, 128), (262144, 128, 256), (65536, 256, 128),
          (65536, 128, 256), (8192, 288, 864), (8192, 288, 288), (8192, 288, 256), (8192, 256, 288),
          (2048, 288, 288), (2048, 288, 864), (640, 288, 576), (8192, 512, 256), (4096, 512, 256)]


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


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
