"""GPU time of the forward / input-gradient GEMMs of the pointwise layers as torch issues them
(Y = X W^T + b, dX = dY W), timed inside a HIP graph:  python tools/bench_fwd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_dw import timed  # noqa: E402

SHAPES = [  # (rows R, Cout, Cin)
    (2048, 288, 288), (8192, 288, 288), (640, 288, 288), (2048, 864, 288), (8192, 864, 288), (8192, 576, 288),
    (2048, 576, 288), (2048, 256, 288), (2048, 288, 256), (8192, 256, 288), (8192, 288, 256), (1056, 576, 288),
    (2048, 64, 288), (2048, 3, 288),
]


def main():
    dev = torch.device("cuda", 0)
    for R, Co, Ci in SHAPES:
        x = torch.randn(R, Ci, device=dev)
        W = torch.randn(Co, Ci, device=dev)
        b = torch.randn(Co, device=dev)
        dy = torch.randn(R, Co, device=dev)
        y = torch.empty(R, Co, device=dev)
        dx = torch.empty(R, Ci, device=dev)
        t_f = timed(lambda: torch.addmm(b, x, W.t(), out=y))
        t_fn = timed(lambda: torch.mm(x, W.t(), out=y))
        t_dx = timed(lambda: torch.mm(dy, W, out=dx))
        gf = 2e-6 * R * Co * Ci
        print(f"R={R:5d} Cout={Co:4d} Cin={Ci:4d}: addmm {t_f:6.1f}us ({gf / t_f:5.1f} TF/s)  mm {t_fn:6.1f}us  "
              f"dX {t_dx:6.1f}us ({gf / t_dx:5.1f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
