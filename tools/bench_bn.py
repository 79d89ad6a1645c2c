"""BN+ReLU forward/backward (csrc/sa_cl.hip) alone at one (rows, channels) shape, for rocprofv3
kernel-trace / PMC passes:  R=1048576 C=64 ITERS=5 python tools/bench_bn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd.sa_ops import BNReLUCL  # noqa: E402


def main():
    R, C, iters = int(os.environ.get("R", 1048576)), int(os.environ.get("C", 64)), int(os.environ.get("ITERS", 5))
    z = torch.randn(R, C, device="cuda", requires_grad=True)
    g = torch.ones(C, device="cuda", requires_grad=True)
    b = torch.zeros(C, device="cuda", requires_grad=True)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    w = torch.randn(R, C, device="cuda")
    for _ in range(iters):
        out = BNReLUCL.apply(z, g, b, rm, rv, 1e-5, 0.1, True, 1)
        out.backward(w)
        z.grad = None
    torch.cuda.synchronize()
    print("done", R, C)


if __name__ == "__main__":
    main()
