"""us per launch of the many-row plain products: gemm_b3_rows_kernel (bf16 x 3; EDA_GEMM_B3ROWS=2: every eligible shape) next to
gemm_dma_kernel (EDA_GEMM_B3ROWS=0, fp32 MFMA), 50 launches back to back in a replayed hipGraph, and the error of both
against fp64.    python tools/bench_gemm_b3rows.py ["RxKxN ..."]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm_kc96 import timed  # noqa: E402
from eda_amd import _lib, gemm  # noqa: E402

DEFAULT = "8192x288x288 8192x288x576 8192x576x288 8192x288x864 4096x288x288 16384x288x288"


def main():
    shapes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[1] if len(sys.argv) > 1 else DEFAULT).split()]
    L = _lib.lib()
    print("%-16s %9s %9s   %9s %9s  (us per launch in a replayed graph | max error vs fp64 / max |ref|)" %
          ("R x K x N", "fp32 MFMA", "bf16 x 3", "err fp32", "err b3"))
    for R, K, N in shapes:
        x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
        y = torch.empty(R, N, device="cuda")
        ref = x.double() @ w.double().t() + b.double()
        row, err = [], []
        for mode in ("0", "2"):
            os.environ["EDA_GEMM_B3ROWS"] = mode
            L.eda_reload_env()
            row.append(timed(lambda: gemm.linear_fwd(x, w, b, out=y)))
            err.append(float((y.double() - ref).abs().max() / ref.abs().max()))
        fl = 2.0 * R * K * N
        print("%-16s %9.2f %9.2f   %9.2e %9.2e   %.1f -> %.1f TFLOP/s" % ("%dx%dx%d" % (R, K, N), row[0], row[1], err[0], err[1],
                                                                           fl / row[0] * 1e-6, fl / row[1] * 1e-6), flush=True)
    os.environ.pop("EDA_GEMM_B3ROWS", None)
    L.eda_reload_env()


if __name__ == "__main__":
    main()
