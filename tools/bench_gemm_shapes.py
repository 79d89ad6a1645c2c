"""The tiled row-GEMM shapes of the step, one after the other, 20 launches each of the forward form (and the library's
torch.mm beside it) -- run under `rocprofv3 --kernel-trace` by tools/prof_gemm_shapes.sh, which prints per shape the
average kernel duration and the fraction of the fp32 MFMA peak.  SHAPES="RxKxN ..." selects; FORM=dgrad times dX."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import gemm  # noqa: E402

DEFAULT = "2048x288x288 640x3072x768 8192x288x288 640x768x3072 640x768x2304 640x768x768 640x288x288 2048x288x256 " \
          "8192x288x576 8192x576x288 640x3456x288 1056x3456x288 2048x288x864"


def main():
    shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", DEFAULT).split()]
    iters = int(os.environ.get("ITERS", 20))
    lib = os.environ.get("LIB", "1") == "1"
    for R, K, N in shapes:
        x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
        y = torch.empty(R, N, device="cuda")
        torch.cuda.synchronize()
        # marker launch: a zero-size-independent tiny kernel whose name the summary script uses to cut the trace per shape
        torch.zeros(R * 1000 + K, device="cuda", dtype=torch.int8)     # (fill kernel with a unique element count)
        for _ in range(iters):
            gemm.linear_fwd(x, w, b, out=y)
        if lib:
            for _ in range(iters):
                torch.addmm(b, x, w.t(), out=y)
        torch.cuda.synchronize()
        print("shape", R, K, N, flush=True)


if __name__ == "__main__":
    main()
