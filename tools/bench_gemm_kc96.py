"""us per launch of the row products with 96-wide chunks (EDA_GEMM_KC96 = tile configuration) next to the default
dispatch, 50 launches back to back in a replayed hipGraph (what a launch costs inside the captured step).
    python tools/bench_gemm_kc96.py ["RxKxN ..."]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import _lib, gemm  # noqa: E402

DEFAULT = "2048x288x288 640x288x288 2048x288x576 2048x576x288 2048x288x864 8192x288x288 8192x288x576 8192x576x288 2048x864x288 1024x288x288 512x288x288"


def timed(fn, n=50, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


def main():
    shapes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[1] if len(sys.argv) > 1 else DEFAULT).split()]
    L = _lib.lib()
    cfgs = [-1, 0] + [int(c) for c in os.environ.get("CFGS", "1 2 3 4 5 6 7 8").split()]
    print("%-16s %-6s " % ("R x K x N", "form") + " ".join("%7s" % ("dflt" if c < 0 else "kc96=%d" % c) for c in cfgs))
    for R, K, N in shapes:
        x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
        y = torch.empty(R, N, device="cuda"); dy = torch.randn(R, N, device="cuda"); dx = torch.empty(R, K, device="cuda")
        for form, fn in (("fwd", lambda: gemm.linear_fwd(x, w, b, out=y)), ("dgrad", lambda: gemm.linear_dgrad(dy, w, out=dx))):
            row = []
            for c in cfgs:
                if c < 0:
                    os.environ.pop("EDA_GEMM_KC96", None)
                else:
                    os.environ["EDA_GEMM_KC96"] = str(c)
                L.eda_reload_env()
                row.append(timed(fn))
            print("%-16s %-6s " % ("%dx%dx%d" % (R, K, N), form) + " ".join("%7.2f" % t for t in row), flush=True)


if __name__ == "__main__":
    main()
