"""One row-GEMM shape, 20 rounds of forward, dX and torch.mm (for rocprofv3; tools/bench_gemm_cfgs.sh).
PAD=<floats> pads the row strides of x, w and dy (channel-camping experiments)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import gemm
R, K, N = [int(v) for v in sys.argv[1:4]]
pad = int(os.environ.get("PAD", "0"))
x = torch.randn(R, K + pad, device="cuda")[:, :K]; w = torch.randn(N, K + pad, device="cuda")[:, :K]
dy = torch.randn(R, N + pad, device="cuda")[:, :N]
y = torch.empty(R, N, device="cuda"); dx = torch.empty(R, K, device="cuda")
for _ in range(20):
    gemm.linear_fwd(x, w, out=y); gemm.linear_dgrad(dy, w, out=dx); torch.mm(x, w.t(), out=y)
torch.cuda.synchronize()
