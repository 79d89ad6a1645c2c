import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import gemm
R, K, N = [int(v) for v in sys.argv[1:4]]
x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda"); dy = torch.randn(R, N, device="cuda")
y = torch.empty(R, N, device="cuda"); dx = torch.empty(R, K, device="cuda")
for _ in range(20):
    gemm.linear_fwd(x, w, out=y); gemm.linear_dgrad(dy, w, out=dx); torch.mm(x, w.t(), out=y)
torch.cuda.synchronize()
