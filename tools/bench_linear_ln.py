"""The fused linear + residual + Dropout + LayerNorm launch next to its two-launch composition (for rocprofv3):
usage: python tools/bench_linear_ln.py R K"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import fused_ln, gemm
R, K = int(sys.argv[1]), int(sys.argv[2])
C = 288
inp = torch.randn(R, K, device="cuda"); W = torch.randn(C, K, device="cuda"); b = torch.randn(C, device="cuda")
x = torch.randn(R, C, device="cuda"); pos = torch.randn(R, C, device="cuda")
norm = torch.nn.LayerNorm(C).cuda()
with torch.no_grad():
    for _ in range(20):
        fused_ln._linear_ln_forward(inp, W, b, x, norm.weight, norm.bias, norm.eps, 0.1, 5, pos)
        y = gemm.linear_fwd(inp, W)
        fused_ln.add_dropout_layer_norm(x, y, norm, 0.1, True, 5, y_bias=b, pos=pos)
torch.cuda.synchronize()
