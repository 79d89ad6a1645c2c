"""HIP-event time of the fused linear + residual + Dropout + LayerNorm launch and of the plain row products around it
(back-to-back launches: what a replayed graph sees).  usage: python tools/time_linear_ln.py [R] [K]
EDA_GEMM_LN_VAR=63|64|92|94 selects an experiment variant of the 16-row kernel (csrc/gemm.hip)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import fused_ln, gemm
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K = int(sys.argv[2]) if len(sys.argv) > 2 else 288
C = 288
inp = torch.randn(R, K, device="cuda"); W = torch.randn(C, K, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
x = torch.randn(R, C, device="cuda"); pos = torch.randn(R, C, device="cuda")
norm = torch.nn.LayerNorm(C).cuda()


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    ref = fused_ln._linear_ln_forward(inp, W, b, x, norm.weight, norm.bias, norm.eps, 0.0, 5, pos)[0]
    y = torch.nn.functional.layer_norm(x + inp @ W.t() + b, (C,), norm.weight, norm.bias, norm.eps)
    err = float((ref - y).abs().max())
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fused_ln._linear_ln_forward(inp, W, b, x, norm.weight, norm.bias, norm.eps, 0.1, 5, pos)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(50):
                fused_ln._linear_ln_forward(inp, W, b, x, norm.weight, norm.bias, norm.eps, 0.1, 5, pos)
    t_ln = timeit(g.replay, 20) / 50
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g2, stream=s):
            for _ in range(50):
                gemm.linear_fwd(inp, W, b)
    t_mm = timeit(g2.replay, 20) / 50
print(f"R={R} K={K} var={os.environ.get('EDA_GEMM_LN_VAR', '0')}: fused linear+LN {t_ln:.2f} us, plain product {t_mm:.2f} us (graph replay of 50 launches), max err vs torch {err:.2e}")
