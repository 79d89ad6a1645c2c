"""Graph-replay time (50 launches back to back) of an attention site's forward as [q-projection launch + core launch]
and as the fused launch (eda_mha_qproj_fwd).  usage: python tools/time_qproj_site.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import attention, gemm

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def graph_of(fn, reps=50):
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    return g

torch.manual_seed(0)
d, H, B = 288, 8, 8
W = torch.randn(3 * d, d, device="cuda") * 0.05
b = torch.randn(3 * d, device="cuda") * 0.1
for Lq, Lk in [(256, 80), (256, 132), (1024, 80), (1024, 132), (256, 130)]:
    x = torch.randn(B, Lq, d, device="cuda")
    kv = torch.randn(B, Lk, 2 * d, device="cuda")
    k, v = kv[..., :d], kv[..., d:]
    mask = torch.zeros(B, Lk, dtype=torch.bool, device="cuda"); mask[1:, Lk - 7:] = True
    m8 = mask.view(torch.uint8)
    with torch.no_grad():
        def separate():
            q = gemm.linear_fwd(x.view(-1, d), W[:d], b[:d]).view(B, Lq, d)
            return attention.attention_core(q, k, v, mask, H, 0.1, 7)
        def fused():
            return attention._qproj_core_fwd(x, W[:d], b[:d], k, v, m8, H, 0.1, 7)
        def proj_only():
            return gemm.linear_fwd(x.view(-1, d), W[:d], b[:d])
        t_sep = timeit(graph_of(separate).replay) / 50
        t_fus = timeit(graph_of(fused).replay) / 50
        t_prj = timeit(graph_of(proj_only).replay) / 50
    fl = 4.0 * B * H * Lq * Lk * 36 + 2.0 * B * Lq * d * d
    print(f"{Lq:5d} x {Lk:4d}: projection {t_prj:6.2f} us + core = {t_sep:6.2f} us; fused {t_fus:6.2f} us ({100 * t_fus / t_sep:.0f} %), "
          f"{fl / t_fus * 1e-6:.1f} TFLOP/s = {fl / t_fus * 1e-6 / 157.3:.2f} of the fp32 MFMA peak (projection + QK^T + PV)")
