"""GPU check of gemm_dma_kernel against gemm_rows_kernel (EDA_GEMM_DMA=0) on the shapes of the path: both compute the same
fp32 MFMA products in the same k order within a 16-chunk, so the results agree to rounding of the chunk order (bitwise in
practice).  usage: python tools/check_gemm_dma.py"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from eda_amd import gemm
    torch.manual_seed(0)
    outs = []
    for R, K, N in [(2048, 288, 288), (8192, 288, 576), (640, 768, 2304), (640, 3072, 768), (2048, 288, 256), (2048, 256, 288),
                    (1000, 64, 100), (37, 32, 4), (8192, 288, 864), (130 * 8, 288, 288)]:
        x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
        dy = torch.randn(R, N, device="cuda")
        y = gemm.linear_fwd(x, w, b); dx = gemm.linear_dgrad(dy, w)
        ref = x.double() @ w.double().t() + b.double(); refdx = dy.double() @ w.double()
        outs.append((y.cpu(), dx.cpu(), float((y - ref).abs().max() / ref.abs().max()), float((dx - refdx).abs().max() / refdx.abs().max())))
    torch.save(outs, sys.argv[1])
else:
    import torch
    res = {}
    for mode in range(0, 11):
        env = dict(os.environ, EDA_GEMM_DMA=str(mode))
        f = f"/tmp/gd_{mode}.pt"
        subprocess.check_call([sys.executable, __file__, f], env=env)
        res[mode] = torch.load(f)
    base = res[0]
    ok = True
    for key, r in res.items():
        for i, (y, dx, e1, e2) in enumerate(r):
            dy_ = float((y - base[i][0]).abs().max()); ddx = float((dx - base[i][1]).abs().max())
            good = e1 < 1e-5 and e2 < 1e-5
            ok = ok and good
            print(key, i, "vs fp64: %.2e %.2e | vs rows kernel: %.2e %.2e" % (e1, e2, dy_, ddx), "" if good else "BAD")
    print("ALL OK" if ok else "FAILED")
