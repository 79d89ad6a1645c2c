"""The grouped weight-gradient kernel (csrc/wgrad.hip) on a workload shaped like the bench step's flush, in both
arithmetic modes (eda_wgrad_set_arith: 0 = fp32 MFMA, 1 = bf16 x 3): time per launch (HIP events, 20 launches) and the
largest error against fp64 relative to the largest entry / to sum |dy|^T |x| (the bound of tests/test_wgrad_gpu.py).
    python tools/bench_wgrad_grouped.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eda_amd import _lib
from eda_amd.wgrad_queue import WgradQueue

WORK = [((288, 288), [2048]) for _ in range(90)] + [((288, 288), [8192]) for _ in range(20)] + \
       [((864, 288), [8192]) for _ in range(3)] + [((576, 288), [8192]) for _ in range(6)] + \
       [((256, 288), [2048]) for _ in range(13)] + [((288, 256), [2048]) for _ in range(12)] + \
       [((64, 288), [14336]), ((288, 64), [14336]), ((288, 288), [2048, 640, 1056]), ((128, 128), [1056]), ((100, 36), [333])]

torch.manual_seed(0)
dev = "cuda"
tot = sum(m * n + m for (m, n), _ in WORK)
params = torch.zeros(tot, device=dev)
grads = torch.zeros_like(params)


def locate(t):
    off = (t.data_ptr() - params.data_ptr()) // 4
    return grads[off:off + t.numel()].view(t.shape)


jobs = []
off = 0
flops = 0.0
for (m, n), Ks in WORK:
    W = params[off:off + m * n].view(m, n); off += m * n
    b = params[off:off + m]; off += m
    js = []
    for K in Ks:
        js.append((torch.randn(K, m, device=dev), torch.randn(K, n, device=dev)))
        flops += 2.0 * m * n * K
    jobs.append((W, b, js))


def run():
    q = WgradQueue(locate)
    q.reserve(torch.device(dev))
    for W, b, js in jobs:
        for dy, x in js:
            assert q.submit(W, b, dy, x)
    q.flush()


res = {}
for mode in (0, 1):
    assert _lib.lib().eda_wgrad_set_arith(mode) == 0
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # (the host builds the descriptors inside run(): bracket the launches only by running them back to back)
    ts = []
    for _ in range(10):
        q = WgradQueue(locate); q.reserve(torch.device(dev))
        for W, b, js in jobs:
            for dy, x in js:
                q.submit(W, b, dy, x)
        torch.cuda.synchronize()
        e0.record(); q.flush(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    worst_rel, worst_bound = 0.0, 0.0
    for W, b, js in jobs[::7] + jobs[-5:]:
        eW = sum(dy.double().t() @ x.double() for dy, x in js)
        bound = sum(dy.abs().double().t() @ x.abs().double() for dy, x in js).max().item()
        err = (locate(W).double() - eW).abs().max().item()
        worst_rel = max(worst_rel, err / eW.abs().max().item())
        worst_bound = max(worst_bound, err / bound)
        eb = sum(dy.double().sum(0) for dy, x in js)
        assert (locate(b).double() - eb).abs().max().item() <= 2e-5 * sum(dy.abs().double().sum(0) for dy, x in js).max().item()
    res[mode] = (ms, worst_rel, worst_bound)
    print("arith %d: %.3f ms per flush (incl. descriptor upload) = %.1f TFLOP/s; max err / max|dW| %.2e; max err / (|dy|^T |x|) %.2e"
          % (mode, ms, flops / ms / 1e9, worst_rel, worst_bound))
_lib.lib().eda_wgrad_set_arith(-1)
print("GFLOP per flush: %.1f; speed-up %.2fx; error ratio bf16x3 / fp32: %.2f" % (flops / 1e9, res[0][0] / res[1][0], res[1][2] / res[0][2]))
