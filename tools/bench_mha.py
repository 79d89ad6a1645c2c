"""Micro-benchmark of the fused attention kernels at EDA's nine (Lq, Lk) shapes, B=8."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eda_amd import attention  # noqa: E402

attention.set_compute_dtype(os.environ.get("ATTN_DTYPE", "f32"))      # f32 | bf16 | f16 contractions

SHAPES = [(1024, 1024, "enc self-vis"), (80, 80, "enc self-lang"), (80, 1024, "enc cross_lv"),
          (1024, 80, "enc cross_vl"), (1024, 132, "enc cross_d"), (256, 256, "dec self"),
          (256, 80, "dec cross_l"), (256, 132, "dec cross_d"), (256, 1024, "dec cross_v"),
          (130, 1024, "enc cross_lv, 130 tokens"), (1024, 130, "enc cross_vl, 130 tokens")]


def main():
    only = os.environ.get("ONLY")
    iters = int(os.environ.get("ITERS", 20))
    B = 8
    for lq, lk, name in SHAPES:
        if only and only != f"{lq}x{lk}":
            continue
        q = torch.randn(B, lq, 288, device="cuda", requires_grad=True)
        k = torch.randn(B, lk, 288, device="cuda", requires_grad=True)
        v = torch.randn(B, lk, 288, device="cuda", requires_grad=True)
        w = torch.randn(B, lq, 288, device="cuda")
        p = float(os.environ.get("PDROP", 0.1))

        def fwd():
            return attention.attention_core(q, k, v, None, 8, p, 1)
        o = fwd()
        o.backward(w)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        for _ in range(iters):
            o = fwd()
        e[1].record()
        for _ in range(iters):
            o.backward(w, retain_graph=True)
        e[2].record()
        torch.cuda.synchronize()
        tf, tb = e[0].elapsed_time(e[1]) / iters, e[1].elapsed_time(e[2]) / iters
        ff = 4.0 * B * 8 * lq * lk * 36
        print(f"{name:14s} Lq={lq:5d} Lk={lk:5d} fwd {tf*1e3:7.1f} us {ff/tf/1e9:6.1f} TF/s | bwd {tb*1e3:7.1f} us "
              f"{2.5*ff/tb/1e9:6.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
