"""W^T shadows of the linear weights, so that autograd's input gradient dX = dY W runs in the same row-GEMM form
as the forward (csrc/gemm.hip "NT": the weight is read fastest along the contraction, 16-byte fragments) instead of the
slower form that contracts over the weight's row index (13.5 vs 10.4 us on the 2048 x 288 x 288 launches of the
decoder, 135 of them per step).

A shadow covers a fixed list of 2-D fp32 parameters (the `[N, K]` weights of nn.Linear / 1x1 convolutions / attention
in-projections); `refresh()` rewrites all transposes with ONE launch (csrc/capi.hip: eda_transpose_batch_f32).
`FlatParams.deferred_wgrad()` refreshes it on entry and makes it `active` for the duration of the backward pass --
between a forward and its backward the weights do not change, so a shadow refreshed there is never stale.  Outside
that context `lookup()` is not consulted and the input gradients take the plain form.
"""
import bisect

import torch

from . import _lib

active = None


class TransposedShadow:
    def __init__(self, params):
        ps = []
        for p in params:
            if p is None or not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                continue
            shape = [d for d in p.shape if d != 1]
            if len(shape) != 2 or min(shape) < 16 or shape[0] % 4 or shape[1] % 4:
                continue
            ps.append((p, shape[0], shape[1]))
        ps.sort(key=lambda e: e[0].data_ptr())
        self.device = ps[0][0].device if ps else None
        total = sum(n * k for _, n, k in ps)
        self.buffer = torch.full((max(total, 4),), 0.0, dtype=torch.float32, device=self.device) if ps else None
        self.starts, self.entries, desc, off, tiles = [], [], [], 0, 0
        for p, n, k in ps:
            view = self.buffer[off:off + n * k].view(k, n)               # W^T: (K, N) row-major
            self.starts.append(p.data_ptr())
            self.entries.append((p.data_ptr(), n, k, view, p))
            desc.append([p.data_ptr(), view.data_ptr(), n, k, tiles])
            tiles += ((n + 31) // 32) * ((k + 31) // 32)
            off += n * k
        self.total_tiles = tiles
        self.desc = torch.tensor(desc, dtype=torch.int64, device=self.device) if ps else None

    def __len__(self):
        return len(self.entries)

    def refresh(self):
        if not self.entries:
            return
        for ptr, _, _, _, p in self.entries:
            if p.data_ptr() != ptr:
                raise RuntimeError("a parameter moved after its W^T shadow was built (rebuild the shadow)")
        with torch.cuda.device(self.device):
            rc = _lib.lib().eda_transpose_batch_f32(self.desc.data_ptr(), len(self.entries), self.total_tiles,
                                                    torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "eda_transpose_batch_f32")

    def lookup(self, w):
        """w: (n, K) row-contiguous view of rows [r0, r0 + n) of a covered weight -> W^T[:, r0:r0+n] (K, n) with
        row stride N; None if w is not covered."""
        if not self.entries or w.dim() != 2 or w.stride(1) != 1 or w.device != self.device:
            return None
        ptr = w.data_ptr()
        i = bisect.bisect_right(self.starts, ptr) - 1
        if i < 0:
            return None
        base, n_all, k, view, _ = self.entries[i]
        if w.shape[1] != k or (w.shape[0] > 1 and w.stride(0) != k):
            return None
        byte_off = ptr - base
        if byte_off % (4 * k):
            return None
        r0 = byte_off // (4 * k)
        if r0 + w.shape[0] > n_all:
            return None
        return view[:, r0:r0 + w.shape[0]]
