"""Deferred weight gradients: every `dW = dY^T X` (+ bias gradient) of a backward pass in ONE
launch of csrc/wgrad.hip's grouped kernel.

A weight gradient is only consumed by the optimizer step, so nothing forces it to be computed
where autograd reaches the layer.  While a queue is active (``FlatParams.deferred_wgrad()``), the
pointwise-linear autograd nodes of the path (``nn_utils._LinearRows``, ``attention._ProjectedMHA``)
hand their (dY, X) pair to the queue instead of launching a GEMM, and return no gradient for the
weight / bias; ``flush()`` then writes all of them straight into the flat gradient buffer:
~200 launch-latency-bound GEMMs + column sums (5-6 ms of a 37 ms step on MI355X) become one
kernel with ~1700 independent 96x96 tiles in flight.  Modules applied at several places (the
contrastive projections run on every decoder layer) become several JOBS of one TARGET and are
summed inside the kernel's accumulators -- one writer per element, fixed order, deterministic.

Outside such a context nothing changes: gradients are computed immediately and returned to
autograd (that path is what the parity tests exercise against the reference goldens).
"""
import os

import numpy as np
import torch

from . import _lib
from .ext import _timed

active = None          # the queue that autograd nodes submit to, or None
_TILE = 96
_MAX_K = 1 << 15       # a tile walks all K rows serially; longer reductions keep their split-K paths


class WgradQueue:
    def __init__(self, locate):
        """locate(t) -> the gradient-buffer view matching parameter view `t`, or None."""
        self.locate = locate
        self._targets = {}       # dW data_ptr -> [dW, db, M, N, [(dy2, x2), ...]]
        self._ln = {}            # dgamma data_ptr -> (partial ws, rows, C, dgamma, dbeta, dbias or None)
        self._ring = []          # descriptor staging of EAGER flushes: [pinned host words, device words, copy-done event]
        self._ring_pos = 0
        self._capture_pool = []  # slots a CAPTURED flush takes for good (its graph replays the copy from them)
        self._captured = []

    _RING = 6                    # eager slots (never shrinks)
    _CAPTURE_SLOTS = 16          # captured stagings per reserve(): 4 per overlapped step capture (LN + targets, two ranges)
    _WORDS = 1 << 17             # 1 MiB of descriptors per slot (bench step: ~15 k words)

    def _new_slot(self, device):
        return [torch.empty((self._WORDS,), dtype=torch.int64, pin_memory=True),
                torch.empty((self._WORDS,), dtype=torch.int64, device=device), None]

    def reserve(self, device, captures=None):
        """Allocate the descriptor staging buffers.  Must happen outside stream capture (pinned
        allocation is not capturable); FlatParams does it at construction.  `captures`: slots to hold for
        captured flushes (each captured flush keeps one for the lifetime of its graph); call again before
        re-capturing many times -- the pool is topped up, never shrunk."""
        while len(self._ring) < self._RING:
            self._ring.append(self._new_slot(device))
        want = self._CAPTURE_SLOTS if captures is None else int(captures)
        while len(self._capture_pool) < want:
            self._capture_pool.append(self._new_slot(device))

    # ------------------------------------------------------------------ submit
    @staticmethod
    def _operand_ok(t, cols):
        return (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == cols
                and t.stride(1) == 1 and (t.shape[0] <= 1 or t.stride(0) % 4 == 0) and t.data_ptr() % 16 == 0)

    def plan(self, W, b, dy2, x2):
        """The (dW, db) gradient views this job would write, or None if it is not eligible."""
        if W.dim() != 2 or not W.is_contiguous():
            return None
        M, N = W.shape
        K = dy2.shape[0]
        if M % 4 or N % 4 or M < 16 or N < 16 or K < 1 or K > _MAX_K or x2.shape[0] != K:
            return None
        if not (self._operand_ok(dy2, M) and self._operand_ok(x2, N)):
            return None
        # the bf16 x 3 kernel addresses operand rows with 32-bit byte offsets from the operand's base
        if K > 1 and max(dy2.stride(0), x2.stride(0)) * 4 * K >= (1 << 32):
            return None
        gW = self.locate(W)
        if gW is None:
            return None
        gb = None
        if b is not None:
            gb = self.locate(b)
            if gb is None:
                return None
        return gW, gb

    def submit(self, W, b, dy2, x2, planned=None):
        """Queue dW(W) += dy2^T x2 (and db(b) += column sums of dy2).  False = not eligible: the
        caller computes the gradient itself."""
        planned = planned or self.plan(W, b, dy2, x2)
        if planned is None:
            return False
        gW, gb = planned
        key = gW.data_ptr()
        t = self._targets.get(key)
        if t is None:
            t = self._targets[key] = [gW, gb, W.shape[0], W.shape[1], []]
        else:
            if (t[2], t[3]) != tuple(W.shape) or (gb is None) != (t[1] is None):
                return False
        t[4].append((dy2, x2))
        return True

    def plan_ln(self, gamma, beta, y_bias):
        """Gradient views (dgamma, dbeta, dbias or None) a fused-LayerNorm site would write, or None
        if it is not eligible (a LayerNorm applied twice in one pass keeps the immediate path)."""
        gg, gb = self.locate(gamma), self.locate(beta)
        if gg is None or gb is None or gg.data_ptr() in self._ln:
            return None
        gy = None
        if y_bias is not None:
            gy = self.locate(y_bias)
            if gy is None or any(gy.data_ptr() == (t[1].data_ptr() if t[1] is not None else 0)
                                 for t in self._targets.values()):
                return None
        return gg, gb, gy

    def submit_ln(self, planned, ws, rows, C):
        """Queue the reduction of a fused LayerNorm backward's per-block partial sums (`rows` rows
        of 3*C floats in `ws`: d(gamma) | d(beta) | d(bias of the linear in front))."""
        gg, gb, gy = planned
        self._ln[gg.data_ptr()] = (ws, int(rows), int(C), gg, gb, gy)

    def __len__(self):
        return sum(len(t[4]) for t in self._targets.values()) + len(self._ln)

    def touched(self):
        """(data_ptr, bytes) of every gradient view (weights and biases) that flush() will write.  A view may be a ROW
        RANGE of a parameter's gradient (packed in-projections: q rows and K | V rows are queued independently)."""
        out = set()
        for gW, gb, *_ in self._targets.values():
            out.add((gW.data_ptr(), gW.numel() * 4))
            if gb is not None:
                out.add((gb.data_ptr(), gb.numel() * 4))
        for _, _, _, gg, gb, gy in self._ln.values():
            out.update((t.data_ptr(), t.numel() * 4) for t in (gg, gb, gy) if t is not None)
        return out

    # ------------------------------------------------------------------- flush
    def flush(self, accumulate=False, select=None):
        """Launch the grouped kernel for everything queued (stores, or adds with accumulate=True)
        and drop the references to the queued activations.  select(views) -> bool restricts the flush to the entries
        it accepts (views = the gradient views the entry writes); the others stay queued (FlatParams.flush_and_reduce
        flushes the buffer range by range so that a range's all-reduce runs underneath the next range's kernel)."""
        if select is None:
            lns = list(self._ln.values())
            self._ln = {}
            targets = list(self._targets.values())
            self._targets = {}
        else:
            ln_keys = [k for k, e in self._ln.items() if select([t for t in e[3:6] if t is not None])]
            lns = [self._ln.pop(k) for k in ln_keys]
            t_keys = [k for k, t in self._targets.items() if select([v for v in t[0:2] if v is not None])]
            targets = [self._targets.pop(k) for k in t_keys]
        if lns:
            self._flush_ln(lns)
        if not targets:
            return len(lns)
        dev = targets[0][0].device
        if os.environ.get("EDA_WGRAD_DUMP"):          # debugging aid: (M, N, [K per job]) of every target
            with open(os.environ["EDA_WGRAD_DUMP"], "w") as f:
                for _, _, M, N, jobs in targets:
                    f.write("%d %d %s\n" % (M, N, " ".join(str(dy.shape[0]) for dy, _ in jobs)))
        # cost of a tile = rows it walks; place whole targets on one XCD (workgroup id % 8), big first
        costs = [sum(dy.shape[0] for dy, _ in t[4]) for t in targets]
        order = sorted(range(len(targets)), key=lambda i: -costs[i] * ((targets[i][2] + _TILE - 1) // _TILE)
                       * ((targets[i][3] + _TILE - 1) // _TILE))
        lanes = [[] for _ in range(8)]
        load = [0] * 8
        for i in order:
            tm_n, tn_n = (targets[i][2] + _TILE - 1) // _TILE, (targets[i][3] + _TILE - 1) // _TILE
            x = min(range(8), key=load.__getitem__)
            load[x] += costs[i] * tm_n * tn_n
            for tm in range(tm_n):
                for tn in range(tn_n):
                    lanes[x].append((costs[i], i, tm, tn))
        depth = max(len(l) for l in lanes)
        ntasks = 8 * depth
        task_arr = np.zeros((ntasks, 4), dtype=np.int64)
        task_arr[:, 0] = -1                                   # padding tasks exit immediately
        for x, l in enumerate(lanes):
            l.sort(key=lambda e: -e[0])
            for slot, (_, i, tm, tn) in enumerate(l):
                task_arr[slot * 8 + x] = (i, tm, tn, 0)
        njobs = sum(len(t[4]) for t in targets)
        targ_arr = np.zeros((len(targets), 8), dtype=np.int64)
        job_arr = np.zeros((njobs, 8), dtype=np.int64)
        j = 0
        for i, (gW, gb, M, N, jobs) in enumerate(targets):
            targ_arr[i] = (gW.data_ptr(), gb.data_ptr() if gb is not None else 0, M, N, j, len(jobs),
                           1 if accumulate else 0, 0)
            for dy2, x2 in jobs:
                K = dy2.shape[0]
                job_arr[j] = (dy2.data_ptr(), dy2.stride(0) if K > 1 else M, x2.data_ptr(),
                              x2.stride(0) if K > 1 else N, K, 0, 0, 0)
                j += 1
        words = np.concatenate([task_arr.ravel(), targ_arr.ravel(), job_arr.ravel()])
        desc = self._stage(words, dev)
        base = desc.data_ptr()
        mmacs = sum(M * N * sum(dy.shape[0] for dy, _ in jobs) for _, _, M, N, jobs in targets) // 1000000
        with torch.cuda.device(dev), _timed("wgrad_grouped", (len(targets), njobs, ntasks, mmacs)):
            rc = _lib.lib().eda_wgrad_grouped_f32(base, ntasks, base + 8 * task_arr.size,
                                                  base + 8 * (task_arr.size + targ_arr.size),
                                                  torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "eda_wgrad_grouped_f32")
        # dY / X of the jobs and `desc` stay referenced by `targets` / this frame until here; the
        # caching allocator only re-uses their memory for work queued later on this stream.
        return njobs + len(lns)

    def _flush_ln(self, lns):
        dev = lns[0][3].device
        arr = np.zeros((len(lns), 8), dtype=np.int64)
        for i, (ws, rows, C, gg, gb, gy) in enumerate(lns):
            arr[i] = (ws.data_ptr(), rows, C, gg.data_ptr(), gb.data_ptr(), gy.data_ptr() if gy is not None else 0,
                      0, 0)
        desc = self._stage(arr.ravel(), dev)
        with torch.cuda.device(dev), _timed("ln_reduce_grouped", (len(lns),)):
            rc = _lib.lib().eda_ln_reduce_grouped_f32(desc.data_ptr(), len(lns), max(l[2] for l in lns),
                                                      torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "eda_ln_reduce_grouped_f32")

    def _stage(self, words, dev):
        """Copy descriptor words to the device through a pinned staging slot; returns the device view."""
        capturing = torch.cuda.is_current_stream_capturing()
        if not self._ring:
            if capturing:
                raise RuntimeError("WgradQueue.reserve() must be called before stream capture")
            self.reserve(dev)
        if words.size > self._WORDS:
            raise RuntimeError(f"wgrad queue: {words.size} descriptor words exceed the staging buffer")
        # Eager flushes rotate over a ring of pinned staging slots: a slot is rewritten only after the copy that last
        # read it has completed.  A captured graph keeps replaying the copy from the slot it was captured with, so a
        # captured flush takes its slot from a separate pool, for good (taking it from the eager ring left
        # that ring empty after two overlapped captures).
        if capturing:
            if not self._capture_pool:
                raise RuntimeError(
                    "wgrad queue: staging slots for captured flushes exhausted (%d graphs hold one each); call "
                    "WgradQueue.reserve(device, captures=N) outside stream capture before capturing again"
                    % len(self._captured))
            slot = self._capture_pool.pop()
        else:
            slot = self._ring[self._ring_pos]
            self._ring_pos = (self._ring_pos + 1) % len(self._ring)
            if slot[2] is not None:
                slot[2].synchronize()
        host, desc = slot[0], slot[1]
        host.numpy()[:words.size] = words
        desc[:words.size].copy_(host[:words.size], non_blocking=True)
        if capturing:
            self._captured.append(slot)                                   # keep alive for the replays
        else:
            slot[2] = torch.cuda.Event()
            slot[2].record()
        return desc[:words.size]
