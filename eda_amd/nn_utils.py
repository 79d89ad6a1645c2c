"""Pointwise (kernel-size-1) convolutions as plain GEMMs.

Every convolution on the hot path is 1x1 (SharedMLP, the conv1d heads, the
positional embeddings).  ``nn.Conv1d/Conv2d`` would send them through MIOpen's
solver search (tens of seconds of warm-up "find" kernels and NCHW direct-conv
solvers); a 1x1 convolution is just W (Cout,Cin) @ X (B,Cin,M), so these
subclasses keep the parameter names and shapes of the torch modules (checkpoint
contract) and run one batched GEMM instead.
"""
import torch
from torch import nn
import torch.nn.functional as F
from torch.autograd import Function

_colsum_counters = {}


def _ticket_counters(dev, n):
    """The array of self-resetting ticket counters of csrc/colsum.hip / wgrad.hip, one per (device, stream): two
    streams of a device may run these kernels concurrently (a side-stream branch's backward next to the main one)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    cnt = _colsum_counters.get(key)
    if cnt is None or cnt.numel() < n:
        # (a fill kernel, not torch.zeros: a memset node inside a captured HIP graph is a hazard on ROCm 7.2)
        cnt = _colsum_counters[key] = torch.full((max(4096, n),), 0, dtype=torch.int32, device=dev)
    return cnt


def colsum(x2, out=None):
    """out[c] = sum_r x2[r, c] for a 2-D fp32 GPU matrix: ONE launch of csrc/colsum.hip (the
    bias gradient of a pointwise linear layer).  The ticket-counter array is per (device, stream)."""
    from . import _lib
    from .ext import _timed
    assert x2.dim() == 2 and x2.is_cuda and x2.dtype == torch.float32
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    R, C = x2.shape
    dev = x2.device
    if out is None:
        out = torch.empty((C,), dtype=torch.float32, device=dev)
    cnt = _ticket_counters(dev, (C + 63) // 64)
    L = _lib.lib()
    ws_bytes = L.eda_colsum_workspace_bytes(R, C)
    ws = torch.empty((max(ws_bytes, 4),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev), _timed("colsum", (R, C)):
        rc = L.eda_colsum_f32(x2.data_ptr(), R, C, x2.stride(0) if R > 1 else C, out.data_ptr(), ws.data_ptr(),
                              ws_bytes, cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_colsum_f32")
    return out


def wgrad(dy2, x2, dW=None, db=None, want_db=True):
    """(dW, db) = (dy2^T x2, column sums of dy2) for 2-D fp32 GPU matrices dy2 (K,M), x2 (K,N):
    the split-K MFMA kernel of csrc/wgrad.hip when the shapes allow 16-byte rows, else the
    library GEMM + colsum.  dW / db may be preallocated (contiguous) destinations."""
    from . import _lib
    from .ext import _timed
    K, M = dy2.shape
    N = x2.shape[1]
    dev = dy2.device
    if dy2.stride(1) != 1:
        dy2 = dy2.contiguous()
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    if dW is None:
        dW = torch.empty((M, N), dtype=torch.float32, device=dev)
    if want_db and db is None:
        db = torch.empty((M,), dtype=torch.float32, device=dev)
    if (M <= 4 and N % 4 == 0 and K > 0 and (K == 1 or x2.stride(0) % 4 == 0) and x2.data_ptr() % 16 == 0
            and dW.is_contiguous()):
        # 1-4 output channels: dW = weighted column sums of x (csrc/colsum.hip), db for free
        L = _lib.lib()
        ws_bytes = L.eda_wcolsum_workspace_bytes(K, N, M)
        ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
        cnt = _ticket_counters(dev, (N + 63) // 64)
        with torch.cuda.device(dev), _timed("wcolsum", (K, M, N)):
            rc = L.eda_wcolsum_f32(x2.data_ptr(), K, N, x2.stride(0) if K > 1 else N, dy2.data_ptr(),
                                   dy2.stride(0) if K > 1 else M, M, dW.data_ptr(),
                                   db.data_ptr() if want_db else None, ws.data_ptr(), ws_bytes,
                                   cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "eda_wcolsum_f32")
        return dW, (db if want_db else None)
    ok = (M % 4 == 0 and N % 4 == 0 and K > 0 and (K == 1 or (dy2.stride(0) % 4 == 0 and x2.stride(0) % 4 == 0))
          and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0 and dW.data_ptr() % 16 == 0
          and dW.is_contiguous() and 32 <= N <= 288 and 32 <= M <= 288)
    # (measured, tools/bench_dw.py: for outputs up to 288x288 the split-K kernel + free db beats
    # library GEMM + colsum by 5 us at K=2048 and 20 us at K=8192; for the packed 576/864-row
    # in-projections the library's 77 TFLOP/s wins, so those keep torch.mm + colsum)
    if not ok:
        torch.mm(dy2.t(), x2, out=dW)
        if want_db:
            colsum(dy2, out=db)
        return dW, (db if want_db else None)
    L = _lib.lib()
    ws_bytes = L.eda_wgrad_workspace_bytes(K, M, N)
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev), _timed("wgrad", (K, M, N)):
        rc = L.eda_wgrad_f32(dy2.data_ptr(), dy2.stride(0) if K > 1 else M, x2.data_ptr(),
                             x2.stride(0) if K > 1 else N, K, M, N, dW.data_ptr(),
                             db.data_ptr() if want_db else None, ws.data_ptr(), ws_bytes,
                             torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_wgrad_f32")
    return dW, (db if want_db else None)


class _LinearRows(Function):
    """y = x W^T + b (optionally ReLU) over the last dimension on the repo's own fp32-MFMA row GEMMs
    (csrc/gemm.hip): forward and dX are eda_linear_fwd/dgrad_f32, bias and ReLU ride in the
    forward's epilogue; dW (and the bias gradient, for free) come from wgrad() / the deferred queue."""

    @staticmethod
    def forward(ctx, x, W, b, relu=False):
        from . import gemm
        x2 = x.reshape(-1, x.shape[-1])
        y = gemm.linear_fwd(x2, W, b, relu)
        ctx.save_for_backward(x2, W, b, y if relu else None)
        ctx.xshape = x.shape
        ctx.has_bias = b is not None
        return y.view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        from . import gemm, wgrad_queue
        x2, W, b, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, W.shape[0])
        if y is not None:
            dy2 = torch.ops.aten.threshold_backward(dy2, y, 0.0)      # ReLU: dy where y > 0
        dx = gemm.linear_dgrad(dy2, W).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        dW = db = None
        q = wgrad_queue.active
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if (q is not None and ctx.needs_input_grad[1] and (want_db or not ctx.has_bias)
                and q.submit(W, b if want_db else None, dy2, x2)):
            return dx, None, None, None    # written into the gradient buffer by the queue's flush
        if ctx.needs_input_grad[1]:
            dW, db = wgrad(dy2, x2, want_db=ctx.has_bias and ctx.needs_input_grad[2])
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy2)
        return dx, dW, db, None


class _MLPChain(Function):
    """Linear (-> ReLU -> Dropout) -> ... -> Linear as ONE autograd node on the repo's GEMMs: the activations ride in
    the GEMM epilogues forward (bias, ReLU, Dropout: csrc/gemm.hip eda_linear_ex_f32) and backward (the ReLU / Dropout
    backward of layer l is the gate of layer l+1's input-gradient GEMM), so a two-layer FFN costs 2 + 2 launches plus
    its queued weight gradients, where Linear / ReLU / Dropout modules under autograd cost 3 + 4 (+ two gradient
    kernels).  cfg: per layer (relu, p, salt, has_bias); tensors: x, then W_l, b_l (b_l may be None) per layer.
    The LAST layer may leave its bias to the caller (has_bias False with b None)."""

    @staticmethod
    def forward(ctx, cfg, seed, x, *wb):
        from . import gemm
        x2 = x.reshape(-1, x.shape[-1])
        acts = [x2]
        for l, (relu, p, salt, _) in enumerate(cfg):
            W, b = wb[2 * l], wb[2 * l + 1]
            acts.append(gemm.linear_ex(acts[-1], W, b, relu, drop=(p, seed, salt) if p > 0 else None))
        ctx.cfg, ctx.xshape = cfg, x.shape
        ctx.save_for_backward(*acts[:-1], *[t for t in wb if t is not None])
        ctx.none_b = [wb[2 * l + 1] is None for l in range(len(cfg))]
        return acts[-1].view(*x.shape[:-1], wb[2 * (len(cfg) - 1)].shape[0])

    @staticmethod
    def backward(ctx, dy):
        from . import gemm, wgrad_queue
        cfg = ctx.cfg
        n = len(cfg)
        acts = ctx.saved_tensors[:n]
        rest = list(ctx.saved_tensors[n:])
        Ws, bs = [], []
        for l in range(n):
            Ws.append(rest.pop(0))
            bs.append(None if ctx.none_b[l] else rest.pop(0))
        grads = [None] * (2 * n)
        q = wgrad_queue.active
        dz = dy.reshape(-1, Ws[-1].shape[0])
        if not (dz.stride(1) == 1 and (dz.shape[0] <= 1 or dz.stride(0) >= dz.shape[1])):
            dz = dz.contiguous()
        dx = None
        for l in range(n - 1, -1, -1):
            W, b, inp = Ws[l], bs[l], acts[l]
            need_w, need_b = ctx.needs_input_grad[3 + 2 * l], b is not None and ctx.needs_input_grad[4 + 2 * l]
            if not (q is not None and need_w and (need_b or b is None) and q.submit(W, b if need_b else None, dz, inp)):
                if need_w:
                    grads[2 * l], db = wgrad(dz, inp, want_db=need_b)
                    if need_b:
                        grads[2 * l + 1] = db
                elif need_b:
                    grads[2 * l + 1] = colsum(dz)
            if l > 0:
                relu, p, _, _ = cfg[l - 1]
                if relu:          # inp = dropout(relu(.)): positive exactly where the gradient passes
                    dz = gemm.linear_dgrad_gated(dz, W, inp, 1.0 / (1.0 - p) if p > 0 else 1.0)
                else:
                    assert p == 0, "Dropout without ReLU in front of it is not a case of the model"
                    dz = gemm.linear_dgrad(dz, W)
            elif ctx.needs_input_grad[2]:
                dx = gemm.linear_dgrad(dz, W).view(ctx.xshape)
        return (None, None, dx, *grads)


def mlp_chain(x, layers, training, seed=None):
    """layers: [(weight, bias or None, relu, p, salt)], applied in order (Dropout only in training).  GPU fp32: one
    autograd node (_MLPChain); otherwise the torch composition (same mathematics, torch's RNG)."""
    if x.is_cuda and x.dtype == torch.float32 and x.numel() > 0:
        cfg = tuple((bool(r), float(p) if training else 0.0, int(s), b is not None) for _, b, r, p, s in layers)
        wb = []
        for W, b, *_ in layers:
            wb += [W, b]
        if seed is None and any(c[1] > 0 for c in cfg):
            from .attention import dropout_state
            seed = dropout_state(x.device)
        return _MLPChain.apply(cfg, seed, x, *wb)
    for W, b, relu, p, _ in layers:
        x = F.linear(x, W, b)
        if relu:
            x = F.relu(x)
        if p > 0:
            x = F.dropout(x, p, training)
    return x


def linear_rows(x, weight, bias, relu=False):
    """F.linear(x, weight, bias) (+ ReLU); on the GPU (fp32) the repo's own MFMA GEMMs."""
    if x.is_cuda and x.dtype == torch.float32 and x.numel() > 0:
        return _LinearRows.apply(x, weight, bias, relu)
    y = F.linear(x, weight, bias)
    return F.relu(y) if relu else y


class Linear(nn.Linear):
    """nn.Linear (same parameters / state_dict) routed through linear_rows."""

    def forward(self, x):
        return linear_rows(x, self.weight, self.bias)


class Conv1dK1(nn.Conv1d):
    def __init__(self, cin, cout, kernel_size=1, bias=True):
        assert kernel_size == 1
        super().__init__(cin, cout, 1, bias=bias)

    def rows(self, x):
        """Channels-last form: x (..., Cin) -> (..., Cout), one row-major GEMM, no transposes."""
        return linear_rows(x, self.weight.squeeze(-1), self.bias)

    def forward(self, x):                                  # (B, Cin, M)
        y = torch.matmul(self.weight.squeeze(-1), x)
        return y if self.bias is None else y + self.bias[:, None]


class Conv2dK1(nn.Conv2d):
    def __init__(self, cin, cout, kernel_size=(1, 1), bias=True):
        assert tuple(kernel_size) == (1, 1)
        super().__init__(cin, cout, (1, 1), bias=bias)

    def forward(self, x):                                  # (B, Cin, H, W)
        B, C, H, W = x.shape
        y = torch.matmul(self.weight.view(self.out_channels, C), x.reshape(B, C, H * W))
        if self.bias is not None:
            y = y + self.bias[:, None]
        return y.view(B, self.out_channels, H, W)


_deferred_counters = None


class deferred_bn_counters:
    """Inside this context the `num_batches_tracked += 1` of every BatchNorm that runs in
    training mode is collected and applied by ONE multi-tensor launch at exit instead of ~70
    single-element kernels (the values afterwards are the same)."""

    def __enter__(self):
        global _deferred_counters
        self.prev, _deferred_counters = _deferred_counters, []
        return self

    def __exit__(self, *exc):
        global _deferred_counters
        mine, _deferred_counters = _deferred_counters, self.prev
        by_count = {}
        for t, n in {id(t): (t, sum(u is t for u in mine)) for t in mine}.values():
            by_count.setdefault(n, []).append(t)        # a module that ran twice counts twice
        for n, ts in by_count.items():
            torch._foreach_add_(ts, n)
        return False


def bump_batches_tracked(bn):
    if _deferred_counters is not None:
        _deferred_counters.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)


def dropout_salt(dropout):
    """The call-site salt of a Dropout module whose masks the library's kernels draw (fused into BN+ReLU): assigned on first
    use and kept ON THE MODULE -- a deep copy of a model draws the same masks as the original for the same counter (until
    round 5 a process-wide table keyed by id(module): two copies of one model trained differently, and a freed module's id
    could hand its salt to an unrelated one)."""
    salt = getattr(dropout, "_eda_salt", None)
    if salt is None:
        from .fused_ln import new_salt_base
        salt = dropout._eda_salt = new_salt_base() + 7
    return salt
_bn_drop_rows = None


def bn_relu_rows(bn, z, dropout=None):
    """relu(BatchNorm1d/2d `bn`(z)) for channels-last rows z (R, C) on the GPU: the fused
    HIP kernel of csrc/sa_cl.hip (batch statistics + running-stat update in training).
    `dropout`: an nn.Dropout to apply behind the ReLU (the heads' Conv-BN-ReLU-Dropout); in
    training it is folded into the same kernel when the row count allows, with this repo's
    counter-based mask (deviations) instead of torch's Philox stream."""
    global _bn_drop_rows
    from . import sa_ops, sync_bn
    if sync_bn.diverts():
        out = sync_bn.bn_relu(bn, z)
        if bn.training and bn.track_running_stats:
            bump_batches_tracked(bn)
        return dropout(out) if dropout is not None else out
    p, salt = 0.0, 0
    if dropout is not None and dropout.training and dropout.p > 0:
        if _bn_drop_rows is None:
            from . import _lib
            _bn_drop_rows = _lib.lib().eda_bn_relu_dropout_max_rows()
        if z.shape[0] <= _bn_drop_rows:
            p = float(dropout.p)
            salt = dropout_salt(dropout)
    out = sa_ops.BNReLUCL.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                bn.momentum, bn.training, 1, p, salt)
    if bn.training and bn.track_running_stats:
        bump_batches_tracked(bn)
    if dropout is not None and p == 0.0:
        out = dropout(out)
    return out


def rows_ok(x, *channels):
    """The rows fast path needs the HIP library (GPU tensors) and channel counts % 4 == 0."""
    return x.is_cuda and all(c % 4 == 0 for c in channels)


class _L2NormalizeRows(torch.autograd.Function):
    """F.normalize(x, p=2, dim=-1) as one HIP launch forward and one backward (csrc/ln.hip)."""

    @staticmethod
    def forward(ctx, x, eps):
        from . import _lib
        from .ext import _timed
        xc = x.contiguous()
        C = xc.shape[-1]
        R = xc.numel() // C
        y = torch.empty_like(xc)
        norm = torch.empty((R,), dtype=torch.float32, device=xc.device)
        with torch.cuda.device(xc.device), _timed("l2norm_fwd", (R, C)):
            rc = _lib.lib().eda_l2norm_rows_fwd_f32(xc.data_ptr(), R, C, float(eps), y.data_ptr(), norm.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "eda_l2norm_rows_fwd_f32")
        ctx.save_for_backward(y, norm)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        from .ext import _timed
        y, norm = ctx.saved_tensors
        dyc = dy.contiguous()
        C = y.shape[-1]
        R = y.numel() // C
        dx = torch.empty_like(y)
        with torch.cuda.device(y.device), _timed("l2norm_bwd", (R, C)):
            rc = _lib.lib().eda_l2norm_rows_bwd_f32(dyc.data_ptr(), y.data_ptr(), norm.data_ptr(), R, C, ctx.eps,
                                                    dx.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "eda_l2norm_rows_bwd_f32")
        return dx, None


class _EmbeddingRows(Function):
    """F.embedding(ids, weight) whose weight gradient is ONE index_add_ into a kernel-filled buffer: torch's
    embedding_dense_backward takes 93 us for the 1056 x 768 rows of the detected-box class embeddings
    (bdetr.py:150-156 of the reference keeps that table trainable: `module.requires_grad = False` sets an attribute of the
    Module, not of its weight)."""

    @staticmethod
    def forward(ctx, weight, ids):
        ctx.save_for_backward(ids)
        ctx.wshape = weight.shape
        return weight.index_select(0, ids.reshape(-1)).view(*ids.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        from . import _lib, deterministic
        if deterministic.enabled() and g.dtype == torch.float32:
            # ordered per-row sums (csrc/scatter_det.hip) instead of index_add_'s fp32 atomics
            g2 = g.reshape(-1, g.shape[-1]).contiguous()
            dW = torch.empty(ctx.wshape, dtype=g.dtype, device=g.device)
            i32 = ids.reshape(-1).to(torch.int32)
            with torch.cuda.device(g.device):
                rc = _lib.lib().eda_index_add_rows_ordered_f32(g2.data_ptr(), i32.data_ptr(), g2.shape[0], g2.shape[1],
                                                               ctx.wshape[0], dW.data_ptr(), torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "eda_index_add_rows_ordered_f32")
            return dW, None
        dW = torch.full(ctx.wshape, 0.0, dtype=g.dtype, device=g.device)        # (a fill kernel, not a memset node)
        dW.index_add_(0, ids.reshape(-1), g.reshape(-1, g.shape[-1]))
        return dW, None


def embedding_rows(emb, ids):
    """emb(ids) for an nn.Embedding without padding_idx / max_norm (GPU: _EmbeddingRows)."""
    if ids.is_cuda and emb.padding_idx is None and emb.max_norm is None and not emb.sparse:
        return _EmbeddingRows.apply(emb.weight, ids)
    return emb(ids)


def l2_normalize(x, eps=1e-12):
    """torch.nn.functional.normalize(x, p=2, dim=-1): fused on the GPU (fp32, rows of <= 1024), the
    torch composition elsewhere (CPU tensors of the host-logic tests)."""
    if x.is_cuda and x.dtype == torch.float32 and 0 < x.shape[-1] <= 1024:
        return _L2NormalizeRows.apply(x, eps)
    return torch.nn.functional.normalize(x, p=2, dim=-1, eps=eps)


class _FanOut(Function):
    """n aliases of one tensor, one per consumer: the gradients the consumers send back are summed in ONE launch
    (include/eda_hip.h eda_add_n_f32) instead of n - 1 accumulation launches of the autograd engine."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        g0 = gs[0]
        ok = (g0.is_cuda and g0.dtype == torch.float32 and g0.numel() % 4 == 0 and len(gs) <= 8
              and all(g.shape == g0.shape and g.dtype == torch.float32 for g in gs))
        if not ok:
            out = gs[0] + gs[1]
            for g in gs[2:]:
                out = out + g
            return out, None
        gs = [g if (g.is_contiguous() and g.data_ptr() % 16 == 0) else g.contiguous() for g in gs]
        out = torch.empty_like(gs[0])
        import ctypes
        from . import _lib
        from .ext import _timed
        arr = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
        with torch.cuda.device(g0.device), _timed("add_n", (len(gs), g0.numel())):
            rc = _lib.lib().eda_add_n_f32(arr, len(gs), g0.numel(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "eda_add_n_f32")
        return out, None


def fan_out(x, n):
    """`n` aliases of `x` for n consumers (training on the GPU: their gradients are then added in one launch); a plain
    tuple of the same tensor where that buys nothing (no gradient, CPU, n < 3: two consumers cost the engine one add
    either way)."""
    if n < 3 or not (torch.is_grad_enabled() and x.requires_grad and x.is_cuda):
        return (x,) * n
    return _FanOut.apply(x, n)
