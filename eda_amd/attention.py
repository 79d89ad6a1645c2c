"""Multi-head attention as EDA uses it (d_model 288, 8 heads x 36, key-padding
masks, attention-probability dropout 0.1).

``MultiheadAttention`` keeps the parameter names and initialisation of
``torch.nn.MultiheadAttention`` (in_proj_weight (3d,d), in_proj_bias, out_proj.*)
-- the reference builds 39 of those (models/encoder_decoder_layers.py:47,62,69,133,
298,306,314,319) and checkpoints must load -- but works batch-first: no
(B,N,F)<->(N,B,F) transposes, one packed projection GEMM when q/k/v share their
input, and the QK^T-softmax-dropout-PV core is ONE fused HIP kernel
(csrc/mha2.hip, fp32 MFMA) that reads the heads in place from the (B,L,288)
projection outputs and never materialises the (B*8,Lq,Lk) probabilities; its
backward is one more kernel (dQ, dK, dV in one pass) that recomputes them.

``attention_core`` has no CPU path (the HIP library is the product); the torch
restatement used by CPU-side tests lives in oracle/attention_ref.py.
"""
import itertools
import os

import torch
from torch import nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib, gemm
from .ext import _timed
from .nn_utils import colsum, linear_rows, wgrad

_DTYPES = {"f32": 0, "bf16": 1, "f16": 2}
_compute_dtype = 0       # arithmetic of the QK^T / PV contractions (include/eda_hip.h EDA_DTYPE_*)


def set_compute_dtype(name):
    """"f32" (default: the parity path, fp32 MFMA), "bf16" or "f16" (16-bit MFMA contractions with fp32
    accumulation and softmax, the same kernels of csrc/mha2.hip -- BASELINE.json configs[2] / configs[4]).  Tensors stay fp32."""
    global _compute_dtype
    _compute_dtype = _DTYPES[name]


def compute_dtype():
    return {v: k for k, v in _DTYPES.items()}[_compute_dtype]


_dropout_state = {}      # device -> int64 counter tensor read by the kernels
_salt_counter = itertools.count(1)


def _initial_dropout_counter():
    """torch.initial_seed() mixed with the data-parallel rank (splitmix64 finaliser), kept below 2^62: ranks draw
    different masks, torch.manual_seed() selects the stream, and a run is reproducible for a given seed."""
    import os
    import torch.distributed as dist
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else int(os.environ.get("RANK", "0"))
    x = (torch.initial_seed() + 0x9E3779B97F4A7C15 * (rank + 1)) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return (x ^ (x >> 31)) & ((1 << 62) - 1)


def dropout_state(device):
    """The device-side counter every dropout kernel of the library hashes (attention, residual LayerNorm, BN+ReLU
    heads, FFN epilogues) together with a per-call-site salt.  Created on first use from torch's seed and the rank;
    advance_dropout_state() bumps it once per step; get/set_dropout_counter() make it checkpointable
    (eda_amd/checkpoint.py stores it)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if device not in _dropout_state:
        _dropout_state[device] = torch.full((1,), _initial_dropout_counter(), dtype=torch.int64, device=device)
    return _dropout_state[device]


def advance_dropout_state(device):
    """Bump the device-side dropout counter (once per training step; capturable in a
    HIP graph, so every replay draws new masks)."""
    dropout_state(device).add_(1)


def get_dropout_counter(device):
    """Current value of the dropout counter (host int; synchronises)."""
    return int(dropout_state(device).item())


def set_dropout_counter(device, value):
    """Continue a saved dropout stream (in place: graphs that captured the counter tensor keep working)."""
    dropout_state(device).fill_(int(value) & ((1 << 62) - 1))


def _rows(t):
    """(B,L,D) view with unit last stride and 16-byte aligned rows, else a copy."""
    if t.stride(-1) != 1 or t.stride(0) % 4 or t.stride(1) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
    return t


_fwd_ws_cache = {}


def _fwd_workspace(dev, B, H, Lq, Lk):
    """Scratch of the key-split forward (eda_mha_fwd_ws: short query sets against the 1024 point keys): per-split
    (O, m, l) partials behind a block of ticket words that must be ZERO before the first call and that every call leaves
    zero again -- so one persistent buffer per (device, stream, size) serves every site that runs on that stream."""
    n = int(_lib.lib().eda_mha_fwd_workspace_bytes(B, H, Lq, Lk))
    if n == 0:
        return None
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, n)
    ws = _fwd_ws_cache.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            # (a buffer born inside a capture lives in that graph's pool and is zeroed only when that graph replays, yet this
            # cache would hand it to eager calls and other graphs: ADVICE r05)
            raise RuntimeError("eda_amd.attention: no key-split workspace for this (device, stream, size) yet -- run the "
                               "step once eagerly on the stream before capturing it in a HIP graph")
        ws = _fwd_ws_cache[key] = torch.zeros((n + 3) // 4, dtype=torch.int32, device=dev)
    return ws


def reset_workspaces():
    """Re-zero every key-split workspace handed out so far (their ticket words must be zero between launches; an aborted
    launch can leave a count behind).  Synchronises."""
    torch.cuda.synchronize()
    for ws in list(_fwd_ws_cache.values()) + list(_bwd_tk_cache.values()):
        ws.zero_()
    torch.cuda.synchronize()


_bwd_tk_cache = {}
_bwd_tk_retired = []


def _bwd_tickets(dev, nbytes):
    """Arrival tickets of the split backward (include/eda_hip.h: eda_mha_bwd_tk): one persistent ZERO buffer per (device,
    stream) -- every launch leaves it zero -- created outside any capture (same reason as _fwd_workspace)."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    tk = _bwd_tk_cache.get(key)
    if tk is None or tk.numel() * 4 < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("eda_amd.attention: no backward ticket buffer for this (device, stream) yet -- run the step "
                               "once eagerly on the stream before capturing it in a HIP graph")
        if tk is not None:
            _bwd_tk_retired.append(tk)          # (a captured graph may still hold the smaller buffer's address: never freed)
        tk = _bwd_tk_cache[key] = torch.zeros(max(16384, (nbytes + 3) // 4), dtype=torch.int32, device=dev)
    return tk


def _mha_bwd_call(q, k, v, m8, B, num_heads, Lq, Lk, hd, p_drop, seed, salt, out, lse, dout, dq, dk, dv, dtype_code):
    dev = q.device
    L = _lib.lib()
    with torch.cuda.device(dev), _timed('mha_bwd', (B, num_heads, Lq, Lk)):
        ws_bytes = L.eda_mha_bwd_workspace_bytes(B, num_heads, Lq, Lk)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev) if ws_bytes else None
        tk_bytes = L.eda_mha_bwd_ticket_bytes(B, num_heads, Lq, Lk)
        tk = _bwd_tickets(dev, tk_bytes) if tk_bytes else None
        rc = L.eda_mha_bwd_tk(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), q.stride(1), k.stride(0),
            k.stride(1), v.stride(0), v.stride(1), m8.data_ptr() if m8 is not None else None,
            B, num_heads, Lq, Lk, hd, hd ** -0.5, p_drop,
            seed.data_ptr() if seed is not None else None, salt, out.data_ptr(), lse.data_ptr(),
            dout.data_ptr(), dout.stride(0), dout.stride(1), None, dq.data_ptr(),
            dk.data_ptr(), dv.data_ptr(), dq.stride(0), dq.stride(1), dk.stride(0), dk.stride(1),
            dv.stride(0), dv.stride(1), ws.data_ptr() if ws is not None else None, ws_bytes,
            tk.data_ptr() if tk is not None else None, tk.numel() * 4 if tk is not None else 0,
            dtype_code, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_mha_bwd_tk")


def _mha_fwd_call(q, k, v, m8, B, num_heads, Lq, Lk, hd, p_drop, salt, out, lse):
    dev = q.device
    seed = dropout_state(dev) if p_drop > 0 else None
    with torch.cuda.device(dev), _timed('mha_fwd', (B, num_heads, Lq, Lk)):
        ws = _fwd_workspace(dev, B, num_heads, Lq, Lk) if _compute_dtype == 0 else None
        rc = _lib.lib().eda_mha_fwd_ws(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), q.stride(1), k.stride(0),
            k.stride(1), v.stride(0), v.stride(1), m8.data_ptr() if m8 is not None else None,
            B, num_heads, Lq, Lk, hd, hd ** -0.5, float(p_drop),
            seed.data_ptr() if seed is not None else None, int(salt), out.data_ptr(),
            lse.data_ptr(), _compute_dtype, ws.data_ptr() if ws is not None else None,
            ws.numel() * 4 if ws is not None else 0, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_mha_fwd_ws")


class _FusedMHA(Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, num_heads, p_drop, salt):
        if not q.is_cuda:
            raise RuntimeError("CPU not supported: attention_core runs on the HIP library only")
        q, k, v = _rows(q), _rows(k), _rows(v)
        B, Lq, D = q.shape
        Lk = k.shape[1]
        hd = D // num_heads
        out = torch.empty((B, Lq, D), dtype=torch.float32, device=q.device)
        lse = torch.empty((B, num_heads, Lq), dtype=torch.float32, device=q.device)
        m8 = None
        if mask is not None:
            m8 = mask.contiguous().view(torch.uint8)
        _mha_fwd_call(q, k, v, m8, B, num_heads, Lq, Lk, hd, p_drop, salt, out, lse)
        ctx.dtype_code = _compute_dtype
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.mask8 = m8
        ctx.cfg = (num_heads, float(p_drop), int(salt))
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        num_heads, p_drop, salt = ctx.cfg
        dout = _rows(dout)
        B, Lq, D = q.shape
        Lk = k.shape[1]
        hd = D // num_heads
        dq = torch.empty((B, Lq, D), dtype=torch.float32, device=q.device)
        dk = torch.empty((B, Lk, D), dtype=torch.float32, device=q.device)
        dv = torch.empty((B, Lk, D), dtype=torch.float32, device=q.device)
        m8 = ctx.mask8
        seed = dropout_state(q.device) if p_drop > 0 else None
        _mha_bwd_call(q, k, v, m8, B, num_heads, Lq, Lk, hd, p_drop, seed, salt, out, lse, dout, dq, dk, dv, ctx.dtype_code)
        return dq, dk, dv, None, None, None, None


def _qproj_fused_ok(d, num_heads, Lk, *tensors):
    """The fused [q-projection | attention core] launch (include/eda_hip.h eda_mha_qproj_fwd) takes this site.
    Measured alone (tools/time_qproj_site.py, graph replay, B = 8): 256 queries x 80 / 132 keys 15.4 / 17.0 us against 17.1 /
    18.1 us for the two launches (one workgroup of 64 queries x 1 head per CU, one round); 1024 queries: 62 / 71 us against
    36 / 42 us (four rounds of 106 KB workgroups).  Measured IN THE STEP, EDA_MHA_QPROJ=auto (the launch on the twelve one-round
    sites: rounds 4-6's default) against =0, four evidence runs: 459.7 / 461.1, 461.4 / 462.5, 463.6 / 467.2, 460.0 / 460.3
    scenes/s -- off is never behind and at most a few tenths of a percent ahead (run-to-run spread on one box: ~1.5 scenes/s), so
    the 1-2 us it saves alone do not arrive in the step.  Why not was NOT investigated.  The default is OFF since the end of
    round 6 (one kernel variant less on the default path); =auto takes it while it is one round of the chip, =1 forces it for
    every supported shape (the 1024-query sites included: 17.49 against 17.30 ms per step)."""
    import os
    mode = os.environ.get("EDA_MHA_QPROJ", "0")
    if mode == "0" or d % num_heads or not all(t.is_cuda and t.dtype == torch.float32 for t in tensors):
        return False
    if not _lib.lib().eda_mha_qproj_supported(num_heads, d // num_heads, int(Lk)):
        return False
    x = tensors[0]
    one_round = x.shape[0] * num_heads * ((x.shape[1] + 63) // 64) <= 256
    return mode == "1" or one_round


def _qproj_core_fwd(x, Wq, bq, k, v, m8, num_heads, p_drop, salt):
    """q = x Wq^T + bq and softmax(q k^T / sqrt(hd) + mask) v in ONE launch.  Returns (q, out, lse)."""
    x = _rows(x)
    B, Lq, d = x.shape
    Lk = k.shape[1]
    hd = d // num_heads
    dev = x.device
    q = torch.empty((B, Lq, d), dtype=torch.float32, device=dev)
    out = torch.empty((B, Lq, d), dtype=torch.float32, device=dev)
    lse = torch.empty((B, num_heads, Lq), dtype=torch.float32, device=dev)
    Wq = gemm._rows2d(Wq)
    seed = dropout_state(dev) if p_drop > 0 else None
    with torch.cuda.device(dev), _timed('mha_qproj_fwd', (B, num_heads, Lq, Lk)):
        rc = _lib.lib().eda_mha_qproj_fwd(
            x.data_ptr(), x.stride(0), x.stride(1), Wq.data_ptr(), gemm._ld(Wq), bq.data_ptr() if bq is not None else None,
            k.data_ptr(), v.data_ptr(), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
            m8.data_ptr() if m8 is not None else None, B, num_heads, Lq, Lk, hd, hd ** -0.5, float(p_drop),
            seed.data_ptr() if seed is not None else None, int(salt), q.data_ptr(), q.stride(0), q.stride(1),
            out.data_ptr(), lse.data_ptr(), _compute_dtype, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_mha_qproj_fwd")
    return q, out, lse


class ResidualLink:
    """One post-norm block's residual gradient, handed from the residual LayerNorm's backward to the attention node's
    instead of to autograd: the block's input x feeds the attention branch AND the residual sum
    (models/encoder_decoder_layers.py:231-245, :87-93), so autograd would add the two gradients in a launch of its own;
    with a link the LayerNorm node returns no gradient for x and leaves it here, and the attention node's input-gradient
    product for x adds it in its epilogue (gemm.linear_dgrad(addend=...)) -- the same terms (where a third gradient meets them,
    in another association: one fp32 rounding), one launch less per block.  Armed by MultiheadAttention.forward only
    when that product will run (x requires grad, is
    one of the node's inputs); the LayerNorm node (which always runs first in a backward pass: it consumes the attention
    node's output) checks `armed`."""
    __slots__ = ("armed", "dx")
    taken = 0          # gradients handed over so far (tests)

    def __init__(self):
        self.armed, self.dx = False, None


def residual_links_enabled():
    return os.environ.get("EDA_RESIDUAL_LINK", "1") != "0"


class _ProjectedMHA(Function):
    """In-projection + attention core as ONE autograd node (GPU training path of
    MultiheadAttention).  `groups` lists, per distinct input tensor, the row range [lo, hi) of
    in_proj_weight it is multiplied with -- ((0, 3d),) for self-attention, ((0, d), (d, 3d))
    when key is value, ((0, 2d), (2d, 3d)) when query is key, three ranges otherwise -- so each
    input costs one packed GEMM and q/k/v are column views of the packed results.  Compared with
    F.linear + split + _FusedMHA under autograd, the backward writes dq/dk/dv straight into the
    packed gradient buffers (no cat), the weight/bias gradients straight into their row ranges
    of ONE (3d,d)/(3d) buffer (no zero-fill + copy + add per slice), and skips nothing else:
    the GEMMs are the repo's own MFMA row GEMMs (csrc/gemm.hip)."""

    @staticmethod
    def forward(ctx, W, b, mask, num_heads, p_drop, salt, groups, link, *xs):
        d = W.shape[1]
        ctx.link = link                      # None or (ResidualLink, index into xs of the block's residual input)
        dev = W.device
        B = xs[0].shape[0]
        x2s, Ps, cols = [], [], {}
        m8 = mask.contiguous().view(torch.uint8) if mask is not None else None
        # a query projected on its own (cross-attention) over a short key set: the projection rides in the attention launch
        fuse_q = (groups[0] == (0, d) and len(groups) > 1 and b is not None
                  and _qproj_fused_ok(d, num_heads, xs[1].shape[1], *xs))
        for gi, (x, (lo, hi)) in enumerate(zip(xs, groups)):
            x2 = x.reshape(-1, d)
            P = None if (fuse_q and gi == 0) else gemm.linear_fwd(x2, W[lo:hi], b[lo:hi]).view(B, -1, hi - lo)
            x2s.append(x2)
            Ps.append(P)
            for j in range(lo // d, hi // d):
                cols[j] = (len(Ps) - 1, j * d - lo)
        if fuse_q:
            k, v = (Ps[g][..., c:c + d] for g, c in (cols[1], cols[2]))
            q, out, lse = _qproj_core_fwd(xs[0], W[:d], b[:d], k, v, m8, num_heads, p_drop, salt)
            Ps[0] = q
        else:
            q, k, v = (Ps[g][..., c:c + d] for g, c in (cols[0], cols[1], cols[2]))
            Lq, Lk = q.shape[1], k.shape[1]
            hd = d // num_heads
            out = torch.empty((B, Lq, d), dtype=torch.float32, device=dev)
            lse = torch.empty((B, num_heads, Lq), dtype=torch.float32, device=dev)
            _mha_fwd_call(q, k, v, m8, B, num_heads, Lq, Lk, hd, p_drop, salt, out, lse)
        ctx.dtype_code = _compute_dtype
        ctx.save_for_backward(W, out, lse, b, *x2s, *Ps)
        ctx.mask8 = m8
        ctx.cfg = (num_heads, float(p_drop), int(salt), groups, cols, [x.shape for x in xs])
        return out

    @staticmethod
    def backward(ctx, dout):
        num_heads, p_drop, salt, groups, cols, shapes = ctx.cfg
        n = len(groups)
        W, out, lse, bias = ctx.saved_tensors[:4]
        x2s, Ps = ctx.saved_tensors[4:4 + n], ctx.saved_tensors[4 + n:]
        d = W.shape[1]
        dev = W.device
        dout = _rows(dout)
        q, k, v = (Ps[g][..., c:c + d] for g, c in (cols[0], cols[1], cols[2]))
        B, Lq, Lk = q.shape[0], q.shape[1], k.shape[1]
        hd = d // num_heads
        dPs = [torch.empty_like(P) for P in Ps]
        dq, dk, dv = (dPs[g][..., c:c + d] for g, c in (cols[0], cols[1], cols[2]))
        m8 = ctx.mask8
        seed = dropout_state(dev) if p_drop > 0 else None
        _mha_bwd_call(q, k, v, m8, B, num_heads, Lq, Lk, hd, p_drop, seed, salt, out, lse, dout, dq, dk, dv, ctx.dtype_code)
        from . import wgrad_queue
        qd = wgrad_queue.active
        deferred = False
        if qd is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            # all row ranges of the packed weight / bias go to the deferred queue, or none
            plans = [qd.plan(W[lo:hi], bias[lo:hi], dPs[i].view(-1, hi - lo), x2s[i])
                     for i, (lo, hi) in enumerate(groups)]
            if all(pl is not None for pl in plans):
                for i, (lo, hi) in enumerate(groups):
                    qd.submit(W[lo:hi], bias[lo:hi], dPs[i].view(-1, hi - lo), x2s[i], planned=plans[i])
                deferred = True
        dW = torch.empty_like(W) if ctx.needs_input_grad[0] and not deferred else None
        db = (torch.empty((W.shape[0],), dtype=torch.float32, device=dev)
              if ctx.needs_input_grad[1] and not deferred else None)
        dxs = []
        for i, (lo, hi) in enumerate(groups):
            dP2 = dPs[i].view(-1, hi - lo)
            if dW is not None:
                wgrad(dP2, x2s[i], dW=dW[lo:hi], db=db[lo:hi] if db is not None else None,
                      want_db=db is not None)
            elif db is not None:
                colsum(dP2, out=db[lo:hi])
            if not ctx.needs_input_grad[8 + i]:
                dxs.append(None)
                continue
            res = None
            if ctx.link is not None and ctx.link[1] == i and ctx.link[0].dx is not None:
                res, ctx.link[0].dx = ctx.link[0].dx, None       # the residual LayerNorm's d(x): added in the product's epilogue
                ResidualLink.taken += 1
            dxs.append(gemm.linear_dgrad(dP2, W[lo:hi], out=res, addend=res).view(shapes[i]))
        if ctx.link is not None and ctx.link[0].dx is not None:
            raise RuntimeError("residual link: the LayerNorm node left a gradient that no input-gradient product took")
        return (dW, db, None, None, None, None, None, None, *dxs)


class KVSink:
    """Where the n consumers of a stacked K/V projection (StackedKV) put their dk | dv: side by side in ONE (B, Lk, n*2d)
    buffer, allocated by the first consumer's backward -- the projection's input gradient then is a single product over
    the concatenated contraction (no per-layer products, no gradient-accumulation adds)."""

    def __init__(self, n, width):
        self.n, self.width, self.buf = n, width, None

    def slot(self, i, B, Lk, device):
        if self.buf is None:
            self.buf = torch.empty((B, Lk, self.n * self.width), dtype=torch.float32, device=device)
        return self.buf[..., i * self.width:(i + 1) * self.width]


class _StackedKV(Function):
    """K | V projections of ONE memory tensor for n attention modules (the decoder layers' cross_l / cross_d / cross_v:
    text, boxes and points are the same for all six layers, encoder_decoder_layers.py:366-401) as one product:
    y = x [W_0[d:]; W_1[d:]; ...]^T.  `stack` = (Wst, bst): persistent (n*2d, d) / (n*2d) buffers refreshed by the caller
    (refresh_kv_stacks).  Outputs: n column views (B, Lk, 2d) of y.  Backward: one dX product over the n*2d columns when
    the consumers wrote their gradients into the sink, the weight / bias gradients per module (queued)."""

    @staticmethod
    def forward(ctx, x, sink, stack, *wb):
        n = len(wb) // 2
        d = x.shape[-1]
        Wst, bst = stack
        x2 = x.reshape(-1, d)
        y = gemm.linear_fwd(x2, Wst, bst).view(*x.shape[:-1], n * 2 * d)
        ctx.save_for_backward(x2, *wb)
        # (persistent buffer, rewritten in place by the next forward's refresh: not a saved tensor; the backward
        # re-assembles the stack from the saved parameters if it was rewritten in between)
        ctx.Wst, ctx.Wst_version = Wst, Wst._version
        ctx.sink, ctx.xshape, ctx.n = sink, x.shape, n
        ctx.set_materialize_grads(False)
        return tuple(y[..., 2 * d * i:2 * d * (i + 1)] for i in range(n))

    @staticmethod
    def backward(ctx, *dys):
        x2, Wst = ctx.saved_tensors[0], ctx.Wst
        wb = ctx.saved_tensors[1:]
        n, sink = ctx.n, ctx.sink
        d = x2.shape[1]
        if Wst._version != ctx.Wst_version:
            Wst = torch.cat([wb[2 * i][d:] for i in range(n)], 0)
        B, Lk = ctx.xshape[0], ctx.xshape[1]
        dall = None
        if sink.buf is not None and all(
                dy is not None and dy.data_ptr() == sink.buf.data_ptr() + 4 * 2 * d * i and dy.stride() == sink.buf.stride()
                for i, dy in enumerate(dys)):
            dall = sink.buf
        else:                                           # (consumers that did not use the sink: assemble)
            dall = torch.cat([dy if dy is not None else torch.full((B, Lk, 2 * d), 0.0, device=x2.device) for dy in dys], -1)
        sink.buf = None
        dall2 = dall.view(-1, n * 2 * d)
        dx = gemm.linear_dgrad(dall2, Wst).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        from . import wgrad_queue
        q = wgrad_queue.active
        grads = []
        for i in range(n):
            W, b = wb[2 * i], wb[2 * i + 1]
            dyi = dall2[:, 2 * d * i:2 * d * (i + 1)]
            need_w, need_b = ctx.needs_input_grad[3 + 2 * i], ctx.needs_input_grad[4 + 2 * i]
            if q is not None and need_w and need_b and q.submit(W[d:], b[d:], dyi, x2):
                grads += [None, None]
                continue
            dW = db = None
            if need_w:
                dW = torch.zeros_like(W)
                db = torch.zeros_like(b) if need_b else None
                wgrad(dyi.contiguous(), x2, dW=dW[d:], db=db[d:] if db is not None else None, want_db=db is not None)
            elif need_b:
                db = torch.zeros_like(b)
                colsum(dyi.contiguous(), out=db[d:])
            grads += [dW, db]
        return (dx, None, None, *grads)


def refresh_kv_stacks(stacks, modules_per_stack):
    """Copy the K | V rows of the modules' in-projection weights / biases into the persistent stacks (one multi-tensor
    launch for all of them): stacks = [(Wst, bst), ...], modules_per_stack = [[MultiheadAttention, ...], ...]."""
    dst, src = [], []
    for (Wst, bst), mods in zip(stacks, modules_per_stack):
        d = mods[0].embed_dim
        for i, m in enumerate(mods):
            dst += [Wst[2 * d * i:2 * d * (i + 1)], bst[2 * d * i:2 * d * (i + 1)]]
            src += [m.in_proj_weight.detach()[d:], m.in_proj_bias.detach()[d:]]
    with torch.no_grad():
        torch._foreach_copy_(dst, src)


class _ProjectedMHAPreKV(Function):
    """q-projection + attention core on K | V that were projected elsewhere (_StackedKV): kv (B, Lk, 2d) column view.
    Backward: dk | dv go into the sink's slot of this consumer (or a fresh tensor without a sink)."""

    @staticmethod
    def forward(ctx, W, b, mask, num_heads, p_drop, salt, sink, slot, x, kv):
        d = W.shape[1]
        dev = W.device
        B, Lq = x.shape[0], x.shape[1]
        x2 = x.reshape(-1, d)
        k, v = kv[..., :d], kv[..., d:]
        Lk = kv.shape[1]
        hd = d // num_heads
        m8 = mask.contiguous().view(torch.uint8) if mask is not None else None
        if b is not None and _qproj_fused_ok(d, num_heads, Lk, x, kv):
            # short key set (text tokens, detected boxes): the q-projection rides in the attention launch
            q, out, lse = _qproj_core_fwd(x, W[:d], b[:d], k, v, m8, num_heads, p_drop, salt)
        else:
            q = gemm.linear_fwd(x2, W[:d], b[:d]).view(B, Lq, d)
            out = torch.empty((B, Lq, d), dtype=torch.float32, device=dev)
            lse = torch.empty((B, num_heads, Lq), dtype=torch.float32, device=dev)
            _mha_fwd_call(q, k, v, m8, B, num_heads, Lq, Lk, hd, p_drop, salt, out, lse)
        ctx.dtype_code = _compute_dtype
        ctx.save_for_backward(W, b, x2, q, kv, out, lse)
        ctx.mask8 = m8
        ctx.cfg = (num_heads, float(p_drop), int(salt), sink, slot, x.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        num_heads, p_drop, salt, sink, slot, xshape = ctx.cfg
        W, bias, x2, q, kv, out, lse = ctx.saved_tensors
        d = W.shape[1]
        dev = W.device
        dout = _rows(dout)
        k, v = kv[..., :d], kv[..., d:]
        B, Lq, Lk = q.shape[0], q.shape[1], kv.shape[1]
        hd = d // num_heads
        dq = torch.empty_like(q)
        dkv = sink.slot(slot, B, Lk, dev) if sink is not None else torch.empty((B, Lk, 2 * d), dtype=torch.float32, device=dev)
        dk, dv = dkv[..., :d], dkv[..., d:]
        m8 = ctx.mask8
        seed = dropout_state(dev) if p_drop > 0 else None
        _mha_bwd_call(q, k, v, m8, B, num_heads, Lq, Lk, hd, p_drop, seed, salt, out, lse, dout, dq, dk, dv, ctx.dtype_code)
        from . import wgrad_queue
        qd = wgrad_queue.active
        dq2 = dq.view(-1, d)
        dW = db = None
        need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (qd is not None and need_w and need_b and qd.submit(W[:d], bias[:d], dq2, x2)):
            if need_w:
                dW = torch.zeros_like(W)
                db = torch.zeros_like(bias) if need_b else None
                wgrad(dq2, x2, dW=dW[:d], db=db[:d] if db is not None else None, want_db=db is not None)
            elif need_b:
                db = torch.zeros_like(bias)
                colsum(dq2, out=db[:d])
        dx = gemm.linear_dgrad(dq2, W[:d]).view(xshape) if ctx.needs_input_grad[8] else None
        return (dW, db, None, None, None, None, None, None, dx, dkv if ctx.needs_input_grad[9] else None)


def _hip_core(q, k, v, key_padding_mask, num_heads, dropout_p, salt):
    return _FusedMHA.apply(q, k, v, key_padding_mask, num_heads, dropout_p, salt)


# The implementation behind attention_core.  Product: the HIP kernels.  (CPU-side
# tests and bench.py's cpu_baseline leg swap in oracle/attention_ref.py themselves.)
_core = _hip_core


def attention_core(q, k, v, key_padding_mask=None, num_heads=8, dropout_p=0.0, salt=0):
    """softmax(q k^T / sqrt(hd) + mask) v per head, for PROJECTED q (B,Lq,D), k/v (B,Lk,D)
    with head h in columns [h*hd, (h+1)*hd).  key_padding_mask: (B,Lk) bool, True = ignore
    (a fully masked row gives NaN, as in the reference).  Returns (B,Lq,D)."""
    return _core(q, k, v, key_padding_mask, num_heads, dropout_p, salt)


class _OutProj(nn.Linear):
    pass


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = _OutProj(embed_dim, embed_dim, bias=True)
        # torch.nn.MultiheadAttention._reset_parameters
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)
        self._salt = next(_salt_counter)       # distinguishes this call site's dropout stream

    def __deepcopy__(self, memo):
        new = MultiheadAttention(self.embed_dim, self.num_heads, self.dropout)
        new.load_state_dict(self.state_dict())
        new.train(self.training)
        memo[id(self)] = new
        return new

    def hip_path(self, query):
        """True when forward() takes the HIP kernels for this query (then skip_out_proj is honoured)."""
        return query.is_cuda and _core is _hip_core

    def forward(self, query, key, value, key_padding_mask=None, need_weights=False,
                attn_mask=None, batch_first=False, defer_out_bias=False, skip_out_proj=False, pre_kv=None,
                residual=None):
        """Returns (output, None).  Inputs are (L,B,F) unless batch_first (then (B,L,F)).
        With defer_out_bias the out-projection is applied WITHOUT its bias and (output, bias) is
        returned: the caller's fused residual+LayerNorm kernel adds it (fused_ln.py).
        With skip_out_proj (HIP path only, see hip_path) the attention output BEFORE the out-projection is
        returned: the caller's fused kernel applies out_proj together with the residual LayerNorm."""
        if attn_mask is not None:
            raise NotImplementedError("EDA always passes attn_mask=None")
        if pre_kv is not None:
            # K | V of this module were projected together with its siblings' (_StackedKV): q-projection + core only
            assert batch_first and skip_out_proj and self.hip_path(query)
            kv, sink, slot = pre_kv
            o = _ProjectedMHAPreKV.apply(self.in_proj_weight, self.in_proj_bias, key_padding_mask, self.num_heads,
                                         self.dropout if self.training else 0.0, self._salt, sink, slot, query, kv)
            return o, None
        same_qk, same_kv = query is key, key is value
        if not batch_first:
            query = query.transpose(0, 1)
            key = query if same_qk else key.transpose(0, 1)
            value = key if same_kv else value.transpose(0, 1)
        d = self.embed_dim
        W, b = self.in_proj_weight, self.in_proj_bias
        if query.is_cuda and _core is _hip_core:
            if query is key and key is value:
                groups, xs = ((0, 3 * d),), (query,)
            elif key is value:
                groups, xs = ((0, d), (d, 3 * d)), (query, key)
            elif query is key:
                groups, xs = ((0, 2 * d), (2 * d, 3 * d)), (query, value)
            else:
                groups, xs = ((0, d), (d, 2 * d), (2 * d, 3 * d)), (query, key, value)
            link = None
            if residual is not None:
                # residual = (ResidualLink, x): x is the block's residual input; arm the link when x is one of this node's
                # inputs and its input-gradient product will run (see ResidualLink)
                rl, rx = residual
                gi = next((i for i, t in enumerate(xs) if t is rx), None)
                if (gi is not None and batch_first and skip_out_proj and rx.requires_grad and torch.is_grad_enabled()
                        and rx.dtype == torch.float32):
                    rl.armed = True
                    link = (rl, gi)
            o = _ProjectedMHA.apply(W, b, key_padding_mask, self.num_heads,
                                    self.dropout if self.training else 0.0, self._salt, groups, link, *xs)
            if skip_out_proj:
                return (o if batch_first else o.transpose(0, 1)), None
            o = linear_rows(o, self.out_proj.weight, None if defer_out_bias else self.out_proj.bias)
            if not batch_first:
                o = o.transpose(0, 1)
            return o, (self.out_proj.bias if defer_out_bias else None)
        assert not skip_out_proj, "skip_out_proj: HIP path only (check hip_path(query) first)"
        if query is key and key is value:
            q, k, v = F.linear(query, W, b).split(d, dim=-1)
        elif key is value:
            q = F.linear(query, W[:d], b[:d])
            k, v = F.linear(key, W[d:], b[d:]).split(d, dim=-1)
        elif query is key:
            q, k = F.linear(query, W[:2 * d], b[:2 * d]).split(d, dim=-1)
            v = F.linear(value, W[2 * d:], b[2 * d:])
        else:
            q = F.linear(query, W[:d], b[:d])
            k = F.linear(key, W[d:2 * d], b[d:2 * d])
            v = F.linear(value, W[2 * d:], b[2 * d:])
        o = attention_core(q, k, v, key_padding_mask, self.num_heads,
                           self.dropout if self.training else 0.0, self._salt)
        o = F.linear(o, self.out_proj.weight, None if defer_out_bias else self.out_proj.bias)
        if not batch_first:
            o = o.transpose(0, 1)
        return o, (self.out_proj.bias if defer_out_bias else None)
