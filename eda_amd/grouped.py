"""Sibling layers in one launch (csrc/gemm.hip grouped launches, csrc/sa_cl.hip grouped BatchNorm).

The prediction heads of the reference (models/modules.py:111-178, ``ClsAgnosticPredictHead``) run
three independent ``ThreeLayerMLP`` stacks of the same shape on the same query features -- fifteen
small launches per head forward in the reference's formulation, seven heads per step, each bound by
launch latency, not by arithmetic.  Here layer l of all siblings is ONE launch:

* ``shared_in_linear``   y_g = x W_g^T            one input, G weights          -> packed (R, sum N_g)
* ``grouped_bn_relu``    relu(BN_g(z_g)) [+ dropout] on column blocks of a packed (R, G*C) matrix
* ``block_linear``       y_g = x_g W_g^T + b_g    x_g = column block g of a packed matrix

with matching one-launch backward passes (the input gradient of ``shared_in_linear`` is a single
GEMM over the concatenated weights when they are adjacent in memory -- FlatParams lays sibling
parameters out that way -- which also removes the gradient-accumulation adds autograd would spend
on a tensor with three consumers).  Weight / bias gradients go to the deferred queue
(eda_amd/wgrad_queue.py) or, outside it, through the immediate kernels, exactly like
``nn_utils._LinearRows``.  GPU only; callers keep the per-module path for CPU tensors.
"""
import ctypes

import torch
from torch.autograd import Function

from . import _lib, gemm
from .ext import _timed


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _parr(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts])


def _larr(vs):
    return (ctypes.c_long * len(vs))(*[int(v) for v in vs])


def _iarr(vs):
    return (ctypes.c_int * len(vs))(*[int(v) for v in vs])


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


def _grouped_fwd(xs, Ws, bs, ys, relu=False):
    G = len(Ws)
    R = xs[0].shape[0]
    if R == 0:
        return
    with torch.cuda.device(xs[0].device), _timed("gemm_grouped_fwd", (G, R, Ws[0].shape[1], sum(w.shape[0] for w in Ws))):
        rc = _lib.lib().eda_linear_grouped_fwd_f32(
            G, _parr(xs), _larr([_ld(x) for x in xs]), R, _iarr([w.shape[1] for w in Ws]), _parr(Ws),
            _larr([_ld(w) for w in Ws]), _iarr([w.shape[0] for w in Ws]),
            _parr(bs) if any(b is not None for b in bs) else None, int(bool(relu)), _parr(ys),
            _larr([_ld(y) for y in ys]), _stream())
    _lib.check(rc, "eda_linear_grouped_fwd_f32")


def _grouped_dgrad(dys, Ws, dxs):
    G = len(Ws)
    R = dys[0].shape[0]
    if R == 0:
        return
    with torch.cuda.device(dys[0].device), _timed("gemm_grouped_dgrad", (G, R, sum(w.shape[0] for w in Ws), Ws[0].shape[1])):
        rc = _lib.lib().eda_linear_grouped_dgrad_f32(
            G, _parr(dys), _larr([_ld(d) for d in dys]), R, _iarr([w.shape[0] for w in Ws]), _parr(Ws),
            _larr([_ld(w) for w in Ws]), _iarr([w.shape[1] for w in Ws]), _parr(dxs), _larr([_ld(d) for d in dxs]),
            _stream())
    _lib.check(rc, "eda_linear_grouped_dgrad_f32")


def _packed_view(Ws):
    """(sum N, K) view over G row-major (N_g, K) weights that lie back to back in memory, else None."""
    K = Ws[0].shape[1]
    ptr = Ws[0].data_ptr()
    for w in Ws:
        if not w.is_contiguous() or w.shape[1] != K or w.data_ptr() != ptr:
            return None
        ptr += w.numel() * 4
    n = sum(w.shape[0] for w in Ws)
    return torch.as_strided(Ws[0], (n, K), (K, 1))


def _weight_grads(Ws, bs, dys, xs, need_w, need_b):
    """(dW_g, db_g) lists: deferred to the queue (None entries) where it accepts the job."""
    from . import wgrad_queue
    from .nn_utils import colsum, wgrad
    q = wgrad_queue.active
    dWs, dbs = [], []
    for g, W in enumerate(Ws):
        b = bs[g]
        want_w, want_b = need_w[g], b is not None and need_b[g]
        if (q is not None and want_w and (want_b or b is None) and q.submit(W, b if want_b else None, dys[g], xs[g])):
            dWs.append(None); dbs.append(None)
            continue
        dW = db = None
        if want_w:
            dW, db = wgrad(dys[g], xs[g], want_db=want_b)
        elif want_b:
            db = colsum(dys[g])
        dWs.append(dW); dbs.append(db)
    return dWs, dbs


class _SharedInLinear(Function):
    @staticmethod
    def forward(ctx, x, *Ws):
        x2 = x if x.stride(1) == 1 else x.contiguous()
        Ws2 = [w.reshape(w.shape[0], -1) for w in Ws]
        R = x2.shape[0]
        widths = [w.shape[0] for w in Ws2]
        out = torch.empty((R, sum(widths)), dtype=torch.float32, device=x.device)
        offs = [sum(widths[:g]) for g in range(len(Ws2))]
        ys = [out[:, o:o + n] for o, n in zip(offs, widths)]
        _grouped_fwd([x2] * len(Ws2), Ws2, [None] * len(Ws2), ys)
        ctx.save_for_backward(x2, *Ws2)
        ctx.offs, ctx.widths = offs, widths
        ctx.wshapes = [w.shape for w in Ws]
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, *Ws2 = ctx.saved_tensors
        G = len(Ws2)
        dout = dout if dout.stride(1) == 1 and dout.stride(0) % 4 == 0 else dout.contiguous()
        dys = [dout[:, o:o + n] for o, n in zip(ctx.offs, ctx.widths)]
        dx = None
        if ctx.needs_input_grad[0]:
            packed = _packed_view(Ws2)
            if packed is not None:
                dx = gemm.linear_dgrad(dout, packed)                    # ONE GEMM, the siblings' terms summed in it
            else:
                dx = gemm.linear_dgrad(dys[0], Ws2[0])
                for g in range(1, G):
                    dx = dx + gemm.linear_dgrad(dys[g], Ws2[g])
        dWs, _ = _weight_grads(Ws2, [None] * G, dys, [x2] * G, ctx.needs_input_grad[1:], [False] * G)
        return (dx, *[d.view(s) if d is not None else None for d, s in zip(dWs, ctx.wshapes)])


class _BlockLinear(Function):
    @staticmethod
    def forward(ctx, xp, pack_out, relu, *params):
        G = len(params) // 2
        Ws = [params[2 * g].reshape(params[2 * g].shape[0], -1) for g in range(G)]
        bs = [params[2 * g + 1] for g in range(G)]
        K = Ws[0].shape[1]
        assert xp.shape[1] == G * K and all(w.shape[1] == K for w in Ws)
        xp2 = xp if xp.stride(1) == 1 else xp.contiguous()
        R = xp2.shape[0]
        xs = [xp2[:, g * K:(g + 1) * K] for g in range(G)]
        widths = [w.shape[0] for w in Ws]
        if pack_out:
            out = torch.empty((R, sum(widths)), dtype=torch.float32, device=xp.device)
            ys = [out[:, sum(widths[:g]):sum(widths[:g + 1])] for g in range(G)]
        else:
            ys = [torch.empty((R, n), dtype=torch.float32, device=xp.device) for n in widths]
        _grouped_fwd(xs, Ws, bs, ys, relu)
        ctx.save_for_backward(xp2, *Ws, *[b for b in bs if b is not None], *(ys if relu else []))
        ctx.cfg = (G, K, widths, bool(pack_out), bool(relu), [b is not None for b in bs],
                   [p.shape for p in params[0::2]])
        return out if pack_out else tuple(ys)

    @staticmethod
    def backward(ctx, *douts):
        G, K, widths, pack_out, relu, has_b, wshapes = ctx.cfg
        sv = list(ctx.saved_tensors)
        xp2, Ws = sv[0], sv[1:1 + G]
        nb = sum(has_b)
        bl = sv[1 + G:1 + G + nb]
        ys = sv[1 + G + nb:]
        bs, k = [], 0
        for g in range(G):
            bs.append(bl[k] if has_b[g] else None)
            k += 1 if has_b[g] else 0
        R = xp2.shape[0]
        if pack_out:
            d = douts[0]
            d = d if d.stride(1) == 1 and d.stride(0) % 4 == 0 else d.contiguous()
            dys = [d[:, sum(widths[:g]):sum(widths[:g + 1])] for g in range(G)]
        else:
            dys = [(d if d is not None else torch.full((R, widths[g]), 0.0, device=xp2.device)) for g, d in enumerate(douts)]
            dys = [d if d.stride(1) == 1 else d.contiguous() for d in dys]
        if relu:
            dys = [torch.ops.aten.threshold_backward(d, y, 0.0) for d, y in zip(dys, ys)]
        xs = [xp2[:, g * K:(g + 1) * K] for g in range(G)]
        dxp = None
        if ctx.needs_input_grad[0]:
            dxp = torch.empty((R, G * K), dtype=torch.float32, device=xp2.device)
            dxs = [dxp[:, g * K:(g + 1) * K] for g in range(G)]
            if all(n % 4 == 0 for n in widths):
                _grouped_dgrad(dys, Ws, dxs)
            else:
                # a 1- or 3-wide output has no 16-byte rows: those groups take the element-wise kernel on their own
                for g in range(G):
                    gemm.linear_dgrad(dys[g], Ws[g], out=dxs[g])
        need_w = [ctx.needs_input_grad[3 + 2 * g] for g in range(G)]
        need_b = [ctx.needs_input_grad[4 + 2 * g] for g in range(G)]
        dWs, dbs = _weight_grads(Ws, bs, dys, xs, need_w, need_b)
        grads = []
        for g in range(G):
            grads += [dWs[g].view(wshapes[g]) if dWs[g] is not None else None, dbs[g]]
        return (dxp, None, None, *grads)


def bn_relu_fwd_raw(zp, cfg, gammas, betas):
    """The grouped BatchNorm+ReLU(+Dropout) forward launch: (out, stats (4, G*C) rows mean | rstd | scale | shift)."""
    G, C, training, eps, momentum, p_drop, salts, running = cfg
    zp = zp.contiguous()
    R = zp.shape[0]
    dev = zp.device
    stats = torch.empty((4, G * C), dtype=torch.float32, device=dev)
    out = torch.empty_like(zp)
    seed = None
    if p_drop > 0:
        from .attention import dropout_state
        seed = dropout_state(dev)
    salt_arr = (ctypes.c_uint * G)(*[int(s) & 0xFFFFFFFF for s in salts])
    with torch.cuda.device(dev), _timed("bn_relu_grouped_fwd", (R, G, C, int(training))):
        rc = _lib.lib().eda_bn_relu_grouped_fwd_f32(
            zp.data_ptr(), R, G, C, _parr(gammas), _parr(betas), _parr([r[0] for r in running]),
            _parr([r[1] for r in running]), float(eps), float(momentum), int(bool(training)),
            stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), out.data_ptr(),
            float(p_drop), seed.data_ptr() if seed is not None else None, salt_arr, _stream())
    _lib.check(rc, "eda_bn_relu_grouped_fwd_f32")
    return zp, out, stats


class _GroupedBNReLU(Function):
    @staticmethod
    def forward(ctx, zp, cfg, *params):
        """cfg = (G, C, training, eps, momentum, p_drop, salts, running=[(rm, rv), ...]); params = gamma_0, beta_0, ..."""
        G, C, training, eps, momentum, p_drop, salts, running = cfg
        gammas, betas = list(params[0::2]), list(params[1::2])
        zp, out, stats = bn_relu_fwd_raw(zp, cfg, gammas, betas)
        ctx.save_for_backward(zp, stats, *gammas)
        ctx.cfg = (G, C, bool(training), float(p_drop), [int(s) & 0xFFFFFFFF for s in salts])
        return out

    @staticmethod
    def backward(ctx, dout):
        zp, stats, *gammas = ctx.saved_tensors
        G, C, training, p_drop, salts = ctx.cfg
        R = zp.shape[0]
        dev = zp.device
        dout = dout.contiguous()
        dz = torch.empty_like(zp)
        dgb = torch.empty((2, G * C), dtype=torch.float32, device=dev)
        seed = None
        if p_drop > 0:
            from .attention import dropout_state
            seed = dropout_state(dev)
        salt_arr = (ctypes.c_uint * G)(*salts)
        with torch.cuda.device(dev), _timed("bn_relu_grouped_bwd", (R, G, C, int(training))):
            rc = _lib.lib().eda_bn_relu_grouped_bwd_f32(
                dout.data_ptr(), zp.data_ptr(), R, G, C, _parr(gammas), stats[0].data_ptr(), stats[1].data_ptr(),
                stats[2].data_ptr(), stats[3].data_ptr(), int(training), dgb[0].data_ptr(), dgb[1].data_ptr(),
                dz.data_ptr(), p_drop, seed.data_ptr() if seed is not None else None, salt_arr, _stream())
        _lib.check(rc, "eda_bn_relu_grouped_bwd_f32")
        grads = []
        for g in range(G):
            grads += [dgb[0][g * C:(g + 1) * C], dgb[1][g * C:(g + 1) * C]]
        return (dz, None, *grads)


def shared_in_linear(x, weights):
    """[x W_g^T for g] packed along the columns: (R, sum N_g)."""
    return _SharedInLinear.apply(x, *weights)


def block_linear(xp, weights, biases, pack_out, relu=False):
    """Column block g of xp times W_g^T (+ b_g): packed (R, sum N_g) or a tuple of (R, N_g)."""
    params = []
    for w, b in zip(weights, biases):
        params += [w, b]
    return _BlockLinear.apply(xp, pack_out, relu, *params)


def bn_relu_cfg(zp, bns, dropouts=None):
    """cfg tuple of the grouped BatchNorm+ReLU(+Dropout) launch for modules `bns` / `dropouts` on packed rows zp."""
    from .nn_utils import dropout_salt
    G = len(bns)
    C = zp.shape[1] // G
    training = bns[0].training
    p = 0.0
    salts = [0] * G
    if dropouts is not None and dropouts[0] is not None and dropouts[0].training and dropouts[0].p > 0:
        p = float(dropouts[0].p)
        for g, d in enumerate(dropouts):
            salts[g] = dropout_salt(d)
    return (G, C, training or not bns[0].track_running_stats, bns[0].eps, bns[0].momentum, p, salts,
            [(bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None) for bn in bns])


def grouped_bn_relu(zp, bns, dropouts=None):
    """relu(BatchNorm bns[g](zp[:, g*C:(g+1)*C])) (+ the heads' Dropout, fused, in training)."""
    from .nn_utils import bump_batches_tracked
    cfg = bn_relu_cfg(zp, bns, dropouts)
    params = []
    for bn in bns:
        params += [bn.weight, bn.bias]
    out = _GroupedBNReLU.apply(zp, cfg, *params)
    if bns[0].training:
        for bn in bns:
            if bn.track_running_stats:
                bump_batches_tracked(bn)
    return out
