"""LayerNorm(x + dropout(y + bias)) as one fused HIP kernel pair (csrc/ln.hip).

Every post-norm block of the reference's encoder/decoder has this shape
(models/encoder_decoder_layers.py:94-96,106-122,154-156,184-186,371-405).  On the GPU the
three stock launches (dropout, add, layer_norm) and their ~five backward launches become one
forward and one backward kernel; on CPU tensors (host-logic tests) the same expression is
evaluated with plain torch ops.  `y_bias` is the bias of the linear that produced y (attention
out-projection / second FFN linear) when the caller applied that linear without it: adding it
here is free, and so is its gradient (a third column sum next to d(gamma), d(beta)), which
saves the separate row-reduction launch autograd would otherwise spend on it.
"""
import itertools

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib
from .attention import dropout_state
from .ext import _timed

_salt_counter = itertools.count(1 << 20)       # distinct from the attention call-site salts


def new_salt_base():
    return next(_salt_counter) * 16


class _AddDropoutLN(Function):
    @staticmethod
    def forward(ctx, x, y, y_bias, gamma, beta, eps, p_drop, salt, pos=None):
        shape = x.shape
        C = shape[-1]
        x2 = x.reshape(-1, C).contiguous()
        y2 = y.reshape(-1, C).contiguous()
        R = x2.shape[0]
        dev = x.device
        out = torch.empty_like(x2)
        pos2 = pos.reshape(-1, C).contiguous() if pos is not None else None
        out_pos = torch.empty_like(x2) if pos is not None else None
        stats = torch.empty((2, R), dtype=torch.float32, device=dev)
        seed = dropout_state(dev) if p_drop > 0 else None
        with torch.cuda.device(dev), _timed("add_dropout_ln_fwd", (R, C)):
            rc = _lib.lib().eda_add_dropout_ln_fwd_f32(
                x2.data_ptr(), y2.data_ptr(), y_bias.data_ptr() if y_bias is not None else None,
                gamma.data_ptr(), beta.data_ptr(), R, C, float(eps),
                float(p_drop), seed.data_ptr() if seed is not None else None, int(salt),
                out.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                pos2.data_ptr() if pos2 is not None else None,
                out_pos.data_ptr() if out_pos is not None else None,
                torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "eda_add_dropout_ln_fwd_f32")
        ctx.save_for_backward(x2, y2, gamma, stats, y_bias)
        ctx.beta_ref = beta                     # only its storage location matters (deferred gradients)
        ctx.cfg = (float(p_drop), int(salt), shape)
        ctx.with_pos = pos is not None
        ctx.set_materialize_grads(False)        # an unused output arrives as None, not as a zero tensor
        if pos is not None:
            return out.view(shape), out_pos.view(shape)
        return out.view(shape)

    @staticmethod
    def backward(ctx, dout, dout_pos=None):
        x2, y2, gamma, stats, y_bias = ctx.saved_tensors
        p_drop, salt, shape = ctx.cfg
        R, C = x2.shape
        dev = x2.device
        if dout is None and dout_pos is not None:           # only out + pos was used
            dout, dout_pos_k = dout_pos, None
        elif dout is None:
            dout, dout_pos_k = torch.full((R, C), 0.0, device=dev), None
        else:
            dout_pos_k = dout_pos
        dout = dout.reshape(R, C).contiguous()
        dout2 = dout_pos_k.reshape(R, C).contiguous() if dout_pos_k is not None else None
        dx = torch.empty_like(x2)
        dy = torch.empty_like(x2)
        from . import wgrad_queue
        q = wgrad_queue.active
        planned = q.plan_ln(gamma, ctx.beta_ref, y_bias) if q is not None else None
        g3 = torch.empty((3, C), dtype=torch.float32, device=dev) if planned is None else None
        L = _lib.lib()
        ws_bytes = L.eda_add_dropout_ln_bwd_workspace_bytes(R, C)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        seed = dropout_state(dev) if p_drop > 0 else None
        with torch.cuda.device(dev), _timed("add_dropout_ln_bwd", (R, C)):
            rc = L.eda_add_dropout_ln_bwd_f32(
                dout.data_ptr(), x2.data_ptr(), y2.data_ptr(),
                y_bias.data_ptr() if y_bias is not None else None, gamma.data_ptr(),
                stats[0].data_ptr(), stats[1].data_ptr(), R, C, p_drop,
                seed.data_ptr() if seed is not None else None, salt, dx.data_ptr(), dy.data_ptr(),
                g3.data_ptr() if g3 is not None else None, ws.data_ptr(), ws_bytes,
                dout2.data_ptr() if dout2 is not None else None,
                torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "eda_add_dropout_ln_bwd_f32")
        dpos = (dout_pos.reshape(shape) if dout_pos is not None else None) if ctx.with_pos else None
        if planned is not None:
            # d(gamma), d(beta), d(bias): per-block partial sums stay in `ws`; the queue reduces all
            # LayerNorm sites of the backward pass in one launch, straight into the gradient buffer
            q.submit_ln(planned, ws, L.eda_add_dropout_ln_bwd_blocks(R), C)
            return dx.view(shape), dy.view(shape), None, None, None, None, None, None, dpos
        return (dx.view(shape), dy.view(shape), g3[2] if y_bias is not None else None, g3[0], g3[1],
                None, None, None, dpos)


def _ln_backward(ctx_dev, dout, dout_pos, with_pos, x_or_z, y2, y_bias, gamma, beta_ref, stats, p_drop, salt):
    """The LayerNorm backward kernel shared by the two autograd nodes: returns (dx, dy, g3 or None, dpos-passthrough
    flag).  y2 None: x_or_z is the pre-norm sum (fused linear forward)."""
    R, C = x_or_z.shape
    dev = ctx_dev
    if dout is None and dout_pos is not None:           # only out + pos was used
        dout, dout_pos_k = dout_pos, None
    elif dout is None:
        dout, dout_pos_k = torch.full((R, C), 0.0, device=dev), None
    else:
        dout_pos_k = dout_pos
    dout = dout.reshape(R, C).contiguous()
    dout2 = dout_pos_k.reshape(R, C).contiguous() if dout_pos_k is not None else None
    dx = torch.empty_like(x_or_z)
    dy = torch.empty_like(x_or_z)
    from . import wgrad_queue
    q = wgrad_queue.active
    planned = q.plan_ln(gamma, beta_ref, y_bias) if q is not None else None
    g3 = torch.empty((3, C), dtype=torch.float32, device=dev) if planned is None else None
    L = _lib.lib()
    ws_bytes = L.eda_add_dropout_ln_bwd_workspace_bytes(R, C)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    seed = dropout_state(dev) if p_drop > 0 else None
    with torch.cuda.device(dev), _timed("add_dropout_ln_bwd", (R, C)):
        rc = L.eda_add_dropout_ln_bwd_f32(
            dout.data_ptr(), x_or_z.data_ptr(), y2.data_ptr() if y2 is not None else None,
            y_bias.data_ptr() if y_bias is not None else None, gamma.data_ptr(),
            stats[0].data_ptr(), stats[1].data_ptr(), R, C, p_drop,
            seed.data_ptr() if seed is not None else None, salt, dx.data_ptr(), dy.data_ptr(),
            g3.data_ptr() if g3 is not None else None, ws.data_ptr(), ws_bytes,
            dout2.data_ptr() if dout2 is not None else None,
            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_add_dropout_ln_bwd_f32")
    if planned is not None:
        # d(gamma), d(beta), d(bias): per-block partial sums stay in `ws`; the queue reduces all
        # LayerNorm sites of the backward pass in one launch, straight into the gradient buffer
        q.submit_ln(planned, ws, L.eda_add_dropout_ln_bwd_blocks(R), C)
    return dx, dy, g3


def _linear_ln_forward(inp2, W, b, x2, gamma, beta, eps, p_drop, salt, pos2, want_z=True):
    """One launch: z = x2 + Dropout(inp2 W^T + b), out = LayerNorm(z) (+ out + pos2).  Returns (out, out_pos, z, stats)."""
    from . import gemm
    R, C = x2.shape
    dev = x2.device
    inp2, W = gemm._rows2d(inp2), gemm._rows2d(W)
    out = torch.empty_like(x2)
    out_pos = torch.empty_like(x2) if pos2 is not None else None
    z = torch.empty_like(x2) if want_z else None
    stats = torch.empty((2, R), dtype=torch.float32, device=dev)
    seed = dropout_state(dev) if p_drop > 0 else None
    # scratch of the split contraction (below 4096 rows two workgroups share a row block: include/eda_hip.h) -- the same
    # persistent per-stream buffer the split row products use
    need = int(_lib.lib().eda_linear_add_dropout_ln_workspace_bytes(R, inp2.shape[1], C))
    ws = gemm.workspace(dev, need) if need else None
    with torch.cuda.device(dev), _timed("linear_add_dropout_ln_fwd", (R, inp2.shape[1], C)):
        rc = _lib.lib().eda_linear_add_dropout_ln_fwd_ws_f32(
            inp2.data_ptr(), gemm._ld(inp2), R, inp2.shape[1], W.data_ptr(), gemm._ld(W), C,
            b.data_ptr() if b is not None else None, x2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps),
            float(p_drop), seed.data_ptr() if seed is not None else None, int(salt),
            z.data_ptr() if z is not None else None, out.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
            pos2.data_ptr() if pos2 is not None else None, out_pos.data_ptr() if out_pos is not None else None,
            ws.data_ptr() if ws is not None else None, ws.numel() * 4 if ws is not None else 0,
            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_linear_add_dropout_ln_fwd_ws_f32")
    return out, out_pos, z, stats


class _LinearAddDropoutLN(Function):
    """out = LayerNorm(x + Dropout(inp W^T + b)) as ONE forward launch (csrc/gemm.hip, gemm_dma_kernel<LN>): the
    attention out-projection and its residual LayerNorm.  Backward: the LayerNorm backward kernel on the saved pre-norm
    sum (d(gamma), d(beta) and d(b) ride in it), then the linear layer's own dX GEMM and queued weight gradient."""

    @staticmethod
    def forward(ctx, inp, W, b, x, gamma, beta, eps, p_drop, salt, pos=None, link=None):
        shape = x.shape
        ctx.link = link if (link is not None and link.armed) else None      # attention.ResidualLink: d(x) goes to the attention node
        C = shape[-1]
        x2 = x.reshape(-1, C).contiguous()
        inp2 = inp.reshape(-1, inp.shape[-1])
        pos2 = pos.reshape(-1, C).contiguous() if pos is not None else None
        out, out_pos, z, stats = _linear_ln_forward(inp2, W, b, x2, gamma, beta, eps, p_drop, salt, pos2)
        ctx.save_for_backward(inp2, W, b, z, gamma, stats)
        ctx.beta_ref = beta
        ctx.cfg = (float(p_drop), int(salt), shape, inp.shape)
        ctx.with_pos = pos is not None
        ctx.set_materialize_grads(False)
        if pos is not None:
            return out.view(shape), out_pos.view(shape)
        return out.view(shape)

    @staticmethod
    def backward(ctx, dout, dout_pos=None):
        from . import gemm, wgrad_queue
        from .nn_utils import wgrad
        inp2, W, b, z, gamma, stats = ctx.saved_tensors
        p_drop, salt, shape, ishape = ctx.cfg
        dx, dy, g3 = _ln_backward(z.device, dout, dout_pos, ctx.with_pos, z, None, b, gamma, ctx.beta_ref, stats, p_drop, salt)
        dpos = (dout_pos.reshape(shape) if dout_pos is not None else None) if ctx.with_pos else None
        dinp = gemm.linear_dgrad(dy, W).view(ishape) if ctx.needs_input_grad[0] else None
        dW = None
        q = wgrad_queue.active
        if ctx.needs_input_grad[1] and not (q is not None and q.submit(W, None, dy, inp2)):
            dW, _ = wgrad(dy, inp2, want_db=False)
        if ctx.link is not None and ctx.needs_input_grad[3]:
            ctx.link.dx, dxo = dx, None         # added in the epilogue of the attention node's input-gradient product for x
        else:
            dxo = dx.view(shape)
        if g3 is None:
            return dinp, dW, None, dxo, None, None, None, None, None, dpos, None
        return (dinp, dW, g3[2] if b is not None else None, dxo, g3[0], g3[1], None, None, None, dpos, None)


class _FFNAddDropoutLN(Function):
    """out = LayerNorm(x + Dropout(W2 Dropout(ReLU(W1 x + b1)) + b2)): the post-norm FFN block as one autograd node and
    TWO forward launches (the first linear with ReLU + Dropout in its epilogue, the second with the residual LayerNorm in
    its epilogue).  Backward: LayerNorm backward, the second layer's gated dX (ReLU / Dropout backward in its epilogue),
    the first layer's dX; weight gradients queued."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, gamma, beta, eps, p1, salt1, p2, salt2, pos=None):
        from . import gemm
        shape = x.shape
        C = shape[-1]
        x2 = x.reshape(-1, C).contiguous()
        pos2 = pos.reshape(-1, C).contiguous() if pos is not None else None
        seed = dropout_state(x.device) if p1 > 0 else None
        h = gemm.linear_ex(x2, W1, b1, True, drop=(p1, seed, salt1) if p1 > 0 else None)
        out, out_pos, z, stats = _linear_ln_forward(h, W2, b2, x2, gamma, beta, eps, p2, salt2, pos2)
        ctx.save_for_backward(x2, h, W1, b1, W2, b2, z, gamma, stats)
        ctx.beta_ref = beta
        ctx.cfg = (float(p1), float(p2), int(salt2), shape)
        ctx.with_pos = pos is not None
        ctx.set_materialize_grads(False)
        if pos is not None:
            return out.view(shape), out_pos.view(shape)
        return out.view(shape)

    @staticmethod
    def backward(ctx, dout, dout_pos=None):
        from . import gemm, wgrad_queue
        from .nn_utils import colsum, wgrad
        x2, h, W1, b1, W2, b2, z, gamma, stats = ctx.saved_tensors
        p1, p2, salt2, shape = ctx.cfg
        dx, dy, g3 = _ln_backward(z.device, dout, dout_pos, ctx.with_pos, z, None, b2, gamma, ctx.beta_ref, stats, p2, salt2)
        dpos = (dout_pos.reshape(shape) if dout_pos is not None else None) if ctx.with_pos else None
        q = wgrad_queue.active
        grads = {}
        if ctx.needs_input_grad[3] and not (q is not None and q.submit(W2, None, dy, h)):
            grads["W2"], _ = wgrad(dy, h, want_db=False)
        dh = gemm.linear_dgrad_gated(dy, W2, h, 1.0 / (1.0 - p1) if p1 > 0 else 1.0)     # d(pre-activation of layer 1)
        need_w1, need_b1 = ctx.needs_input_grad[1], b1 is not None and ctx.needs_input_grad[2]
        if not (q is not None and need_w1 and (need_b1 or b1 is None) and q.submit(W1, b1 if need_b1 else None, dh, x2)):
            if need_w1:
                grads["W1"], db1 = wgrad(dh, x2, want_db=need_b1)
                if need_b1:
                    grads["b1"] = db1
            elif need_b1:
                grads["b1"] = colsum(dh)
        dxin = None
        if ctx.needs_input_grad[0]:
            dx2 = dx.reshape(-1, x2.shape[1])
            dxin = gemm.linear_dgrad(dh, W1, out=dx2, addend=dx2).view(shape)     # residual term in the product's epilogue
        if g3 is None:
            return (dxin, grads.get("W1"), grads.get("b1"), grads.get("W2"), None, None, None, None, None, None, None, None, dpos)
        return (dxin, grads.get("W1"), grads.get("b1"), grads.get("W2"), g3[2] if b2 is not None else None, g3[0], g3[1],
                None, None, None, None, None, dpos)


def fuses_linear(x, norm, K):
    """True when the linear layer in front of `norm` can ride in the fused kernel (eda_linear_add_dropout_ln_fwd_f32)."""
    import os
    return (fuses_bias(x, norm) and os.environ.get("EDA_FUSED_LINEAR_LN", "1") != "0"
            and bool(_lib.lib().eda_linear_add_dropout_ln_supported(int(K), int(x.shape[-1]))))


def linear_add_dropout_layer_norm(inp, weight, bias, x, norm, p_drop, training, salt, pos=None, link=None):
    """norm(x + dropout(inp @ weight.T + bias)) (+ pos): one launch on the GPU when fuses_linear(x, norm, K)."""
    p = float(p_drop) if training else 0.0
    if fuses_linear(x, norm, weight.shape[1]) and inp.dtype == torch.float32:
        return _LinearAddDropoutLN.apply(inp, weight, bias, x, norm.weight, norm.bias, norm.eps, p, salt, pos, link)
    from .nn_utils import linear_rows
    fb = fuses_bias(x, norm)
    y = linear_rows(inp, weight, None if fb else bias)
    return add_dropout_layer_norm(x, y, norm, p_drop, training, salt, y_bias=bias if fb else None, pos=pos)


def fuses_bias(x, norm):
    """True when add_dropout_layer_norm takes the HIP path for (x, norm), i.e. when a caller may
    apply the preceding linear without its bias and hand the bias over as `y_bias`."""
    return (x.is_cuda and norm.elementwise_affine and x.shape[-1] <= 1024
            and x.dtype == torch.float32)


def add_dropout_layer_norm(x, y, norm, p_drop, training, salt, y_bias=None, pos=None):
    """out = norm(x + dropout(y + y_bias, p_drop)) for an nn.LayerNorm `norm` over the last dim.
    With `pos` (same shape as x): returns (out, out + pos) -- the next attention block's query comes
    out of the same launch, and in the backward the gradients of both are summed inside the kernel."""
    p = float(p_drop) if training else 0.0
    if fuses_bias(x, norm):
        if pos is not None:
            return _AddDropoutLN.apply(x, y, y_bias, norm.weight, norm.bias, norm.eps, p, salt, pos)
        return _AddDropoutLN.apply(x, y, y_bias, norm.weight, norm.bias, norm.eps, p, salt)
    if y_bias is not None:
        y = y + y_bias
    out = norm(x + F.dropout(y, p, training=p > 0))
    return (out, out + pos) if pos is not None else out
