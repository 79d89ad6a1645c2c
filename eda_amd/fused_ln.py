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
