"""The backward passes of the decoder layers' learned positional embeddings, issued together at the END of the backward.

`BiDecoderLayer.self_posembed` (models/encoder_decoder_layers.py:341-365: conv1d 6 -> 288, BatchNorm, ReLU, conv1d 288 -> 288)
embeds box estimates that reach it detached (models/bdetr.py:262-264, 316-320), so nothing waits for its backward: the
gradient of its output only ends in the module's own parameters.  Per layer that backward is an input-gradient product, a
single-launch BatchNorm+ReLU backward, a 6-column weight gradient through the library's GEMM and a column sum -- ~5 launches
of <= 192 workgroups, six layers.  Here the forward runs where it always ran but without autograd (same launches, intermediates
kept); a stub node hands the output to the graph, its backward only stores the incoming gradient and queues one callback for the
end of the backward pass (`Engine.queue_callback`), which runs every stage for all layers at once: grouped input gradient
(eda_linear_grouped_dgrad_f32), the BatchNorm backward of six matrices in one launch (eda_bn_relu_grouped_bwd_multi_f32), the
6-column weight gradients as one batched product; the 288-wide weight / bias gradients go to the deferred queue.  The
parameters' `.grad` is assigned by the callback exactly where autograd would have put it.

OPT-IN.  Because the callback writes `.grad` itself, autograd never DELIVERS these 6 x 6 gradients: grad-accumulator hooks (the
DDP reducer's), tensor hooks and `torch.autograd.grad` do not see them.  The batched form is therefore off unless the owner of
the gradients says it reads `.grad` after the backward and nothing else -- `eda_amd.parallel.FlatParams` does (it gathers
`.grad` into its flat buffer and all-reduces that), and calls `enable(True)`; a host that wraps the model in
`torch.nn.parallel.DistributedDataParallel` leaves it off and gets the per-layer autograd nodes.
A parameter with a registered tensor hook, or a frozen one among the six, also selects the per-layer nodes.
EDA_BATCHED_POSEMBED=0 switches it off regardless.
"""
import os

import torch
from torch.autograd import Function, Variable

from . import gemm, grouped
from .grouped import _weight_grads
from .heads_batched import _bn_bwd_multi, _dgrad_groups


_ENABLED = False


def enable(on=True):
    """Declare that parameter gradients are consumed from `.grad` after the backward (no autograd-delivery hooks)."""
    global _ENABLED
    _ENABLED = bool(on)


def enabled():
    return _ENABLED


class _Rec:
    __slots__ = ("mod", "x", "z1", "a1", "st1", "cfg", "dpos")


def _acc(p, g):
    if g is None:
        return
    g = g.view(p.shape)
    p.grad = g if p.grad is None else p.grad + g


class _LeafOut(Function):
    """out (computed without autograd) as a node of the graph: backward stores d(out) and schedules the batched backward."""

    @staticmethod
    def forward(ctx, out, hook_param, batch, rec):
        ctx.batch, ctx.rec = batch, rec
        return out.view_as(out)

    @staticmethod
    def backward(ctx, d):
        rec = ctx.rec
        rec.dpos = d.reshape(-1, d.shape[-1]).contiguous()
        ctx.batch._schedule()
        return None, None, None, None


class PosEmbedBatch:
    def __init__(self):
        self.recs = []
        self._queued = False

    @staticmethod
    def usable(mod, xyz):
        from . import _lib, sync_bn
        from .nn_utils import rows_ok
        if not _ENABLED or os.environ.get("EDA_BATCHED_POSEMBED", "1") == "0" or mod is None:
            return False
        head = mod.position_embedding_head
        bn = head[1]
        six = [head[0].weight, head[0].bias, bn.weight, bn.bias, head[3].weight, head[3].bias]
        if any(p is None or not p.requires_grad or getattr(p, "_backward_hooks", None) for p in six):
            return False
        R = xyz.shape[0] * xyz.shape[1]
        return (xyz.is_cuda and torch.is_grad_enabled() and bn.training and bn.track_running_stats and not sync_bn.diverts()
                and rows_ok(xyz, head[0].out_channels) and head[0].out_channels % 16 == 0
                and R <= _lib.lib().eda_bn_relu_dropout_max_rows())

    def add(self, mod, xyz):
        from .nn_utils import bump_batches_tracked
        head = mod.position_embedding_head
        B, N, C = xyz.shape
        r = _Rec()
        r.mod, r.dpos = mod, None
        with torch.no_grad():
            r.x = xyz.detach().reshape(B * N, C).contiguous()
            z1 = gemm.linear_fwd(r.x, head[0].weight.reshape(head[0].weight.shape[0], -1), head[0].bias)
            cfg = grouped.bn_relu_cfg(z1, [head[1]], None)
            r.z1, r.a1, r.st1 = grouped.bn_relu_fwd_raw(z1, cfg, [head[1].weight], [head[1].bias])
            bump_batches_tracked(head[1])
            out = gemm.linear_fwd(r.a1, head[3].weight.reshape(head[3].weight.shape[0], -1), head[3].bias)
        r.cfg = (cfg[0], cfg[1], bool(cfg[2]), float(cfg[5]), [int(s) & 0xFFFFFFFF for s in cfg[6]])
        self.recs.append(r)
        return _LeafOut.apply(out.view(B, N, -1), head[3].weight, self, r)

    def _schedule(self):
        if not self._queued:
            self._queued = True
            Variable._execution_engine.queue_callback(self._run)

    def _run(self):
        recs = [r for r in self.recs if r.dpos is not None]      # (the records live as long as the graph: a second backward
        self._queued = False                                     #  over a retained graph stores new gradients and comes here again)
        if not recs:
            return
        G = len(recs)
        heads = [r.mod.position_embedding_head for r in recs]
        dev = recs[0].x.device
        R, C = recs[0].z1.shape
        with torch.no_grad():
            W2 = [h[3].weight.reshape(h[3].weight.shape[0], -1) for h in heads]
            dpos = [r.dpos for r in recs]
            # second convolution: weight / bias gradients (deferred queue where it is active), input gradient of all layers
            dW2, db2 = _weight_grads(W2, [h[3].bias for h in heads], dpos, [r.a1 for r in recs], [True] * G,
                                     [h[3].bias is not None for h in heads])
            da1 = torch.empty((G, R, C), dtype=torch.float32, device=dev)
            _dgrad_groups(dpos, W2, [da1[g] for g in range(G)])
            # BatchNorm + ReLU
            dz1, dgb = _bn_bwd_multi([da1[g] for g in range(G)], [r.z1 for r in recs], [r.st1 for r in recs],
                                     [[h[1].weight] for h in heads], [r.cfg for r in recs])
            # first convolution (6 input channels): one batched product and one reduction for all layers
            dz1_all = torch.stack(dz1, 0)
            x_all = torch.stack([r.x for r in recs], 0)
            dW1 = torch.bmm(dz1_all.transpose(1, 2), x_all)                 # (G, 288, 6)
            db1 = dz1_all.sum(1)
            for r in recs:
                r.dpos = None
            for g, h in enumerate(heads):
                _acc(h[3].weight, dW2[g])
                if h[3].bias is not None:
                    _acc(h[3].bias, db2[g])
                _acc(h[1].weight, dgb[g][0]); _acc(h[1].bias, dgb[g][1])
                _acc(h[0].weight, dW1[g])
                if h[0].bias is not None:
                    _acc(h[0].bias, db1[g])
