"""The backward passes of ALL prediction heads of a step as one set of launches.

A head's outputs reach the next decoder layer detached (models/bdetr.py:262-264, 316-320: `base_xyz.detach()`), so its
backward depends on the loss alone: the seven heads' (proposal + six decoder layers) backward chains -- per head two
BatchNorm+ReLU+Dropout backwards, three input-gradient products, ~11 launches of 18-192 workgroups -- are independent of
each other and of the decoder's backward until they meet a layer's output gradient.  The forward cannot be batched (layer
i + 1's positional term needs head i's boxes), so it runs where it always did, but WITHOUT autograd, keeping its
intermediates; after the decoder loop ONE autograd node takes the seven feature tensors and hands out the recorded outputs,
and its backward issues every stage for all heads together: grouped input gradients over 3 x 7 = 21 problems
(include/eda_hip.h: eda_linear_grouped_dgrad_f32), the grouped BatchNorm backward of seven matrices in one launch
(eda_bn_relu_grouped_bwd_multi_f32); weight / bias gradients go to the deferred queue as before.  Same kernels on the same
values as the per-head path (eda_amd/grouped.py): results are bitwise equal; EDA_BATCHED_HEADS=0 keeps the per-head nodes.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib, gemm, grouped
from .ext import _timed
from .grouped import _grouped_dgrad, _packed_view, _parr, _stream, _weight_grads

_MAXG = 24          # csrc/gemm.h G_MAXGROUPS
_MAXM = 8           # csrc/sa_cl.hip BN_MAXMAT


class _Record:
    """Intermediates of one head's forward (rows x, packed z1 / a1 / z2 / a2, per-stack outputs, BatchNorm statistics)."""
    __slots__ = ("x", "xd", "z1", "a1", "z2", "a2", "outs", "st1", "st2", "cfg1", "cfg2", "nets", "names", "center")


def _bn_bwd_multi(douts, zs, stats, gammas, cfgs):
    """BatchNorm+ReLU+Dropout backward of len(zs) packed matrices of one shape: [(dz, dgb (2, G*C))]."""
    G, C, training, p_drop, _ = cfgs[0]
    dev = zs[0].device
    R = zs[0].shape[0]
    dzs = [torch.empty_like(z) for z in zs]
    dgbs = [torch.empty((2, G * C), dtype=torch.float32, device=dev) for _ in zs]
    seed = None
    if p_drop > 0:
        from .attention import dropout_state
        seed = dropout_state(dev)
    for m0 in range(0, len(zs), _MAXM):
        sl = slice(m0, m0 + _MAXM)
        nm = len(zs[sl])
        salts = [s for c in cfgs[sl] for s in c[4]]
        salt_arr = (ctypes.c_uint * len(salts))(*salts)
        gs = [g for gl in gammas[sl] for g in gl]
        with torch.cuda.device(dev), _timed("bn_relu_grouped_bwd_multi", (nm, R, G, C, int(training))):
            rc = _lib.lib().eda_bn_relu_grouped_bwd_multi_f32(
                nm, _parr(douts[sl]), _parr(zs[sl]), R, G, C, _parr(gs), _parr(stats[sl]), int(training), _parr(dgbs[sl]),
                _parr(dzs[sl]), float(p_drop), seed.data_ptr() if seed is not None else None, salt_arr, _stream())
        _lib.check(rc, "eda_bn_relu_grouped_bwd_multi_f32")
    return dzs, dgbs


def _dgrad_groups(dys, Ws, dxs):
    """dx_g = dy_g W_g for any number of groups: grouped launches of <= 24 problems whose widths allow 16-byte rows, the
    element-wise kernel one by one for the others (1- / 3-wide outputs)."""
    fast = [g for g in range(len(Ws)) if Ws[g].shape[0] % 4 == 0]
    slow = [g for g in range(len(Ws)) if Ws[g].shape[0] % 4 != 0]
    for i in range(0, len(fast), _MAXG):
        idx = fast[i:i + _MAXG]
        _grouped_dgrad([dys[g] for g in idx], [Ws[g] for g in idx], [dxs[g] for g in idx])
    for g in slow:
        gemm.linear_dgrad(dys[g], Ws[g], out=dxs[g])


def _tiny_out_bwd(dys, Ws, a_blocks, da_blocks, bs):
    """Layers with <= 4 output channels, all of one shape: da = dy W, dW = dy^T a, db = colsum(dy) for every layer in two
    launches (include/eda_hip.h: eda_tiny_out_bwd_multi_f32).  Returns ([dW], [db])."""
    from .grouped import _larr
    G = len(Ws)
    MW, C = Ws[0].shape
    R = dys[0].shape[0]
    dev = dys[0].device
    dWs = [torch.empty((MW, C), dtype=torch.float32, device=dev) for _ in range(G)]
    dbs = [torch.empty((MW,), dtype=torch.float32, device=dev) if b is not None else None for b in bs]
    L = _lib.lib()
    ws = torch.empty((L.eda_tiny_out_bwd_workspace_bytes(G, C, MW) // 4,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _timed("tiny_out_bwd_multi", (G, R, C, MW)):
        rc = L.eda_tiny_out_bwd_multi_f32(G, R, C, MW, _parr(dys), _parr(Ws), _parr(a_blocks), _larr([a.stride(0) for a in a_blocks]),
                                          _parr(da_blocks), _larr([d.stride(0) for d in da_blocks]), _parr(dWs), _parr(dbs),
                                          ws.data_ptr(), ws.numel() * 4, _stream())
    _lib.check(rc, "eda_tiny_out_bwd_multi_f32")
    return dWs, dbs


class _HeadsBatched(Function):
    @staticmethod
    def forward(ctx, recs, *tensors):
        H, S = len(recs), len(recs[0].nets)
        ctx.recs = recs
        ctx.shape = (H, S)
        outs = []
        for r in recs:
            # the centre stack hands out base_xyz + residual as computed in add() (base_xyz carries no gradient: the
            # centre's gradient IS the residual's) -- not the residual for a second, recorded addition per head
            outs += [(r.center if (r.center is not None and n == "center_residual_head") else o).view_as(o)
                     for n, o in zip(r.names, r.outs)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        recs = ctx.recs
        H, S = ctx.shape
        R = recs[0].xd.shape[0]
        C = recs[0].z1.shape[1] // S
        dev = recs[0].xd.device
        W1 = [[n[0].weight.reshape(n[0].weight.shape[0], -1) for n in r.nets] for r in recs]
        W2 = [[n[4].weight.reshape(n[4].weight.shape[0], -1) for n in r.nets] for r in recs]
        W3 = [[n[8].weight.reshape(n[8].weight.shape[0], -1) for n in r.nets] for r in recs]
        b3 = [[n[8].bias for n in r.nets] for r in recs]
        # ---- layer 3: d(a2) = d(out) W3, packed per head
        dys3 = []
        for h in range(H):
            for s in range(S):
                d = douts[h * S + s]
                if d is None:
                    d = torch.full((R, W3[h][s].shape[0]), 0.0, device=dev)
                dys3.append(d if d.stride(1) == 1 and (d.shape[0] <= 1 or d.stride(0) == d.shape[1]) else d.contiguous())
        da2 = [torch.empty((R, S * C), dtype=torch.float32, device=dev) for _ in range(H)]
        flatW3 = [w for ws in W3 for w in ws]
        flatb3 = [b for bs in b3 for b in bs]
        da2b = [da2[h][:, s * C:(s + 1) * C] for h in range(H) for s in range(S)]
        a2b = [recs[h].a2[:, s * C:(s + 1) * C] for h in range(H) for s in range(S)]
        dW3, db3 = [None] * (H * S), [None] * (H * S)
        # the 1-4-channel outputs (box centre / size) of all heads: input, weight and bias gradients in one pass
        tiny = [g for g in range(H * S) if flatW3[g].shape[0] <= 4 and flatW3[g].is_contiguous() and dys3[g].is_contiguous()]
        if len(tiny) > 1 and len({flatW3[g].shape for g in tiny}) == 1 and len(tiny) <= 16 and C % 4 == 0:
            tw, tb = _tiny_out_bwd([dys3[g] for g in tiny], [flatW3[g] for g in tiny], [a2b[g] for g in tiny],
                                   [da2b[g] for g in tiny], [flatb3[g] for g in tiny])
            for k, g in enumerate(tiny):
                dW3[g], db3[g] = tw[k], tb[k]
        else:
            tiny = []
        rest = [g for g in range(H * S) if g not in tiny]
        _dgrad_groups([dys3[g] for g in rest], [flatW3[g] for g in rest], [da2b[g] for g in rest])
        rw, rb = _weight_grads([flatW3[g] for g in rest], [flatb3[g] for g in rest], [dys3[g] for g in rest],
                               [a2b[g] for g in rest], [True] * len(rest), [True] * len(rest))
        for k, g in enumerate(rest):
            dW3[g], db3[g] = rw[k], rb[k]
        # ---- BatchNorm 2 (+ ReLU, Dropout)
        g2 = [[n[5].weight for n in r.nets] for r in recs]
        dz2, dgb2 = _bn_bwd_multi(da2, [r.z2 for r in recs], [r.st2 for r in recs], g2, [r.cfg2 for r in recs])
        # ---- layer 2: d(a1) blocks = d(z2) blocks W2
        da1 = [torch.empty((R, S * C), dtype=torch.float32, device=dev) for _ in range(H)]
        dz2b = [dz2[h][:, s * C:(s + 1) * C] for h in range(H) for s in range(S)]
        flatW2 = [w for ws in W2 for w in ws]
        _dgrad_groups(dz2b, flatW2, [da1[h][:, s * C:(s + 1) * C] for h in range(H) for s in range(S)])
        a1b = [recs[h].a1[:, s * C:(s + 1) * C] for h in range(H) for s in range(S)]
        dW2, _ = _weight_grads(flatW2, [None] * (H * S), dz2b, a1b, [True] * (H * S), [False] * (H * S))
        # ---- BatchNorm 1
        g1 = [[n[1].weight for n in r.nets] for r in recs]
        dz1, dgb1 = _bn_bwd_multi(da1, [r.z1 for r in recs], [r.st1 for r in recs], g1, [r.cfg1 for r in recs])
        # ---- layer 1: d(x) = d(z1) [W1_0; W1_1; ..] (the siblings' weights lie back to back: one product per head)
        dxs = [None] * H
        need_x = [ctx.needs_input_grad[1 + h] for h in range(H)]
        packed = [_packed_view(W1[h]) for h in range(H)]
        idx = [h for h in range(H) if need_x[h]]
        if idx and all(packed[h] is not None for h in idx):
            for h in idx:
                dxs[h] = torch.empty((R, W1[h][0].shape[1]), dtype=torch.float32, device=dev)
            _dgrad_groups([dz1[h] for h in idx], [packed[h] for h in idx], [dxs[h] for h in idx])
        else:
            for h in idx:
                acc = None
                for s in range(S):
                    t = gemm.linear_dgrad(dz1[h][:, s * C:(s + 1) * C], W1[h][s])
                    acc = t if acc is None else acc + t
                dxs[h] = acc
        dz1b = [dz1[h][:, s * C:(s + 1) * C] for h in range(H) for s in range(S)]
        flatW1 = [w for ws in W1 for w in ws]
        dW1, _ = _weight_grads(flatW1, [None] * (H * S), dz1b, [recs[h].xd for h in range(H) for _ in range(S)],
                               [True] * (H * S), [False] * (H * S))
        # ---- gradients in the order of forward()'s tensors: xs, then per head W1, gamma1, beta1, W2, gamma2, beta2, W3, b3
        grads = [None] + [dxs[h] for h in range(H)]
        for h in range(H):
            nets = recs[h].nets
            k = h * S
            grads += [dW1[k + s].view(nets[s][0].weight.shape) if dW1[k + s] is not None else None for s in range(S)]
            grads += [dgb1[h][0][s * C:(s + 1) * C] for s in range(S)] + [dgb1[h][1][s * C:(s + 1) * C] for s in range(S)]
            grads += [dW2[k + s].view(nets[s][4].weight.shape) if dW2[k + s] is not None else None for s in range(S)]
            grads += [dgb2[h][0][s * C:(s + 1) * C] for s in range(S)] + [dgb2[h][1][s * C:(s + 1) * C] for s in range(S)]
            grads += [dW3[k + s].view(nets[s][8].weight.shape) if dW3[k + s] is not None else None for s in range(S)]
            grads += [db3[k + s] for s in range(S)]
        return tuple(grads)


class HeadsBatch:
    """Collects the heads of one forward pass: add() runs a head's five forward launches without autograd and keeps what
    its backward needs; finalize() creates the one autograd node and fills `end_points` with its outputs."""

    def __init__(self):
        self.recs, self.meta = [], []
        self.last_query_pos = None          # cat([center, size]) of the head added last (values; None: not formed)

    @staticmethod
    def usable(heads, rows):
        if os.environ.get("EDA_BATCHED_HEADS", "1") == "0" or not (rows.is_cuda and torch.is_grad_enabled()):
            return False
        names = heads[0].sibling_stacks()
        return (len(heads) <= _MAXM and len(names) * len(heads) <= _MAXG and
                all((not h.heading) and h.training and h.sibling_stacks() == names and h._grouped_ok(rows) for h in heads))

    def add(self, head, rows, base_xyz, end_points, prefix, B, Q):
        from .nn_utils import bump_batches_tracked
        names = head.sibling_stacks()
        nets = [getattr(head, n).net for n in names]
        r = _Record()
        r.nets, r.names, r.x = nets, names, rows
        with torch.no_grad():
            r.xd = rows.detach()
            r.z1 = grouped.shared_in_linear(r.xd, [n[0].weight.squeeze(-1) for n in nets])
            cfg1 = grouped.bn_relu_cfg(r.z1, [n[1] for n in nets], [n[3] for n in nets])
            r.z1, r.a1, r.st1 = grouped.bn_relu_fwd_raw(r.z1, cfg1, [n[1].weight for n in nets], [n[1].bias for n in nets])
            r.z2 = grouped.block_linear(r.a1, [n[4].weight.squeeze(-1) for n in nets], [None] * len(nets), pack_out=True)
            cfg2 = grouped.bn_relu_cfg(r.z2, [n[5] for n in nets], [n[7] for n in nets])
            r.z2, r.a2, r.st2 = grouped.bn_relu_fwd_raw(r.z2, cfg2, [n[5].weight for n in nets], [n[5].bias for n in nets])
            r.outs = list(grouped.block_linear(r.a2, [n[8].weight.squeeze(-1) for n in nets], [n[8].bias for n in nets],
                                               pack_out=False))
            for n in nets:
                for bn in (n[1], n[5]):
                    if bn.training and bn.track_running_stats:
                        bump_batches_tracked(bn)
        r.cfg1 = (cfg1[0], cfg1[1], bool(cfg1[2]), float(cfg1[5]), [int(s) & 0xFFFFFFFF for s in cfg1[6]])
        r.cfg2 = (cfg2[0], cfg2[1], bool(cfg2[2]), float(cfg2[5]), [int(s) & 0xFFFFFFFF for s in cfg2[6]])
        self.recs.append(r)
        self.meta.append((head, base_xyz, end_points, prefix, B, Q))
        o = dict(zip(names, r.outs))
        res, size = o["center_residual_head"], o["size_pred_head"]
        self.last_query_pos = None
        if (not base_xyz.requires_grad and base_xyz.is_contiguous() and res.is_contiguous() and size.is_contiguous()
                and base_xyz.dtype == res.dtype == size.dtype == torch.float32):
            # center = base_xyz + residual AND the next decoder layer's position input cat([center, size]) (bdetr.py:300-308;
            # both are values only here: the differentiable outputs are formed in finalize()) in one launch
            from . import _lib
            center = torch.empty((B, Q, 3), dtype=torch.float32, device=res.device)
            qpos = torch.empty((B, Q, 6), dtype=torch.float32, device=res.device)
            with torch.cuda.device(res.device):
                rc = _lib.lib().eda_center_query_pos_f32(base_xyz.data_ptr(), res.data_ptr(), size.data_ptr(), B * Q, center.data_ptr(),
                                                         qpos.data_ptr(), torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "eda_center_query_pos_f32")
            self.last_query_pos = qpos
        else:
            center = base_xyz + res.view(B, Q, 3)
        # (kept for finalize(): the recorded output of the centre stack, unless base_xyz itself is differentiable)
        r.center = center.view(B * Q, 3) if not base_xyz.requires_grad else None
        return center, size.view(B, Q, 3)

    def finalize(self):
        if not self.recs:
            return
        tensors = [r.x for r in self.recs]
        for r in self.recs:
            nets = r.nets
            tensors += [n[0].weight for n in nets] + [n[1].weight for n in nets] + [n[1].bias for n in nets]
            tensors += [n[4].weight for n in nets] + [n[5].weight for n in nets] + [n[5].bias for n in nets]
            tensors += [n[8].weight for n in nets] + [n[8].bias for n in nets]
        outs = _HeadsBatched.apply(self.recs, *tensors)
        S = len(self.recs[0].nets)
        for h, (head, base_xyz, end_points, prefix, B, Q) in enumerate(self.meta):
            o = dict(zip(self.recs[h].names, outs[h * S:(h + 1) * S]))
            if head.objectness:
                end_points[f"{prefix}objectness_scores"] = o["objectness_scores_head"].view(B, Q)
            end_points[f"{prefix}base_xyz"] = base_xyz
            end_points[f"{prefix}center"] = (o["center_residual_head"].view(B, Q, 3) if self.recs[h].center is not None
                                             else base_xyz + o["center_residual_head"].view(B, Q, 3))
            end_points[f"{prefix}pred_size"] = o["size_pred_head"].view(B, Q, 3)
            if head.compute_sem_scores:
                end_points[f"{prefix}sem_cls_scores"] = o["sem_cls_scores_head"].view(B, Q, -1)
        self.recs, self.meta = [], []
