"""Autograd wrappers of the channels-last set-abstraction kernels (csrc/sa_cl.hip).

Rows are positions; see the kernel file for the mapping to the reference's
QueryAndGroup / SharedMLP / max_pool2d chain.  GPU only (HIP library).
"""
import torch
from torch.autograd import Function

from . import _lib
from .ext import _timed
from .nn_utils import bump_batches_tracked


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("CPU not supported: the channels-last SA kernels run on the HIP library only")


class GroupConcatCL(Function):
    """[ (xyz[idx]-centre)*(1/r) | feats_cl[idx] ] rows: (B, m*ns, 3+C)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, feats_cl, idx, radius, normalize_xyz):
        _need_gpu(xyz)
        B, N, _ = xyz.shape
        m, ns = idx.shape[1], idx.shape[2]
        C = 0 if feats_cl is None else feats_cl.shape[2]
        xyz, new_xyz, idx = xyz.contiguous(), new_xyz.contiguous(), idx.contiguous()
        if feats_cl is not None:
            feats_cl = feats_cl.contiguous()
        out = torch.empty((B, m * ns, 3 + C), dtype=torch.float32, device=xyz.device)
        with torch.cuda.device(xyz.device), _timed('group_concat_cl', (B, N, m, ns, C)):
            rc = _lib.lib().eda_group_concat_cl_f32(
                xyz.data_ptr(), new_xyz.data_ptr(), feats_cl.data_ptr() if C else None, idx.data_ptr(),
                B, N, m, ns, C, float(radius), int(bool(normalize_xyz)), out.data_ptr(), _stream())
        _lib.check(rc, "eda_group_concat_cl_f32")
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, m, ns, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, N, m, ns, C = ctx.dims
        dfeats = None
        if C and ctx.needs_input_grad[2]:
            dout = dout.contiguous()
            dfeats = torch.empty((B, N, C), dtype=torch.float32, device=dout.device)
            with torch.cuda.device(dout.device), _timed('group_concat_cl_grad', (B, N, m, ns, C)):
                rc = _lib.lib().eda_group_concat_cl_grad_f32(dout.data_ptr(), idx.data_ptr(), B, N, m, ns,
                                                             C, dfeats.data_ptr(), _stream())
            _lib.check(rc, "eda_group_concat_cl_grad_f32")
        return None, None, dfeats, None, None, None


def _split_k(rows):
    """Number of K-slices for the weight-gradient GEMM (contraction over `rows` positions)."""
    for s in (256, 128, 64, 32, 16, 8, 4, 2):
        if rows % s == 0 and rows // s >= 512:
            return s
    return 1


class PointwiseLinearCL(Function):
    """Z = A W^T for rows A (R,Cin) and a 1x1-conv weight W (Cout,Cin,...).  The weight
    gradient contracts over up to 10^6 rows; it is computed split-K as a batched GEMM
    of partial (Cout,Cin) products (a single GEMM would run on a handful of CUs)."""

    @staticmethod
    def forward(ctx, a, weight):
        w2 = weight.reshape(weight.shape[0], -1)
        ctx.save_for_backward(a, weight)
        return a @ w2.t()

    @staticmethod
    def backward(ctx, dz):
        a, weight = ctx.saved_tensors
        w2 = weight.reshape(weight.shape[0], -1)
        da = dw = None
        dz = dz.contiguous()
        if ctx.needs_input_grad[0]:
            da = dz @ w2
        if ctx.needs_input_grad[1]:
            R = a.shape[0]
            s = _split_k(R)
            if s > 1:
                dw = torch.bmm(dz.view(s, R // s, -1).transpose(1, 2), a.view(s, R // s, -1)).sum(0)
            else:
                dw = dz.t() @ a
            dw = dw.view_as(weight)
        return da, dw


_bn_ws = {}


def _bn_workspace(dev):
    """The scratch of the BN FORWARD statistics kernels (2*1024 + 1 doubles: column sums + a ticket), one per
    (device, stream).  The kernels require it ZERO on entry and leave it zero (the last block finalises and cleans up),
    so it is allocated and zeroed once; launches on ONE stream are ordered, launches on different streams use different
    workspaces."""
    dev = torch.device(dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _bn_ws:
        # zeroed by a fill KERNEL on the current stream (torch.zeros is a memset: on ROCm 7.2 the
        # first BN launch right after it was observed to see stale data)
        _bn_ws[key] = torch.full((2 * 1024 + 2,), 0.0, dtype=torch.float64, device=dev)
    return _bn_ws[key]


class BNReLUCL(Function):
    """relu(batch_norm(z)) on rows z (R,C), optionally max-pooled over `pool` consecutive rows."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, eps, momentum, training, pool,
                p_drop=0.0, salt=0):
        """p_drop > 0: element dropout behind the ReLU inside the same kernel (small-row path
        only: pool == 1 and R <= eda_bn_relu_dropout_max_rows(); see nn_utils.bn_relu_rows)."""
        _need_gpu(z)
        z = z.contiguous()
        R, C = z.shape
        dev = z.device
        stats = torch.empty((4, C), dtype=torch.float32, device=dev)     # mean, rstd, scale, shift
        ws = _bn_workspace(dev)
        if pool > 1:
            out = torch.empty((R // pool, C), dtype=torch.float32, device=dev)
            argmax = torch.empty((R // pool, C), dtype=torch.uint8, device=dev)
        else:
            out = torch.empty((R, C), dtype=torch.float32, device=dev)
            argmax = None
        seed = None
        if p_drop > 0:
            from .attention import dropout_state
            seed = dropout_state(dev)
        with torch.cuda.device(dev), _timed('bn_relu_fwd', (R, C, pool, int(bool(training)))):
            rc = _lib.lib().eda_bn_relu_fwd_f32(
                z.data_ptr(), R, C, gamma.data_ptr(), beta.data_ptr(), float(eps), float(momentum),
                int(bool(training)), running_mean.data_ptr(), running_var.data_ptr(), int(pool),
                ws.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(),
                stats[3].data_ptr(), out.data_ptr(), argmax.data_ptr() if argmax is not None else None,
                float(p_drop), seed.data_ptr() if seed is not None else None, int(salt), _stream())
        _lib.check(rc, "eda_bn_relu_fwd_f32")
        ctx.save_for_backward(z, gamma, stats, argmax)
        ctx.cfg = (int(pool), bool(training), float(p_drop), int(salt))
        return out

    @staticmethod
    def backward(ctx, dout):
        z, gamma, stats, argmax = ctx.saved_tensors
        pool, training, p_drop, salt = ctx.cfg
        R, C = z.shape
        seed = None
        if p_drop > 0:
            from .attention import dropout_state
            seed = dropout_state(z.device)
        dout = dout.contiguous()
        dz = torch.empty_like(z)
        ws = torch.empty((2 * C,), dtype=torch.float64, device=z.device)     # zeroed by the library per call
        dgb = torch.empty((2, C), dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device), _timed('bn_relu_bwd', (R, C, pool, int(training))):
            rc = _lib.lib().eda_bn_relu_bwd_f32(
                dout.data_ptr(), argmax.data_ptr() if argmax is not None else None, z.data_ptr(), R, C,
                pool, gamma.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(),
                stats[3].data_ptr(), int(training), ws.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(),
                dz.data_ptr(), p_drop, seed.data_ptr() if seed is not None else None, salt, _stream())
        _lib.check(rc, "eda_bn_relu_bwd_f32")
        return dz, dgb[0], dgb[1], None, None, None, None, None, None, None, None


import ctypes
import os

_FUSED = os.environ.get("EDA_SA_FUSED", "1") != "0"


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


class FusedMLP(Function):
    """QueryAndGroup (or plain rows) -> L x [conv1x1, BatchNorm, ReLU] -> max over `pool` rows, as
    ONE native call per direction (eda_sa_fused_fwd/bwd_f32): per layer a single MFMA GEMM launch
    whose operand staging does the neighbourhood gather / the previous layer's BN+ReLU and whose
    epilogue produces the BatchNorm statistics.  Saved for the backward: the pre-activations z_l.

    cfg = dict(gather, radius, normalize_xyz, pool, training, eps, momentum, running=[(rm, rv), ...]);
    params = W0, gamma0, beta0, W1, gamma1, beta1, ...  (W_l may be (Cout, Cin, 1, 1))."""

    @staticmethod
    def forward(ctx, cfg, x_rows, xyz, new_xyz, feats_cl, idx, *params):
        L = len(params) // 3
        Ws = [params[3 * l].reshape(params[3 * l].shape[0], -1).contiguous() for l in range(L)]
        gammas = [params[3 * l + 1] for l in range(L)]
        betas = [params[3 * l + 2] for l in range(L)]
        gather = bool(cfg["gather"])
        if gather:
            _need_gpu(xyz)
            xyz, new_xyz, idx = xyz.contiguous(), new_xyz.contiguous(), idx.contiguous()
            B, N, _ = xyz.shape
            m, ns = idx.shape[1], idx.shape[2]
            C = 0 if feats_cl is None else feats_cl.shape[2]
            if feats_cl is not None:
                feats_cl = feats_cl.contiguous()
            R = B * m * ns
            dev = xyz.device
            c0 = 3 + C
        else:
            _need_gpu(x_rows)
            if x_rows.stride(1) != 1:
                x_rows = x_rows.contiguous()
            R, c0 = x_rows.shape
            B = N = m = ns = C = 0
            dev = x_rows.device
        chans = [c0] + [w.shape[0] for w in Ws]
        for l in range(L):
            assert Ws[l].shape[1] == chans[l], (Ws[l].shape, chans)
        pool = int(cfg["pool"])
        training = bool(cfg["training"])
        z = [torch.empty((R, chans[l + 1]), dtype=torch.float32, device=dev) for l in range(L)]
        stats = [torch.empty((4, chans[l + 1]), dtype=torch.float32, device=dev) for l in range(L)]
        out = torch.empty((R // pool, chans[L]), dtype=torch.float32, device=dev)
        argmax = torch.empty((R // pool, chans[L]), dtype=torch.uint8, device=dev) if pool > 1 else None
        running = cfg["running"]
        ws = _bn_workspace(dev)
        from . import sync_bn
        if sync_bn.fused_hook_installed():
            sync_bn.note_buffer(ws)              # (its column sums are all-reduced between the layers' kernels)
        chan_arr = (ctypes.c_int * (L + 1))(*chans)
        with torch.cuda.device(dev), _timed('sa_fused_fwd', (R, pool, int(training)) + tuple(chans)):
            rc = _lib.lib().eda_sa_fused_fwd_f32(
                x_rows.data_ptr() if not gather else None, x_rows.stride(0) if not gather else 0,
                xyz.data_ptr() if gather else None, new_xyz.data_ptr() if gather else None,
                feats_cl.data_ptr() if gather and C else None, idx.data_ptr() if gather else None,
                B, N, m, ns, C, float(cfg["radius"]) if gather else 1.0, int(bool(cfg["normalize_xyz"])) if gather else 0,
                R, L, chan_arr, _ptr_array(Ws), _ptr_array(gammas), _ptr_array(betas),
                _ptr_array([r[0] for r in running]), _ptr_array([r[1] for r in running]),
                float(cfg["eps"]), float(cfg["momentum"]), int(training), pool, _ptr_array(z), _ptr_array(stats),
                ws.data_ptr(), out.data_ptr(), argmax.data_ptr() if argmax is not None else None, _stream())
        _lib.check(rc, "eda_sa_fused_fwd_f32")
        ctx.save_for_backward(x_rows, xyz, new_xyz, feats_cl, idx, argmax, *Ws, *gammas, *z, *stats)
        ctx.cfg = (gather, float(cfg["radius"]) if gather else 1.0, bool(cfg["normalize_xyz"]) if gather else False,
                   pool, training, L, chans, (B, N, m, ns, C), [p.shape for p in params[0::3]])
        return out

    @staticmethod
    def backward(ctx, dout):
        gather, radius, normalize, pool, training, L, chans, (B, N, m, ns, C), wshapes = ctx.cfg
        sv = ctx.saved_tensors
        x_rows, xyz, new_xyz, feats_cl, idx, argmax = sv[:6]
        Ws, gammas = sv[6:6 + L], sv[6 + L:6 + 2 * L]
        z, stats = sv[6 + 2 * L:6 + 3 * L], sv[6 + 3 * L:6 + 4 * L]
        dev = dout.device
        dout = dout.contiguous()
        R = z[0].shape[0]
        cmax = max(chans[1:])
        sa = torch.empty((R, cmax), dtype=torch.float32, device=dev)
        sb = torch.empty((R, cmax), dtype=torch.float32, device=dev) if L > 1 else sa
        dW = [torch.empty((chans[l + 1], chans[l]), dtype=torch.float32, device=dev) for l in range(L)]
        dgb = [torch.empty((2, chans[l + 1]), dtype=torch.float32, device=dev) for l in range(L)]
        need_in = ctx.needs_input_grad[1] if not gather else (ctx.needs_input_grad[4] and C > 0)
        dx = dfeats = None
        if need_in and gather:
            dfeats = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        elif need_in:
            dx = torch.empty((R, chans[0]), dtype=torch.float32, device=dev)
        chan_arr = (ctypes.c_int * (L + 1))(*chans)
        lib = _lib.lib()
        ws_bytes = lib.eda_sa_fused_bwd_workspace_bytes(R, L, chan_arr, int(gather))
        ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
        from . import sync_bn
        hooked = sync_bn.fused_hook_installed()
        if hooked:
            sync_bn.note_buffer(ws)
        # W^T of the layers' weights, where the caller's shadow has them (eda_amd/wt_shadow.py): the call then skips
        # its per-layer transpose launches
        from . import wt_shadow
        wts = [None] * L
        if wt_shadow.active is not None:
            for l in range(L):
                v = wt_shadow.active.lookup(Ws[l])
                wts[l] = v if (v is not None and v.is_contiguous()) else None
        with torch.cuda.device(dev), _timed('sa_fused_bwd', (R, pool, int(training)) + tuple(chans)):
            rc = lib.eda_sa_fused_bwd_wt_f32(
                dout.data_ptr(), argmax.data_ptr() if argmax is not None else None,
                x_rows.data_ptr() if not gather else None, x_rows.stride(0) if not gather else 0,
                xyz.data_ptr() if gather else None, new_xyz.data_ptr() if gather else None,
                feats_cl.data_ptr() if gather and C else None, idx.data_ptr() if gather else None,
                B, N, m, ns, C, radius, int(normalize), R, L, chan_arr, _ptr_array(Ws), _ptr_array(wts), _ptr_array(gammas),
                _ptr_array(z), _ptr_array(stats), int(training), pool, sa.data_ptr(), sb.data_ptr(),
                ws.data_ptr(), ws_bytes, _ptr_array(dW), _ptr_array([g[0] for g in dgb]),
                _ptr_array([g[1] for g in dgb]), dx.data_ptr() if dx is not None else None,
                dx.stride(0) if dx is not None else 0, dfeats.data_ptr() if dfeats is not None else None, _stream())
        if hooked:
            sync_bn.forget_buffer(ws)
        _lib.check(rc, "eda_sa_fused_bwd_wt_f32")
        grads = []
        for l in range(L):
            grads += [dW[l].view(wshapes[l]), dgb[l][0], dgb[l][1]]
        return (None, dx, None, None, dfeats, None, *grads)


_ONE_PASS_EVAL = os.environ.get("EDA_SA_ONE_PASS_EVAL", "1") != "0"


def _one_pass_eval(cfg, xyz, new_xyz, feats_cl, idx, layers, bns, pool):
    """Inference (running statistics, no autograd): gather -> 3 x (conv1x1 + BN + ReLU) -> max over the neighbourhood in ONE
    launch that writes only the pooled output (csrc/sa_eval.hip; SURVEY section 8d's fused-layer byte count).  Returns None
    when the stack is not the one the kernel is built for (SA1: 3 feature channels, 6 -> 64 -> 64 -> 128)."""
    if feats_cl is None or not all(bn.track_running_stats for bn in bns):
        return None
    B, N, _ = xyz.shape
    m, ns = idx.shape[1], idx.shape[2]
    C = feats_cl.shape[2]
    Ws = [l.conv.weight.reshape(l.conv.weight.shape[0], -1).contiguous() for l in layers]
    chans = [3 + C] + [w.shape[0] for w in Ws]
    chan_arr = (ctypes.c_int * len(chans))(*chans)
    lib = _lib.lib()
    if pool != ns or not lib.eda_sa_fused_eval_supported(C, len(layers), chan_arr, ns):
        return None
    xyz, new_xyz, feats_cl, idx = xyz.contiguous(), new_xyz.contiguous(), feats_cl.contiguous(), idx.contiguous()
    out = torch.empty((B * m, chans[-1]), dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device), _timed('sa_fused_eval', (B * m * ns, pool) + tuple(chans)):
        rc = lib.eda_sa_fused_eval_f32(
            xyz.data_ptr(), new_xyz.data_ptr(), feats_cl.data_ptr(), idx.data_ptr(), B, N, m, ns, C, float(cfg["radius"]),
            int(bool(cfg["normalize_xyz"])), len(layers), chan_arr, _ptr_array(Ws), _ptr_array([bn.weight for bn in bns]),
            _ptr_array([bn.bias for bn in bns]), _ptr_array([bn.running_mean for bn in bns]),
            _ptr_array([bn.running_var for bn in bns]), float(cfg["eps"]), out.data_ptr(), _stream())
    _lib.check(rc, "eda_sa_fused_eval_f32")
    return out


def _fusable(layers):
    from . import sync_bn
    if sync_bn.enabled() and not sync_bn.fused_hook_installed():
        return False               # global-batch statistics without the library hook: layer by layer (eda_amd/sync_bn.py)
    return _FUSED and all(l.bn is not None and l.conv.bias is None and l.conv.out_channels % 4 == 0 for l in layers)


def fused_mlp(mlp, pool, x_rows=None, xyz=None, new_xyz=None, feats_cl=None, idx=None, radius=1.0,
              normalize_xyz=False):
    """Run SharedMLP `mlp` through FusedMLP on plain rows (x_rows) or gathered neighbourhoods."""
    layers = mlp.layers()
    bns = [layer.bn.bn for layer in layers]
    training = bns[0].training
    assert all(bn.training == training for bn in bns)
    assert all(bn.eps == bns[0].eps and bn.momentum == bns[0].momentum for bn in bns)
    cfg = dict(gather=idx is not None, radius=radius, normalize_xyz=normalize_xyz, pool=pool,
               training=training or not bns[0].track_running_stats, eps=bns[0].eps, momentum=bns[0].momentum,
               running=[(bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None) for bn in bns])
    params = []
    for layer, bn in zip(layers, bns):
        params += [layer.conv.weight, bn.weight, bn.bias]
    if idx is not None and not cfg["training"] and not torch.is_grad_enabled() and _ONE_PASS_EVAL:
        out = _one_pass_eval(cfg, xyz, new_xyz, feats_cl, idx, layers, bns, pool)
        if out is not None:
            return out
    out = FusedMLP.apply(cfg, x_rows, xyz, new_xyz, feats_cl, idx, *params)
    if training:
        for bn in bns:
            if bn.track_running_stats:
                bump_batches_tracked(bn)
    return out


def shared_mlp_rows(mlp, rows, pool):
    """Run a SharedMLP (conv1x1 -> BN -> ReLU stack) on rows (R,Cin); the LAST layer's
    BN+ReLU is fused with a max over each `pool` consecutive rows.  Returns (R/pool, Cout)."""
    layers = mlp.layers()
    if _fusable(layers) and rows.is_cuda:
        return fused_mlp(mlp, pool, x_rows=rows)
    x = rows
    from . import sync_bn
    from .nn_utils import linear_rows
    for i, layer in enumerate(layers):
        if sync_bn.diverts():
            z = linear_rows(x, layer.conv.weight.reshape(layer.conv.weight.shape[0], -1), None)
            bn = layer.bn.bn
            x = sync_bn.bn_relu(bn, z, pool if i == len(layers) - 1 else 1)
            if bn.training and bn.track_running_stats:
                bump_batches_tracked(bn)
            continue
        z = PointwiseLinearCL.apply(x, layer.conv.weight)
        bn = layer.bn.bn
        last = i == len(layers) - 1
        training = bn.training
        x = BNReLUCL.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                           bn.momentum, training, pool if last else 1)
        if training and bn.track_running_stats:
            bump_batches_tracked(bn)
    return x
