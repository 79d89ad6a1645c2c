"""Row GEMMs of the pointwise layers on the repo's own fp32-MFMA kernels (csrc/gemm.hip).

``linear_fwd`` / ``linear_dgrad`` are thin ctypes calls; shapes follow ``F.linear``:
x (R,K), w (N,K), y (R,N).  Row-strided 2-D views are accepted (unit column stride).
"""
import torch

from . import _lib
from .ext import _timed


def _rows2d(t):
    if t.dim() != 2 or t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


_SK_BYTES = 16 << 20
_sk_ws_cache = {}


def _stream_workspace(dev):
    """The persistent 16 MiB scratch of (device, current stream).  Its leading ticket words must be ZERO before a split
    launch and every launch leaves them zero.  It is created OUTSIDE any capture: a buffer first allocated while a hipGraph
    is being captured would live in that graph's private pool and be zeroed by a captured fill -- i.e. only when THAT graph
    replays -- while this cache hands it to eager calls and other graphs as well."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _sk_ws_cache.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("eda_amd.gemm: no split-launch workspace for this (device, stream) yet -- run the step once "
                               "eagerly on the stream (or call eda_amd.gemm.prepare_workspace()) before capturing it")
        ws = _sk_ws_cache[key] = torch.zeros(_SK_BYTES // 4, dtype=torch.int32, device=dev)
    return ws


def prepare_workspace(dev=None):
    """Create the current stream's workspace (outside a capture); returns it."""
    dev = torch.device("cuda", torch.cuda.current_device()) if dev is None else torch.device(dev)
    return _stream_workspace(dev)


def reset_workspaces():
    """Re-zero the ticket words of every workspace handed out so far (after a failed or aborted launch left an arrival
    count behind, every later split product on that stream would merge at the wrong arrival).  Synchronises."""
    torch.cuda.synchronize()
    for ws in _sk_ws_cache.values():
        ws.zero_()
    torch.cuda.synchronize()


def splitk_workspace(dev, R, K, N):
    """Scratch of the split-contraction launches (include/eda_hip.h: eda_linear_ex_ws_f32) for a product with R rows,
    contraction K, N columns, or None when the library does not split that shape: one persistent 16 MiB buffer per
    (device, stream) -- its leading ticket words start at zero and every call leaves them zero."""
    n = int(_lib.lib().eda_linear_splitk_workspace_bytes(R, K, N))
    if n == 0 or n > _SK_BYTES:
        return None
    return _stream_workspace(dev)


def workspace(dev, nbytes):
    """The persistent zero-initialised scratch of the current stream (the one splitk_workspace hands out) if `nbytes` fit."""
    if nbytes > _SK_BYTES:
        return None
    return _stream_workspace(dev)


def linear_fwd(x2, w, bias=None, relu=False, out=None):
    """y = x2 @ w.T (+ bias) (ReLU) for fp32 GPU matrices."""
    x2, w = _rows2d(x2), _rows2d(w)
    R, K = x2.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty((R, N), dtype=torch.float32, device=x2.device)
    if R == 0:
        return out
    with torch.cuda.device(x2.device), _timed("gemm_fwd", (R, K, N)):
        ws = splitk_workspace(x2.device, R, K, N)
        if ws is not None:
            rc = _lib.lib().eda_linear_ex_ws_f32(x2.data_ptr(), _ld(x2), R, K, w.data_ptr(), _ld(w), N,
                                                 bias.data_ptr() if bias is not None else None, int(relu), 0.0, None, 0,
                                                 None, 0, 1.0, out.data_ptr(), _ld(out), ws.data_ptr(), ws.numel() * 4,
                                                 torch.cuda.current_stream().cuda_stream)
        else:
            rc = _lib.lib().eda_linear_fwd_f32(x2.data_ptr(), _ld(x2), R, K, w.data_ptr(), _ld(w), N,
                                               bias.data_ptr() if bias is not None else None, int(relu),
                                               out.data_ptr(), _ld(out), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_linear_fwd_f32")
    return out


def linear_ex(x2, w, bias=None, relu=False, drop=None, gate=None, out=None):
    """linear_fwd with Dropout and / or a ReLU-backward gate in the epilogue (include/eda_hip.h: eda_linear_ex_f32).
    drop = (p, seed tensor (1,) int64 on the device, salt); gate = (activated output (R,N) of the layer the result
    flows into, scale)."""
    x2, w = _rows2d(x2), _rows2d(w)
    R, K = x2.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty((R, N), dtype=torch.float32, device=x2.device)
    if R == 0:
        return out
    p, seed, salt = drop if drop is not None else (0.0, None, 0)
    g, gscale = gate if gate is not None else (None, 1.0)
    if g is not None:
        g = _rows2d(g)
        assert g.shape == (R, N)
    with torch.cuda.device(x2.device), _timed("gemm_fwd", (R, K, N)):
        ws = splitk_workspace(x2.device, R, K, N)
        rc = _lib.lib().eda_linear_ex_ws_f32(x2.data_ptr(), _ld(x2), R, K, w.data_ptr(), _ld(w), N,
                                             bias.data_ptr() if bias is not None else None, int(relu),
                                             float(p), seed.data_ptr() if (seed is not None and p > 0) else None,
                                             int(salt) & 0xFFFFFFFF, g.data_ptr() if g is not None else None,
                                             _ld(g) if g is not None else 0, float(gscale),
                                             out.data_ptr(), _ld(out), ws.data_ptr() if ws is not None else None,
                                             ws.numel() * 4 if ws is not None else 0, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_linear_ex_ws_f32")
    return out


def linear_dgrad_gated(dy2, w, gate, scale, out=None):
    """dx = (dy2 @ w) * (gate > 0) * scale: the input gradient of a linear layer whose INPUT was relu(.) (* dropout
    mask / keep probability) = `gate`.  One launch when W^T is at hand (wt_shadow), two otherwise."""
    from . import wt_shadow
    if wt_shadow.active is not None:
        wt = wt_shadow.active.lookup(w)
        if wt is not None:
            return linear_ex(dy2, wt, gate=(gate, scale), out=out)
    dx = linear_dgrad(dy2, w, out=out)
    return dx.mul_(scale).mul_(gate > 0) if scale != 1.0 else dx.mul_(gate > 0)


def _addend_ok(addend, R, C):
    return (addend.dtype == torch.float32 and addend.is_cuda and addend.dim() == 2 and tuple(addend.shape) == (R, C)
            and addend.stride(1) == 1 and addend.stride(0) >= C)


def linear_addend(x2, w, addend, bias=None, out=None):
    """y = x2 @ w.T (+ bias) + addend in one launch (include/eda_hip.h: eda_linear_addend_ws_f32); out may be `addend`."""
    x2, w, addend = _rows2d(x2), _rows2d(w), _rows2d(addend)
    R, K = x2.shape
    N = w.shape[0]
    assert w.shape[1] == K and _addend_ok(addend, R, N)
    if out is None:
        out = torch.empty((R, N), dtype=torch.float32, device=x2.device)
    if R == 0:
        return out
    with torch.cuda.device(x2.device), _timed("gemm_fwd", (R, K, N)):
        ws = splitk_workspace(x2.device, R, K, N)
        rc = _lib.lib().eda_linear_addend_ws_f32(x2.data_ptr(), _ld(x2), R, K, w.data_ptr(), _ld(w), N,
                                                 bias.data_ptr() if bias is not None else None, addend.data_ptr(),
                                                 _ld(addend), out.data_ptr(), _ld(out),
                                                 ws.data_ptr() if ws is not None else None,
                                                 ws.numel() * 4 if ws is not None else 0,
                                                 torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_linear_addend_ws_f32")
    return out


def linear_dgrad(dy2, w, out=None, addend=None):
    """dx = dy2 @ w (+ addend, in the same launch) for fp32 GPU matrices dy2 (R,N), w (N,K), addend (R,K)."""
    from . import wt_shadow
    if wt_shadow.active is not None:
        wt = wt_shadow.active.lookup(w)
        if wt is not None:                       # W^T is at hand: the forward's GEMM form on it
            return linear_fwd(dy2, wt, out=out) if addend is None else linear_addend(dy2, wt, addend, out=out)
    dy2, w = _rows2d(dy2), _rows2d(w)
    R, N = dy2.shape
    K = w.shape[1]
    assert w.shape[0] == N
    if out is None:
        out = torch.empty((R, K), dtype=torch.float32, device=dy2.device)
    if R == 0:
        return out
    if addend is not None:
        addend = _rows2d(addend)
        assert _addend_ok(addend, R, K)
        with torch.cuda.device(dy2.device), _timed("gemm_dgrad", (R, N, K)):
            ws = splitk_workspace(dy2.device, R, N, K)
            rc = _lib.lib().eda_linear_dgrad_addend_ws_f32(dy2.data_ptr(), _ld(dy2), R, N, w.data_ptr(), _ld(w), K,
                                                           addend.data_ptr(), _ld(addend), out.data_ptr(), _ld(out),
                                                           ws.data_ptr() if ws is not None else None,
                                                           ws.numel() * 4 if ws is not None else 0,
                                                           torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "eda_linear_dgrad_addend_ws_f32")
        return out
    with torch.cuda.device(dy2.device), _timed("gemm_dgrad", (R, N, K)):
        ws = splitk_workspace(dy2.device, R, N, K)
        rc = _lib.lib().eda_linear_dgrad_ws_f32(dy2.data_ptr(), _ld(dy2), R, N, w.data_ptr(), _ld(w), K,
                                                out.data_ptr(), _ld(out), ws.data_ptr() if ws is not None else None,
                                                ws.numel() * 4 if ws is not None else 0,
                                                torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_linear_dgrad_ws_f32")
    return out


def frozen_planes(w):
    """The three bf16 planes (h | m | l, v = h + m + l exactly) of a FROZEN (N, K) fp32 weight, in the layout
    `linear_frozen` consumes: (3, N, K) int16 storage (include/eda_hip.h: eda_bf16x3_split_f32).  Built once per weight."""
    w = _rows2d(w.detach())
    N, K = w.shape
    planes = torch.empty((3, N, K), dtype=torch.int16, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.lib().eda_bf16x3_split_f32(w.data_ptr(), _ld(w), N, K, planes.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_bf16x3_split_f32")
    return planes


def linear_frozen_supported(R, K, N):
    return bool(_lib.lib().eda_linear_frozen_b3_supported(R, K, N))


def linear_frozen(x2, planes, bias=None, act=0, out=None):
    """act(x2 @ W.T + bias) for the frozen weight whose planes `frozen_planes` made: bf16 x 3 on v_mfma_f32_16x16x32_bf16,
    fp32 accumulation, fp32 accuracy (csrc/gemm_frozen.hip).  act: 0 none, 1 ReLU, 2 GELU (erf).  Inference only."""
    x2 = _rows2d(x2)
    R, K = x2.shape
    N = planes.shape[1]
    assert planes.shape[2] == K
    if out is None:
        out = torch.empty((R, N), dtype=torch.float32, device=x2.device)
    if R == 0:
        return out
    with torch.cuda.device(x2.device), _timed("gemm_frozen_b3", (R, K, N)):
        rc = _lib.lib().eda_linear_frozen_b3_f32(x2.data_ptr(), _ld(x2), R, K, planes.data_ptr(), N,
                                                 bias.data_ptr() if bias is not None else None, int(act), out.data_ptr(), _ld(out),
                                                 torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_linear_frozen_b3_f32")
    return out
