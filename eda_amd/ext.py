"""Drop-in for the reference's ``pointnet2._ext`` pybind11 module.

The reference binds nine functions (pointnet2/_ext_src/src/bindings.cpp:11-24);
this module exposes the same nine callables -- same positional signatures, same
argument checks and messages (include/utils.h:10-30), same allocation behaviour
(fresh, callee-owned outputs on the input's device) -- on top of the C ABI of
``libeda_hip.so`` (include/eda_hip.h).  Work is enqueued on torch's current HIP
stream and never synchronises, like the reference.

``install_as_pointnet2_ext()`` registers it as ``sys.modules['pointnet2._ext']``
so the reference's own ``pointnet2_utils.py`` (which does ``import
pointnet2._ext as _ext``, pointnet2_utils.py:25-33) runs unmodified on MI355X.

CPU tensors raise "CPU not supported" exactly like the reference
(sampling.cpp:39,65,87 ...): there is no CPU path in the product.
"""
import sys
import types

import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Optional per-op timing (bench.py): when `op_timer` is set, every native call is
# bracketed with events on the stream it is launched on (torch's current stream).
op_timer = None


op_tag = "main"       # which stream of the step schedule the launches being issued belong to (see `tagged`)


class tagged:
    """with tagged("side"): native launches issued inside are recorded by an active OpTimer as work of the step's SECOND
    stream (the frozen text encoder, the next batch's coordinate geometry): bench.py prices its `roofline` object on the
    critical path, i.e. on "main" launches only."""

    def __init__(self, tag):
        self.tag, self.prev = tag, None

    def __enter__(self):
        global op_tag
        self.prev, op_tag = op_tag, self.tag

    def __exit__(self, *a):
        global op_tag
        op_tag = self.prev
        return False


class OpTimer:
    """Collects (start, end) HIP event pairs per native entry point."""

    def __init__(self):
        self.events = {}
        self.tags = {}

    def record(self, name):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        self.events.setdefault(name, []).append((s, e))
        self.tags.setdefault(name, op_tag)
        return s, e

    def summary(self):
        """name -> (calls, ms per call); call after torch.cuda.synchronize().  ms per call: the MEDIAN of the brackets when there
        are at least five (an eager launch whose enqueue the host delayed leaves the queue idle inside its bracket: one such
        outlier of 0.6 ms among 99 brackets of 22 us moved a mean by 27 % and flipped bench.py's kernel ranking), the mean below."""
        out = {}
        for k, v in self.events.items():
            t = sorted(s.elapsed_time(e) for s, e in v)
            n = len(t)
            out[k] = (n, (t[n // 2] if n % 2 else 0.5 * (t[n // 2 - 1] + t[n // 2])) if n >= 5 else sum(t) / n)
        return out


class _timed:
    def __init__(self, name, dims=()):
        self.name = (name,) + tuple(int(d) for d in dims)   # op name + its shape
        self.pair = None

    def __enter__(self):
        if op_timer is not None:
            self.pair = op_timer.record(self.name)
            self.pair[0].record()

    def __exit__(self, *a):
        if self.pair is not None:
            self.pair[1].record()
        return False


def _check_contiguous(t, name):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")


def _check_float(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float tensor")


def _check_int(t, name):
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")


def _check_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")


def _require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("CPU not supported")


_fps_ws = {}      # (device, stream) -> persistent workspace; its first int32 is the sticky give-up flag


def _fps_workspace(device, nbytes):
    """One FPS workspace per (device, stream), grown on demand.  The status block in front (see
    include/eda_hip.h) is zeroed once here and never by the kernels, so a give-up of ANY call stays
    visible until fps_status() is asked -- at a natural synchronisation point, not per call."""
    key = (device, _stream())
    ws = _fps_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            # a fill kernel captured here would re-zero the sticky flag on every replay (a give-up of an earlier
            # replay would vanish): the workspace must exist before the capture starts
            raise RuntimeError("furthest_point_sampling: no workspace for this (device, stream) yet -- run one eager "
                               "sampling of this size on the stream before capturing it in a HIP graph "
                               "(eda_amd.ext.fps_prepare)")
        new = torch.full((max(nbytes, 4096),), 0, dtype=torch.uint8, device=device)
        if ws is not None:
            new[:16].copy_(ws[:16])
        _fps_ws[key] = ws = new
    return ws


def fps_prepare(device, batch, n_points, n_samples):
    """Create (outside any capture) the current stream's FPS workspace for samplings of up to this size."""
    nbytes = int(_lib.lib().eda_fps_workspace_bytes(int(batch), int(n_points), int(n_samples)))
    _fps_workspace(torch.device(device), nbytes)


def fps_status(device=None, reset=False):
    """Number of FPS workspaces whose sticky give-up flag is set (0 = every furthest-point-sampling call
    so far ran to completion).  Synchronises with the device: call it where the host waits anyway
    (end of a step / epoch).  A non-zero value means some sampled index sets are NOT the FPS result."""
    bad = 0
    for (dev, _), ws in _fps_ws.items():
        if device is not None and torch.device(device) != dev:
            continue
        if int(ws[:4].view(torch.int32)[0].item()) != 0:
            bad += 1
            if reset:
                ws[:16].zero_()
    return bad


def fps_repaired(device=None):
    """How many furthest-point-sampling calls so far gave up their inter-workgroup spin (the cluster kernels were not
    co-resident) and were REPAIRED on the device by the single-workgroup bucket sampler (int 2 of the status block;
    include/eda_hip.h eda_fps_set_policy).  Their indices are correct; the number says how often ~1 s was lost.
    Synchronises."""
    total = 0
    for (dev, _), ws in _fps_ws.items():
        if device is None or torch.device(device) == dev:
            total += int(ws[:16].view(torch.int32)[2].item())
    return total


def fps_set_policy(name):
    """"auto" (cluster kernels + gated bucket-sampler repair), "cluster", "bucket": include/eda_hip.h eda_fps_set_policy."""
    _lib.check(_lib.lib().eda_fps_set_policy({"auto": 0, "cluster": 1, "bucket": 2}[name]), "eda_fps_set_policy")


def fps_last_duration_ms(device=None):
    """Wall time the multi-workgroup sampler last took on the device (its own 100 MHz clock reads, int 3 of the
    workspace's status block), per workspace: a diagnostic for runs that replay graphs, where no host-side event can
    bracket a single kernel.  Synchronises."""
    out = []
    for (dev, _), ws in _fps_ws.items():
        if device is None or torch.device(device) == dev:
            out.append(int(ws[:16].view(torch.int32)[3].item()) * 1e-5)
    return out


def furthest_point_sampling(points, nsamples):
    """sampling.cpp:70-91 -- points (B,N,3) f32 -> (B,nsamples) i32."""
    _check_contiguous(points, "points")
    _check_float(points, "points")
    _require_gpu(points)
    L = _lib.lib()
    b, n = points.shape[0], points.shape[1]
    nsamples = int(nsamples)
    out = torch.empty((b, nsamples), dtype=torch.int32, device=points.device)
    if b == 0 or nsamples == 0:
        return out
    ws_bytes = L.eda_fps_workspace_bytes(b, n, nsamples)
    ws = _fps_workspace(points.device, ws_bytes)
    with torch.cuda.device(points.device), _timed('furthest_point_sampling', (b, n, nsamples)):
        rc = L.eda_furthest_point_sampling_f32(points.data_ptr(), b, n, nsamples, out.data_ptr(),
                                               ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "eda_furthest_point_sampling_f32")
    return out


def furthest_point_sampling_prefix(points, nsamples):
    """furthest_point_sampling with the same result for every input, fast when `points` already is in
    sampling order (eda_furthest_point_sampling_prefix_f32: 0..m-1 is verified, not assumed).  An extra of
    this module, not one of the nine reference callables."""
    _check_contiguous(points, "points")
    _check_float(points, "points")
    _require_gpu(points)
    L = _lib.lib()
    b, n = points.shape[0], points.shape[1]
    nsamples = int(nsamples)
    out = torch.empty((b, nsamples), dtype=torch.int32, device=points.device)
    if b == 0 or nsamples == 0:
        return out
    ws = _fps_workspace(points.device, L.eda_fps_prefix_workspace_bytes(b, n, nsamples))
    with torch.cuda.device(points.device), _timed('furthest_point_sampling', (b, n, nsamples)):
        rc = L.eda_furthest_point_sampling_prefix_f32(points.data_ptr(), b, n, nsamples, out.data_ptr(),
                                                      ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "eda_furthest_point_sampling_prefix_f32")
    return out


def gather_points(points, idx):
    """sampling.cpp:20-43 -- points (B,C,N), idx (B,m) -> (B,C,m)."""
    _check_contiguous(points, "points"); _check_contiguous(idx, "idx")
    _check_float(points, "points"); _check_int(idx, "idx")
    if points.is_cuda:
        _check_cuda(idx, "idx")
    _require_gpu(points)
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device), _timed('gather_points', (b, c, n, m)):
        rc = _lib.lib().eda_gather_points_f32(points.data_ptr(), idx.data_ptr(), b, c, n, m,
                                              out.data_ptr(), _stream())
    _lib.check(rc, "eda_gather_points_f32")
    return out


def gather_points_grad(grad_out, idx, n):
    """sampling.cpp:45-69 -- grad_out (B,C,m), idx (B,m) -> (B,C,n)."""
    _check_contiguous(grad_out, "grad_out"); _check_contiguous(idx, "idx")
    _check_float(grad_out, "grad_out"); _check_int(idx, "idx")
    if grad_out.is_cuda:
        _check_cuda(idx, "idx")
    _require_gpu(grad_out)
    b, c, m = grad_out.shape
    n = int(n)
    out = torch.empty((b, c, n), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device), _timed('gather_points_grad', (b, c, n, m)):
        rc = _lib.lib().eda_gather_points_grad_f32(grad_out.data_ptr(), idx.data_ptr(), b, c, n, m,
                                                   out.data_ptr(), _stream())
    _lib.check(rc, "eda_gather_points_grad_f32")
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """ball_query.cpp:13-37 -- new_xyz (B,m,3), xyz (B,N,3) -> (B,m,nsample) i32.
    NB: argument order differs from pointnet2_utils.ball_query(radius, nsample, xyz, new_xyz)."""
    _check_contiguous(new_xyz, "new_xyz"); _check_contiguous(xyz, "xyz")
    _check_float(new_xyz, "new_xyz"); _check_float(xyz, "xyz")
    if new_xyz.is_cuda:
        _check_cuda(xyz, "xyz")
    _require_gpu(new_xyz)
    b, n = xyz.shape[0], xyz.shape[1]
    m = new_xyz.shape[1]
    nsample = int(nsample)
    idx = torch.empty((new_xyz.shape[0], m, nsample), dtype=torch.int32, device=new_xyz.device)
    L = _lib.lib()
    ws, ws_bytes = None, 0
    if n >= 4096:                      # large scenes: uniform-grid candidate search needs scratch
        ws_bytes = L.eda_ball_query_workspace_bytes(b, n, m)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=new_xyz.device)
    with torch.cuda.device(new_xyz.device), _timed('ball_query', (b, n, m, nsample)):
        rc = L.eda_ball_query_f32(new_xyz.data_ptr(), xyz.data_ptr(), b, n, m, float(radius), nsample,
                                  idx.data_ptr(), ws.data_ptr() if ws is not None else None, ws_bytes,
                                  _stream())
    _lib.check(rc, "eda_ball_query_f32")
    return idx


def group_points(points, idx):
    """group_points.cpp:17-40 -- points (B,C,N), idx (B,m,ns) -> fresh (B,C,m,ns)."""
    _check_contiguous(points, "points"); _check_contiguous(idx, "idx")
    _check_float(points, "points"); _check_int(idx, "idx")
    if points.is_cuda:
        _check_cuda(idx, "idx")
    _require_gpu(points)
    b, c, n = points.shape
    npoints, nsample = idx.shape[1], idx.shape[2]
    out = torch.empty((b, c, npoints, nsample), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device), _timed('group_points', (b, c, n, npoints, nsample)):
        rc = _lib.lib().eda_group_points_f32(points.data_ptr(), idx.data_ptr(), b, c, n, npoints,
                                             nsample, out.data_ptr(), _stream())
    _lib.check(rc, "eda_group_points_f32")
    return out


def group_points_grad(grad_out, idx, n):
    """group_points.cpp:42-65 -- grad_out (B,C,m,ns), idx (B,m,ns) -> (B,C,n)."""
    _check_contiguous(grad_out, "grad_out"); _check_contiguous(idx, "idx")
    _check_float(grad_out, "grad_out"); _check_int(idx, "idx")
    if grad_out.is_cuda:
        _check_cuda(idx, "idx")
    _require_gpu(grad_out)
    b, c = grad_out.shape[0], grad_out.shape[1]
    npoints, nsample = idx.shape[1], idx.shape[2]
    n = int(n)
    out = torch.empty((b, c, n), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device), _timed('group_points_grad', (b, c, n, npoints, nsample)):
        rc = _lib.lib().eda_group_points_grad_f32(grad_out.data_ptr(), idx.data_ptr(), b, c, n,
                                                  npoints, nsample, out.data_ptr(), _stream())
    _lib.check(rc, "eda_group_points_grad_f32")
    return out


def three_nn(unknowns, knows):
    """interpolate.cpp:19-45 -- unknowns (B,n,3), knows (B,m,3) -> [dist2 (B,n,3) f32, idx (B,n,3) i32]."""
    _check_contiguous(unknowns, "unknowns"); _check_contiguous(knows, "knows")
    _check_float(unknowns, "unknowns"); _check_float(knows, "knows")
    if unknowns.is_cuda:
        _check_cuda(knows, "knows")
    _require_gpu(unknowns)
    b, n = unknowns.shape[0], unknowns.shape[1]
    m = knows.shape[1]
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=unknowns.device)
    dist2 = torch.empty((b, n, 3), dtype=torch.float32, device=unknowns.device)
    with torch.cuda.device(unknowns.device), _timed('three_nn', (b, n, m)):
        rc = _lib.lib().eda_three_nn_f32(unknowns.data_ptr(), knows.data_ptr(), b, n, m,
                                         dist2.data_ptr(), idx.data_ptr(), _stream())
    _lib.check(rc, "eda_three_nn_f32")
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """interpolate.cpp:47-75 -- points (B,C,m), idx (B,n,3), weight (B,n,3) -> (B,C,n)."""
    _check_contiguous(points, "points"); _check_contiguous(idx, "idx"); _check_contiguous(weight, "weight")
    _check_float(points, "points"); _check_int(idx, "idx"); _check_float(weight, "weight")
    if points.is_cuda:
        _check_cuda(idx, "idx"); _check_cuda(weight, "weight")
    _require_gpu(points)
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.empty((b, c, n), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device), _timed('three_interpolate', (b, c, m, n)):
        rc = _lib.lib().eda_three_interpolate_f32(points.data_ptr(), idx.data_ptr(), weight.data_ptr(),
                                                  b, c, m, n, out.data_ptr(), _stream())
    _lib.check(rc, "eda_three_interpolate_f32")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """interpolate.cpp:76-104 -- grad_out (B,C,n), idx/weight (B,n,3) -> (B,C,m)."""
    _check_contiguous(grad_out, "grad_out"); _check_contiguous(idx, "idx"); _check_contiguous(weight, "weight")
    _check_float(grad_out, "grad_out"); _check_int(idx, "idx"); _check_float(weight, "weight")
    if grad_out.is_cuda:
        _check_cuda(idx, "idx"); _check_cuda(weight, "weight")
    _require_gpu(grad_out)
    b, c, n = grad_out.shape
    m = int(m)
    out = torch.empty((b, c, m), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device), _timed('three_interpolate_grad', (b, c, n, m)):
        rc = _lib.lib().eda_three_interpolate_grad_f32(grad_out.data_ptr(), idx.data_ptr(),
                                                       weight.data_ptr(), b, c, n, m,
                                                       out.data_ptr(), _stream())
    _lib.check(rc, "eda_three_interpolate_grad_f32")
    return out


def device_copy(src, dst):
    """dst <- src with the library's copy kernel (eda_device_copy_f32): the achievable-HBM yardstick
    of bench.py, not part of the reference's path."""
    _require_gpu(src)
    _require_gpu(dst)
    if src.dtype != torch.float32 or dst.dtype != torch.float32 or src.numel() != dst.numel():
        raise RuntimeError("device_copy: two float32 tensors of equal size")
    if not (src.is_contiguous() and dst.is_contiguous()):
        raise RuntimeError("device_copy: tensors must be contiguous tensor")
    _lib.check(_lib.lib().eda_device_copy_f32(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()),
               "eda_device_copy_f32")
    return dst


def set_fma_mode(mode):
    """0 = nvcc-style contracted distance arithmetic (default), 1 = strict IEEE."""
    _lib.check(_lib.lib().eda_set_fma_mode(int(mode)), "eda_set_fma_mode")


_NAMES = ("gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
          "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
          "group_points_grad")


def install_as_pointnet2_ext():
    """Make ``import pointnet2._ext`` resolve to this module's nine functions."""
    mod = types.ModuleType("pointnet2._ext")
    for name in _NAMES:
        setattr(mod, name, globals()[name])
    sys.modules["pointnet2._ext"] = mod
    pkg = sys.modules.get("pointnet2")
    if pkg is not None:
        setattr(pkg, "_ext", mod)
    return mod
