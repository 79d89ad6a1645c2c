"""SyncBatchNorm-equivalent statistics exchange for the channels-last BN+ReLU sites of the path.

The reference converts every BatchNorm to ``SyncBatchNorm`` when more than one GPU trains
(main_utils.py:336-338): at N > 1 its batch statistics cover the GLOBAL batch.  This repo's default
at N > 1 is per-GPU statistics (8 scenes x >= 2048 positions per channel already);
``enable(group)`` switches every BN+ReLU site (set-abstraction / feature-propagation MLPs, heads,
positional embeddings) to the reference's semantics:

* forward: ONE all-reduce of the packed vector [sum z | sum z^2 | count] (2C+1 doubles) per BN layer
  (torch's SyncBatchNorm all-gathers mean / invstd / count: same information), then mean, biased
  variance, running-statistics update with the unbiased GLOBAL variance;
* backward: ONE all-reduce of [sum gy | sum gy*xhat] (2C doubles), then
  dz = gamma*rstd*(gy - s1/N - xhat*s2/N) with the global N; d(gamma), d(beta) stay LOCAL sums (the
  gradient all-reduce of eda_amd/parallel.py adds them over ranks like every other gradient).

Two implementations:

* the FUSED set-abstraction / feature-propagation calls (eda_sa_fused_fwd/bwd_f32: 16 of the 68 BatchNorm layers, and
  all of the ones with 10^5..10^6 rows) keep their fusion: `enable()` registers a hook with the library
  (include/eda_hip.h: eda_set_bn_sync) and the native call hands every layer's packed fp64 sums to
  `dist.all_reduce` between the kernel that accumulates them and the kernel that finalises them -- one collective per
  layer and direction on the same stream, no host synchronisation (every rank must hold the same number of rows);
* the single-launch small-row kernels of the heads / positional embeddings cannot stop in the middle for a collective:
  those sites run the device-agnostic torch ops below (also what runs under gloo on CPU in
  tests/test_parallel_cpu.py), one packed collective per BatchNorm and direction, no host synchronisation either.

Round 5, `enable(native=True)`: NO collective at all.  Every rank's process maps a slab of every other rank's device memory
(csrc/peer.hip; the 64-byte IPC handles travel once through `dist.all_gather_object`), and the kernels that hold a
channel's local sums exchange them there themselves (csrc/peer.h) -- inside the single-launch heads / positional-embedding
kernels, in the last block of the statistics kernel, and as a one-launch vector exchange behind the fused calls' hook.
Every BatchNorm site stays on its fused kernel, nothing runs on the host between kernels, and the training step captures
into hipGraphs with global-batch statistics in it (what RCCL collectives inside a capture did not allow on this stack:
111 scenes/s eager, profiles/r04_bench_force_dist_sync_bn.json).  One node, <= 8 ranks.
"""
import ctypes

import torch
import torch.distributed as dist
from torch.autograd import Function

_group = None
_enabled = False
_fused_hook = None          # the ctypes callback object (must stay alive while registered)
_buffers = {}               # id -> tensor: device buffers the native calls may ask to all-reduce parts of
_reduce = None              # test seam: replaces dist.all_reduce(t, group) in the hook
_single_rank_too = False    # run the exchange even in a ONE-rank group (bench.py --force-dist: the N > 1 code path on one GPU)

_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p)


def note_buffer(t):
    """Register a device tensor whose memory a fused native call may hand to the statistics hook."""
    _buffers[id(t)] = t
    return t


def forget_buffer(t):
    _buffers.pop(id(t), None)


def _hook(user, buf, n, stream):
    """eda_bn_sync_fn: sum `n` doubles at device address `buf` over the ranks, in place, enqueued on the current stream
    (the native call launches on torch's current stream, so the collective is ordered between its kernels)."""
    try:
        for t in list(_buffers.values()):
            lo = t.data_ptr()
            if lo <= buf and buf + 8 * n <= lo + t.numel() * t.element_size():
                off = buf - lo
                view = t.view(torch.uint8).reshape(-1)[off:off + 8 * n].view(torch.float64)
                if _reduce is not None:
                    _reduce(view)
                else:
                    dist.all_reduce(view, group=_group)
                return 0
        return 2                      # unknown buffer
    except Exception:                 # never let an exception cross the C boundary
        import traceback
        traceback.print_exc()
        return 3


def install_fused_hook(world):
    """Register the hook with libeda_hip.so (GPU builds): the fused SA / FP calls then exchange their statistics."""
    global _fused_hook
    from . import _lib
    _fused_hook = _CB(_hook)
    _lib.check(_lib.lib().eda_set_bn_sync(ctypes.cast(_fused_hook, ctypes.c_void_p), None, int(world)), "eda_set_bn_sync")


def remove_fused_hook():
    global _fused_hook
    if _fused_hook is not None:
        from . import _lib
        _lib.lib().eda_set_bn_sync(None, None, 1)
        _fused_hook = None


def fused_hook_installed():
    return _fused_hook is not None or _native


_native = False


def native():
    """True while the library's own kernels exchange the statistics (enable(native=True)): no site leaves its fused kernel."""
    return _native


def _connect_peers(group):
    """Create this rank's slab, gather every rank's IPC handle, map the peers, start a clean session (csrc/peer.hip)."""
    from . import _lib
    L = _lib.lib()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = ctypes.create_string_buffer(64)
    _lib.check(L.eda_peer_create(mine), "eda_peer_create")
    handles = [None] * world
    if world > 1:
        dist.all_gather_object(handles, bytes(mine.raw), group=group)
    else:
        handles[0] = bytes(mine.raw)
    blob = ctypes.create_string_buffer(b"".join(handles), 64 * world)
    _lib.check(L.eda_peer_connect(rank, world, blob), "eda_peer_connect")
    # a new session starts from sequence number 0 on every rank, with no counted timeout and no stale tag: every rank zeroes
    # its own slab, and nobody's kernels write into a slab before every rank has done so and mapped all of them
    _lib.check(L.eda_peer_reset(), "eda_peer_reset")
    if world > 1:
        dist.barrier(group=group)
    return world


def peer_selftest(group=None, inject_wrong_tag=False):
    """One exchange of a known vector through the peer slabs by EVERY rank of the group (csrc/peer.hip: eda_peer_selftest);
    True when every rank got the right sums with no timed-out poll.  After a failure the slabs are reset (a counted
    timeout is sticky and a wrong tag may be left behind), so the result is the same on every rank and the exchange can
    be tried again."""
    from . import _lib
    L = _lib.lib()
    rc = L.eda_peer_selftest(torch.cuda.current_stream().cuda_stream, int(bool(inject_wrong_tag)))
    ok = rc == 0
    world = dist.get_world_size(group)
    if world > 1:
        flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        ok = int(flag.item()) == 0
    if not ok:
        if world > 1:
            dist.barrier(group=group)          # nobody is still polling
        _lib.check(L.eda_peer_reset(), "eda_peer_reset")
        if world > 1:
            dist.barrier(group=group)
    return ok


def enable(group=None, fused=True, single_rank_too=False, native=False):
    """Use global-batch statistics in every BN+ReLU site (requires an initialised process group).  fused=True (and a
    GPU): the fused SA / FP calls stay fused and exchange their sums through the library hook.  single_rank_too: also
    in a one-rank group (same results as without; exercises the collectives' code path on a one-GPU box).
    native=True: the exchange happens inside the library's kernels through peer-mapped memory -- no collective, every
    site stays fused, capturable (module docstring).  The peer path is only trusted after a self-test exchange on THIS
    set of GPUs (peer_selftest); if that fails on any rank, every rank logs one line and falls back to the collective
    hook (fused=True semantics).

    Preconditions of native=True (csrc/peer.h; they make the per-rank sequence numbers agree): every rank issues the same
    BatchNorm launches in the same order on ONE stream, with the same shapes -- equal batch per rank (drop_last), the
    same launch-structure knobs on every rank.  `check()` is the host-side guard: call it once per step or per epoch."""
    global _enabled, _group, _single_rank_too, _native
    if not dist.is_initialized():
        raise RuntimeError("sync_bn.enable() needs torch.distributed to be initialised")
    _enabled, _group, _single_rank_too = True, group, bool(single_rank_too)
    if native:
        if not torch.cuda.is_available():
            raise RuntimeError("sync_bn.enable(native=True) needs the GPU library")
        if dist.get_world_size(group) > 1 or single_rank_too:
            from . import _lib
            world = _connect_peers(group)
            if peer_selftest(group, inject_wrong_tag=_selftest_inject):
                _lib.check(_lib.lib().eda_set_bn_sync_native(world), "eda_set_bn_sync_native")
                _native = True
                return
            import sys
            print("[eda_amd.sync_bn] rank %d: the peer-memory self-test failed (%s); BatchNorm statistics go through "
                  "collectives instead" % (dist.get_rank(group), _lib.last_error()), file=sys.stderr, flush=True)
            _lib.lib().eda_peer_disconnect()
        else:
            return
    if fused and torch.cuda.is_available() and (dist.get_world_size(group) > 1 or single_rank_too):
        install_fused_hook(dist.get_world_size(group))


_selftest_inject = False    # test seam: enable(native=True) runs its self-test with an injected wrong tag


def check():
    """Raise if an in-kernel statistics exchange ever ran into its poll bound (the statistics since then are garbage).
    A host read: it synchronises the device, so call it once per step at most -- bench.py does after its timed region."""
    if _native:
        n = peer_timeouts()
        if n != 0:
            raise RuntimeError("sync_bn: %d in-kernel BatchNorm statistics exchanges timed out -- the ranks' launch "
                               "sequences diverged or a peer died; statistics are invalid from that point on" % n)


def disable():
    global _enabled, _group, _single_rank_too, _native
    _enabled, _group, _single_rank_too = False, None, False
    if _native:
        from . import _lib
        _lib.lib().eda_set_bn_sync_native(0)
        _lib.lib().eda_peer_disconnect()       # the next enable() maps its own group's slabs and starts a clean session
        _native = False
    remove_fused_hook()


def peer_timeouts():
    """Exchanges that gave up their bounded spin since start-up (0 in a healthy run; host read, synchronises)."""
    from . import _lib
    return int(_lib.lib().eda_peer_timeouts())


def enabled():
    return _enabled and dist.is_initialized() and (dist.get_world_size(_group) > 1 or _single_rank_too)


def diverts():
    """Do the small-row BN+ReLU sites have to leave their fused kernels for the torch ops below (a collective in the middle)?"""
    return enabled() and not _native


class _SyncBNReLU(Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, eps, momentum, training):
        R, C = z.shape
        if training:
            zd = z.double()
            packed = torch.cat([zd.sum(0), (zd * zd).sum(0), torch.full((1,), float(R), dtype=torch.float64, device=z.device)])
            dist.all_reduce(packed, group=_group)
            n = packed[2 * C]
            mean = packed[:C] / n
            var = (packed[C:2 * C] / n - mean * mean).clamp_min(0.0)
            if running_mean is not None:
                unbiased = var * (n / (n - 1.0).clamp_min(1.0))          # (on the device: no host synchronisation)
                running_mean.mul_(1.0 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
                running_var.mul_(1.0 - momentum).add_(unbiased.to(running_var.dtype), alpha=momentum)
            mean, var = mean.float(), var.float()
        else:
            mean, var = running_mean, running_var
            n = torch.tensor(float(R), dtype=torch.float64, device=z.device)
        rstd = torch.rsqrt(var + eps)
        xhat = (z - mean) * rstd
        y = torch.relu(xhat * gamma + beta)
        ctx.save_for_backward(xhat, y, gamma, rstd, n)
        ctx.training = bool(training)
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, y, gamma, rstd, n = ctx.saved_tensors
        gy = dy * (y > 0)
        s1 = gy.sum(0, dtype=torch.float64)
        s2 = (gy * xhat).sum(0, dtype=torch.float64)
        dgamma, dbeta = s2.float(), s1.float()
        if ctx.training:
            packed = torch.cat([s1, s2])
            dist.all_reduce(packed, group=_group)
            C = s1.numel()
            m1 = (packed[:C] / n).float()
            m2 = (packed[C:] / n).float()
            dz = (gy - m1 - xhat * m2) * (gamma * rstd)
        else:
            dz = gy * (gamma * rstd)
        return dz, dgamma, dbeta, None, None, None, None, None


def bn_relu(bn, z, pool=1):
    """relu(SyncBatchNorm `bn`(z)) on rows z (R, C), optionally max-pooled over `pool` consecutive rows."""
    training = bn.training or not bn.track_running_stats
    y = _SyncBNReLU.apply(z, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                          bn.running_var if bn.track_running_stats else None, bn.eps, bn.momentum, training)
    if pool > 1:
        R, C = y.shape
        y = y.view(R // pool, pool, C).max(dim=1)[0]
    return y
