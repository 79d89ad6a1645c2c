"""Data parallelism for the hot path: one process per GPU, scenes sharded across
ranks, gradients summed with ONE all-reduce of a flat fp32 buffer.

The reference wraps the model in DistributedDataParallel (main_utils.py:343-346:
25 MB buckets -> 4 NCCL all-reduces of 85.7 MB total per step) plus SyncBatchNorm
(68 layers -> 136 latency-bound micro-collectives, main_utils.py:336-338).  On
MI355X the 8 GPUs are fully connected by xGMI (7 links x ~153 GB/s per GPU): a
single large all-reduce lets RCCL use every link at once (direct
reduce-scatter + all-gather ~ 2 x 10.7 MB per link).  So all trainable parameters
are views of ONE flat buffer and all gradients are gathered into ONE flat buffer
(a multi-tensor copy; autograd itself only hands over freshly produced tensors, no
per-parameter accumulate kernels), which is what RCCL reduces and what the
optimizer updates -- one fused AdamW kernel per learning-rate group instead of one
per ~50 parameter tensors.  Batch-norm statistics stay per-GPU (8 scenes x >= 4096
positions per channel); that deviation from SyncBN is stated in DESIGN.md.
"""
import contextlib
import os

import torch
import torch.distributed as dist


def _siblings_adjacent(module, ordered):
    """Re-order (name, parameter) pairs so that the same-named parameters of sibling sub-stacks (modules
    exposing ``sibling_stacks()``: the three ThreeLayerMLPs of a prediction head) lie back to back: the
    grouped kernels then see ONE packed weight (eda_amd/grouped.py: the shared input's gradient is a
    single GEMM over the concatenated weights)."""
    names = [n for n, _ in ordered]
    pos = {n: i for i, n in enumerate(names)}
    key = {n: (i, 0, 0) for i, n in enumerate(names)}
    for mname, mod in module.named_modules():
        fn = getattr(mod, "sibling_stacks", None)
        if fn is None:
            continue
        stacks = fn()
        pre = (mname + ".") if mname else ""
        first = [n for n in names if n.startswith(pre + stacks[0] + ".")]
        if not first:
            continue
        leaves = [n[len(pre + stacks[0] + "."):] for n in first]
        anchor = min(pos[n] for n in first)
        for li, leaf in enumerate(leaves):
            for si, st in enumerate(stacks):
                n = pre + st + "." + leaf
                if n in key:
                    key[n] = (anchor, 1 + li, si)
    return sorted(ordered, key=lambda np_: key[np_[0]])


class FlatParams:
    """Flat parameter / gradient storage, split into learning-rate groups.

    ``group_of(name) -> key`` assigns every trainable parameter to a group (the
    reference uses three: backbone_net / text_encoder / rest, main_utils.py:277-305).
    ``self.groups`` maps key -> nn.Parameter over that group's slice of the flat
    buffer, whose ``.grad`` is the matching slice of the flat gradient buffer: hand
    these to the optimizer.

    Difference from per-parameter optimisation that a caller must know: a parameter that received NO gradient in a
    step has a ZERO slice here, where the reference leaves ``p.grad = None`` and its AdamW skips the parameter
    (no weight decay, no moment decay for that step).  In BeaUTyDETR every trainable parameter receives a gradient in
    every step (the frozen text encoder is excluded from the buffer), so the two coincide; a model with
    conditionally unused branches would see weight decay applied to them.
    """

    def __init__(self, module, group_of=None):
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters")
        group_of = group_of or (lambda name: "all")
        keys = []
        for n, _ in named:
            k = group_of(n)
            if k not in keys:
                keys.append(k)
        ordered = [(n, p) for k in keys for (n, p) in named if group_of(n) == k]
        ordered = _siblings_adjacent(module, ordered)
        self.params = [p for _, p in ordered]
        dev = self.params[0].device
        # every parameter starts on a 16-byte boundary (4 floats): the MFMA GEMMs read weight rows
        # with 16-byte loads (csrc/gemm.hip); the <= 3 padding floats in front of a parameter stay
        # zero in both buffers (zero gradient -> AdamW leaves a zero parameter at zero)
        total = sum((p.numel() + 3) // 4 * 4 for p in self.params)
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        # this class reads gradients from `.grad` after the backward (collect_grads / the deferred queue) and reduces the
        # flat buffer itself: the backward forms that assign `.grad` without autograd delivery may run
        from . import posembed_batched
        posembed_batched.enable(True)
        self._grad_views = []
        self.layout = []                   # (parameter name, group key, offset in the flat buffers, numel)
        off = 0
        bounds = {}
        for n, p in ordered:
            k = group_of(n)
            off = (off + 3) // 4 * 4
            sl = slice(off, off + p.numel())
            self.layout.append((n, k, off, p.numel()))
            self.flat_param[sl].copy_(p.data.reshape(-1))
            p.data = self.flat_param[sl].view_as(p)          # the parameter now lives in the flat buffer
            p.grad = None
            self._grad_views.append(self.flat_grad[sl].view_as(p))
            lo, hi = bounds.get(k, (off, off))
            bounds[k] = (min(lo, off), off + p.numel())
            off += p.numel()
        from .wgrad_queue import WgradQueue
        self.queue = WgradQueue(self._locate_grad)
        self.shadow = None                 # W^T of the linear weights (eda_amd/wt_shadow.py)
        if dev.type == "cuda":
            self.queue.reserve(dev)              # pinned staging must exist before any stream capture
            if os.environ.get("EDA_WT_SHADOW", "1") != "0":
                # ... and so must the shadow (its descriptor table is a host -> device copy): built here, not at the
                # first deferred backward, which may already run under capture (eda_amd/pipeline.py)
                from . import wt_shadow
                self.shadow = wt_shadow.TransposedShadow([p for p in self.params if p.requires_grad])
        self._prefilled = False
        self._deferred_ptrs = set()
        self.group_bounds = dict(bounds)   # group key -> [lo, hi) in the flat buffers
        self.groups = {}
        for k, (lo, hi) in bounds.items():
            gp = torch.nn.Parameter(self.flat_param[lo:hi], requires_grad=True)
            gp.grad = self.flat_grad[lo:hi]
            self.groups[k] = gp

    def _locate_grad(self, t):
        """The slice of the flat GRADIENT buffer that corresponds to `t`, a contiguous view into
        the flat PARAMETER buffer (a whole parameter, a squeezed conv weight, a row range of a
        packed in-projection ...); None for anything else."""
        if not t.is_contiguous() or t.dtype != torch.float32 or t.device != self.flat_param.device:
            return None
        off = t.data_ptr() - self.flat_param.data_ptr()
        if off < 0 or off % 4 or off // 4 + t.numel() > self.flat_param.numel():
            return None
        off //= 4
        return self.flat_grad[off:off + t.numel()].view(t.shape)

    @contextlib.contextmanager
    def deferred_wgrad(self, flush=True):
        """Run the backward pass inside this context: the weight / bias gradients of the
        pointwise linear layers are queued (eda_amd/wgrad_queue.py) and written into the flat
        gradient buffer by ONE grouped kernel when the context exits; collect_grads() then only
        gathers what autograd still produced itself.  flush=False leaves the queue full: the caller
        continues with flush_and_reduce() (range-wise flush with the all-reduce underneath)."""
        from . import wgrad_queue, wt_shadow
        self.flat_grad.fill_(0.0)          # (a kernel, not a memset node: see DESIGN.md on graphs)
        prev, wgrad_queue.active = wgrad_queue.active, self.queue
        prev_shadow = wt_shadow.active
        if self.shadow is None and self.flat_grad.is_cuda and os.environ.get("EDA_WT_SHADOW", "1") != "0":
            self.shadow = wt_shadow.TransposedShadow([p for p in self.params if p.requires_grad])
        if self.shadow is not None and len(self.shadow):
            self.shadow.refresh()          # W^T of every linear weight, one launch: input gradients in the forward's GEMM form
            wt_shadow.active = self.shadow
        try:
            yield self.queue
        finally:
            wgrad_queue.active = prev
            wt_shadow.active = prev_shadow
            self._deferred_ptrs = self.queue.touched()
            if flush:
                self.queue.flush()
            self._prefilled = True

    def collect_grads(self, lo=None, hi=None):
        """Gather the gradients autograd produced this step into the flat buffer
        (multi-tensor copy) and release them, so the next backward again ASSIGNS
        instead of accumulating.  lo / hi (element offsets at parameter boundaries, deferred mode only): just the
        parameters inside [lo, hi) -- flush_and_reduce() gathers range by range."""
        ranged = lo is not None
        if ranged:
            assert self._prefilled, "range-wise gather: only after deferred_wgrad()"
            base = self.flat_grad.data_ptr()
            sel = [base + 4 * lo <= v.data_ptr() < base + 4 * hi for v in self._grad_views]
            have = [(v, p.grad) for v, p, s_ in zip(self._grad_views, self.params, sel) if s_ and p.grad is not None]
        else:
            have = [(v, p.grad) for v, p in zip(self._grad_views, self.params) if p.grad is not None]
        if self._prefilled:
            # deferred_wgrad() zeroed the buffer and its flush wrote the queued gradients; a
            # parameter that ALSO got a gradient from autograd (used outside the queue) adds to it
            # (range test, not pointer equality: the queue may have written only a ROW RANGE of a packed parameter --
            # the K | V rows of an in-projection whose q rows went through autograd -- and the autograd tensor then
            # holds zeros there; a copy would wipe the queued rows)
            import bisect
            spans = sorted(self._deferred_ptrs)
            starts = [s_ for s_, _ in spans]

            def overlaps(v):
                lo, hi = v.data_ptr(), v.data_ptr() + v.numel() * 4
                i = bisect.bisect_left(starts, hi) - 1      # last span starting below hi (spans are disjoint views)
                return i >= 0 and spans[i][0] + spans[i][1] > lo
            flags = [overlaps(v) for v, _ in have]
            both = [vg for vg, f in zip(have, flags) if f]
            have = [vg for vg, f in zip(have, flags) if not f]
            if both:
                torch._foreach_add_([v for v, _ in both], [g for _, g in both])
            if not ranged:
                self._prefilled = False
                self._deferred_ptrs = set()
        elif len(have) != len(self.params):
            self.flat_grad.fill_(0.0)          # (a fill kernel rather than a memset node)
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if ranged:
            for p, s_ in zip(self.params, sel):
                if s_:
                    p.grad = None
            return
        for p in self.params:
            p.grad = None

    def split_offset(self, frac=0.5):
        """Element offset of the parameter boundary closest to `frac` of the flat buffer."""
        total = self.flat_grad.numel()
        best = min((off for _, _, off, _ in self.layout), key=lambda o: abs(o - frac * total))
        return int(best)

    def flush_range(self, lo, hi):
        """Second half of a deferred backward for the buffer range [lo, hi): the grouped weight-gradient kernel over the
        queued entries that write into the range (an entry with ANY output inside it -- the LayerNorm reductions also
        write the bias of the linear in front, which may lie in another range -- runs now; later ranges only store
        into their own entries), then the gather of what autograd produced for the range's parameters.  Capturable."""
        base = self.flat_grad.data_ptr()
        a, b = base + 4 * lo, base + 4 * hi
        self.queue.flush(select=lambda views: any(a <= v.data_ptr() < b for v in views))
        self.collect_grads(lo, hi)

    def finish_ranges(self):
        """After the last flush_range(): anything still queued (nothing, if the ranges cover the buffer), state reset."""
        self.queue.flush()
        self._prefilled = False
        self._deferred_ptrs = set()

    def flush_and_reduce(self, world=None, group=None, frac=0.5):
        """Continue a backward that ran under deferred_wgrad(flush=False): the all-reduce of the first part of the flat
        buffer runs UNDERNEATH the grouped weight-gradient kernel of the second part (DDP overlaps its buckets with
        the backward, main_utils.py:343-346; here everything the collective needs is produced at the very end of
        the backward, so the overlap is between the two halves of that tail).  Returns a handle; wait() completes
        both collectives and applies 1/world (the global-norm clip needs the complete reduced gradient, so the
        optimizer cannot start earlier).  world <= 1: flushes, gathers, returns a no-op handle."""
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        total = self.flat_grad.numel()
        flat = self.flat_grad

        class _Done:
            def __init__(self, works):
                self.works = works

            def wait(self):
                for w in self.works:
                    w.wait()
                if world > 1:
                    flat.mul_(1.0 / world)
                return True
        if world <= 1:
            self.flush_range(0, total)
            self.finish_ranges()
            return _Done([])
        mid = self.split_offset(frac)
        works = []
        for lo, hi in ((0, mid), (mid, total)):
            if hi > lo:
                self.flush_range(lo, hi)
                works.append(dist.all_reduce(self.flat_grad[lo:hi], group=group, async_op=True))
        self.finish_ranges()
        return _Done(works)

    def backward_overlapped(self, loss, world=None, group=None):
        """loss.backward() with the weight gradients deferred and the gradient all-reduce overlapped with their
        computation; returns the handle of flush_and_reduce()."""
        with self.deferred_wgrad(flush=False):
            loss.backward()
        return self.flush_and_reduce(world, group)

    def all_reduce_mean(self, world=None, async_op=False):
        """Mean of the flat gradient over the ranks (DDP semantics): one all-reduce of the whole buffer, then 1/world.
        async_op=True returns a handle whose wait() completes the collective AND applies the 1/world scaling (the
        gradient is not a mean before wait() returns)."""
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
        if world <= 1:
            return None
        work = dist.all_reduce(self.flat_grad, async_op=async_op)
        if not async_op:
            self.flat_grad.mul_(1.0 / world)
            return None
        flat = self.flat_grad

        class _MeanWork:
            def wait(self_inner):
                work.wait()
                flat.mul_(1.0 / world)
                return True
        return _MeanWork()

    def clip_grad_norm_(self, max_norm, pre_scale=1.0):
        """torch.nn.utils.clip_grad_norm_ over all parameters (main_utils.py:483-486) on the flat buffer.
        pre_scale: the buffer still holds the SUM over ranks (the all-reduce's 1/world has not been applied): the
        mean's norm is pre_scale * |sum| and both factors go into the one multiplication."""
        norm = torch.linalg.vector_norm(self.flat_grad)
        if pre_scale != 1.0:
            norm = norm * pre_scale
        self.flat_grad.mul_(torch.clamp(max_norm / (norm + 1e-6), max=1.0) * pre_scale)
        return norm


class FlatClipAdamW:
    """clip_grad_norm_ + torch.optim.AdamW.step() of a FlatParams model as TWO launches (csrc/optim.hip) instead of ~12: the
    norm of the flat gradient (+ the step counters, + the learning rates read from a pinned host array), then one pass that
    applies the clip coefficient, the decoupled weight decay, both moments and the bias corrections.  `optimizer` is a
    torch.optim.AdamW over flat.groups (amsgrad / maximize off) and stays the OWNER of the state: its state_dict()
    (step, exp_avg, exp_avg_sq per group) is what checkpoints carry (eda_amd/checkpoint.py) and stepping it with
    optimizer.step() instead gives the same numbers to fp32 rounding (tests/test_flat_adamw_gpu.py).  Learning rates are read
    from optimizer.param_groups at every step() -- a host write, so a scheduler works across replays of a captured graph:
    call sync_lrs() before each replay."""

    def __init__(self, flat, optimizer):
        import ctypes
        from . import _lib
        self.flat, self.opt = flat, optimizer
        groups = optimizer.param_groups
        assert 1 <= len(groups) <= 4 and all(len(g["params"]) == 1 for g in groups), "one flat parameter per group, at most 4 groups"
        assert not any(g.get("amsgrad") or g.get("maximize") for g in groups), "amsgrad / maximize are not implemented"
        dev = flat.flat_param.device
        base = flat.flat_param.data_ptr()
        self.lo, self.hi = [], []
        for g in groups:
            p = g["params"][0]
            off = (p.data_ptr() - base) // 4
            assert p.is_contiguous() and 0 <= off and off + p.numel() <= flat.flat_param.numel() and off % 4 == 0
            self.lo.append(off)
            self.hi.append(off + p.numel())
            st = optimizer.state[p]
            if "step" not in st:                                   # torch's own lazy initialisation, capturable layout
                st["step"] = torch.zeros((), dtype=torch.float32, device=dev)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            assert st["step"].is_cuda and st["step"].dtype == torch.float32, "AdamW(capturable=True) keeps its step on the device"
        n = len(groups)
        self.n = n
        self.ws = torch.zeros(_lib.lib().eda_grad_sumsq_workspace_bytes(), dtype=torch.uint8, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.lr_host = torch.zeros(4, dtype=torch.float32).pin_memory()
        self.lr_dev = torch.zeros(12, dtype=torch.float32, device=dev)      # rates | bias corrections (csrc/optim.hip)
        self._c, self._lib = ctypes, _lib
        self.sync_lrs()

    def sync_lrs(self):
        """Current learning rates of the groups into the pinned array the next step (or graph replay) reads."""
        for k, g in enumerate(self.opt.param_groups):
            self.lr_host[k] = float(g["lr"])

    def step(self, max_norm=0.0, pre_scale=1.0):
        """Returns |grad|_2 * pre_scale as a device tensor (what clip_grad_norm_ returns)."""
        c, L = self._c, self._lib
        self.sync_lrs()
        fl, groups = self.flat, self.opt.param_groups
        sts = [self.opt.state[g["params"][0]] for g in groups]
        P = c.c_void_p * self.n
        F = c.c_float * self.n
        LG = c.c_long * self.n
        steps = P(*[s["step"].data_ptr() for s in sts])
        n = fl.flat_grad.numel()
        stream = torch.cuda.current_stream().cuda_stream
        with torch.cuda.device(fl.flat_grad.device):
            D = c.c_double * self.n
            b1, b2 = D(*[g["betas"][0] for g in groups]), D(*[g["betas"][1] for g in groups])      # (doubles: 1 - beta is formed in double)
            rc = L.lib().eda_grad_sumsq_f32(fl.flat_grad.data_ptr(), n, self.ws.data_ptr(), self.norm.data_ptr(), self.n, steps, b1, b2,
                                            self.lr_host.data_ptr(), self.lr_dev.data_ptr(), stream)
            L.check(rc, "eda_grad_sumsq_f32")
            rc = L.lib().eda_adamw_flat_f32(
                fl.flat_param.data_ptr(), fl.flat_grad.data_ptr(), n, self.n, LG(*self.lo), LG(*self.hi),
                P(*[s["exp_avg"].data_ptr() for s in sts]), P(*[s["exp_avg_sq"].data_ptr() for s in sts]), steps,
                self.lr_dev.data_ptr(), b1, b2,
                F(*[g["eps"] for g in groups]), F(*[g["weight_decay"] for g in groups]), self.norm.data_ptr(), float(max_norm),
                float(pre_scale), stream)
            L.check(rc, "eda_adamw_flat_f32")
        return self.norm * pre_scale if pre_scale != 1.0 else self.norm


def reserve_cus_for_collectives(cus=32):
    """N > 1: keep `cus` CUs out of the furthest point sampler's co-residency plan (include/eda_hip.h:
    eda_fps_set_cu_reserve) -- RCCL's channel workgroups spin on their peers like the sampler's workgroups spin on each
    other, and both must be resident at the same time when the sampling of batch i+1 runs underneath step i's
    all-reduce."""
    from . import _lib
    _lib.check(_lib.lib().eda_fps_set_cu_reserve(int(cus)), "eda_fps_set_cu_reserve")


def sampler_without_co_residency():
    """N > 1: furthest point sampling of the large scenes on the single-workgroup bucket sampler (include/eda_hip.h:
    eda_fps_set_policy(EDA_FPS_BUCKET)) -- no workgroup of this library then waits for another one, whatever RCCL's
    channel kernels occupy.  (The default policy repairs a cluster launch that was not co-resident, but only after its
    spin limit: ~1 s.)"""
    from . import _lib
    _lib.check(_lib.lib().eda_fps_set_policy(2), "eda_fps_set_policy")


def broadcast_parameters(module, src=0):
    """DDP constructor semantics: every rank starts from rank `src`'s weights and buffers."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def shard_scene_seeds(global_batch, rank, world):
    """Scene indices of this rank (DistributedSampler-equivalent over the synthetic generator)."""
    per = global_batch // world
    return list(range(rank * per, (rank + 1) * per))


def reference_lr_groups(name):
    """The reference's optimiser groups (main_utils.py:277-305)."""
    if "backbone_net" in name:
        return "backbone_net"
    if "text_encoder" in name:
        return "text_encoder"
    return "base"
