"""The software-pipelined training step: three HIP graphs on two streams, with the double buffers a data loader
feeds.

The reference's loop (`main_utils.py:463-470`: `for batch in loader: loss = model(batch); backward; step`) runs
everything of a batch inside its step.  Two pieces of a BeaUTyDETR step depend on the INPUT alone -- the furthest
point sampling of SA1 (3 ms of dependent rounds on ~100 of the 256 CUs) and the frozen RoBERTa text encoder -- so they
can be computed for batch i+1 on a second stream while step i trains, and handed to the model through the reference's
own hooks (`PointnetSAModuleVotes.forward(xyz, features, inds)`, pointnet2_modules.py:217-235; the hidden states the
frozen encoder would produce).  This class owns what that needs:

    nxt   static buffers of the batch that will be trained NEXT (the loader writes them: `step(next_batch=...)`)
    cur   static buffers of the batch being trained (rotated from nxt by the first nodes of the point graph)
    inds_next / inds_cur, text_next / text_cur   the prefetched sampling indices and hidden states, same rotation

    main stream   [wait prefetch(i)] point graph(i): rotate nxt -> cur, SA stack     | rest graph(i): encoder/decoder,
                                                                                        loss, backward, (clip + AdamW)
    side stream                      [wait rotation(i)] copy batch i+1 -> nxt, FPS(i+1), text encoder(i+1)

Every step executes exactly one sampling and one text-encoder pass (nothing is cached or skipped); with
`next_batch=None` the nxt buffers are left as they are (the same batch again).  tests/test_pipeline_gpu.py drives it
with alternating batches and checks, per step, that the indices used are the FPS of the batch being trained and that
the loss history equals eager training on the same sequence.
"""
import ctypes
import os

import torch

from . import attention, ext, pointnet2_utils


def _flat(batch):
    """The tensors of a batch dict in a fixed order (nested dict `tokenized` included)."""
    out = []
    for k in sorted(batch):
        v = batch[k]
        if isinstance(v, dict):
            out.extend(v[kk] for kk in sorted(v))
        elif torch.is_tensor(v):
            out.append(v)
    return out


def _clone(batch):
    return {k: ({kk: vv.clone() for kk, vv in v.items()} if isinstance(v, dict) else (v.clone() if torch.is_tensor(v) else v))
            for k, v in batch.items()}


def _rebuild(batch, tensors):
    """`batch` with its tensors replaced, in _flat() order, by `tensors`."""
    it = iter(tensors)
    out = {}
    for k in sorted(batch):
        v = batch[k]
        if isinstance(v, dict):
            out[k] = {kk: next(it) for kk in sorted(v)}
        elif torch.is_tensor(v):
            out[k] = next(it)
        else:
            out[k] = v
    return out


def _packed_like(tensors):
    """One byte buffer holding a contiguous copy-shaped view per tensor (256-byte aligned, any mix of dtypes): returns
    (buffer, views).  Rotating / handing over ALL of them is then one copy of the buffer instead of one memcpy node per
    tensor (a hipGraph memcpy node costs ~7 us of queue time around a few-KB copy: 24 of them opened every step)."""
    offs, total = [], 0
    for t in tensors:
        offs.append(total)
        total += (t.numel() * t.element_size() + 255) // 256 * 256
    buf = torch.zeros(max(total, 256), dtype=torch.uint8, device=tensors[0].device)
    views = [buf[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape) for o, t in zip(offs, tensors)]
    return buf, views


# TIMING EXPERIMENTS ONLY (the step then trains on stale prefetch results): EDA_TIMING_SKIP_SIDE=fps,text leaves the second
# stream's graphs out of step() -- what the main stream costs alone (profiles/r05_side_stream_interference.md)
_TIMING_SKIP = os.environ.get("EDA_TIMING_SKIP_SIDE", "").split(",")


def _side_stream(device):
    """The second stream.  EDA_SIDE_CU_MASK (experiments: profiles/r06_side_cu_mask.md) restricts it to a set of CUs
    (hipExtStreamCreateWithCUMask): "N" = N CUs spread evenly over the mask's 256 bits, "lowN" = the first N bits, "0x..." = the
    mask itself (bit i = CU i in the runtime's numbering).  The cluster sampler is told to plan for that many CUs."""
    spec = os.environ.get("EDA_SIDE_CU_MASK", "")
    if not spec:
        return torch.cuda.Stream()
    total = torch.cuda.get_device_properties(device).multi_processor_count
    if spec.startswith("0x"):
        mask = int(spec, 16)
    elif spec.startswith("low"):
        mask = (1 << int(spec[3:])) - 1
    else:
        n = int(spec)
        mask = 0
        for i in range(n):
            mask |= 1 << (i * total // n)
    n = bin(mask).count("1")
    words = (ctypes.c_uint32 * ((total + 31) // 32))(*[(mask >> (32 * i)) & 0xFFFFFFFF for i in range((total + 31) // 32)])
    hip = ctypes.CDLL("libamdhip64.so")
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), len(words), words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed (%d)" % rc)
    from . import _lib
    _lib.check(_lib.lib().eda_fps_set_cu_reserve(max(0, total - n)), "eda_fps_set_cu_reserve")
    return torch.cuda.ExternalStream(handle.value, device=device)


class PipelinedTrainStep:
    def __init__(self, model, first_batch, loss_fn, backward_fn, update_fn, *, stream=None, all_reduce=None,
                 split_update=False, prefetch="sa1", text_prefetch=True, after_loss=None, sa1_samples=2048,
                 post_stages=None):
        """model: BeaUTyDETR (train mode, text encoder frozen).  first_batch: dict of DEVICE tensors (the layout of
        every later batch; extra tensors, e.g. loss targets, ride along).  loss_fn(end_points, batch) -> scalar.  backward_fn(loss): backward + gradient gather (e.g. under
        FlatParams.deferred_wgrad()).  update_fn(): clip + optimizer step (capturable).  all_reduce(): eager
        collective between the backward graph and the update graph (N > 1; implies split_update).
        post_stages: instead of all_reduce, a list of (capturable_fn or None, eager_fn or None) pairs executed in order
        between the backward graph and the update graph -- each capturable_fn becomes its own small graph, each eager_fn
        is called on the host after it (bench.py: [(None, start all-reduce of range A), (flush + gather of range B, start
        all-reduce of range B and wait for both)]: the first collective runs underneath the second weight-gradient kernel).
        prefetch: "sa1" (SA1's sampling for the next batch), "geometry" (everything the backbone derives from the
        coordinates alone: Pointnet2Backbone.geometry) or None (sampling inside the step; then the text encoder runs for
        the CURRENT batch underneath the point backbone).  text_prefetch: the text encoder, too, works for the next batch.
        stream: the stream the graphs are captured / replayed on (warm-up eager steps must have run on it)."""
        prefetch_geometry = prefetch == "geometry"
        text_prefetch = bool(text_prefetch) and prefetch is not None
        self.model, self.loss_fn = model, loss_fn
        self.all_reduce = all_reduce
        split_update = split_update or all_reduce is not None or bool(post_stages)
        assert all_reduce is None or not post_stages, "all_reduce and post_stages are alternatives"
        dev = first_batch["point_clouds"].device
        self.main = stream or torch.cuda.current_stream()
        self.side = _side_stream(dev)
        mode = dict(capture_error_mode="thread_local")

        # ---- side stream, eager once: creates this stream's FPS workspace outside the capture (its sticky give-up flag
        # must not be re-zeroed by a captured fill), warms the text encoder's library kernels, and tells the shapes of
        # everything that is handed from the side stream to the step
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            xyz = first_batch["point_clouds"][..., 0:3].contiguous()
            geo_keys, warm = [], []
            if prefetch_geometry:
                geo = model.backbone_net.geometry(xyz)
                geo_keys, warm = list(geo.keys()), list(geo.values())
            elif prefetch is not None:
                warm = [pointnet2_utils.furthest_point_sample(xyz, sa1_samples)]
            text_warm = model.encode_text_frozen(first_batch["tokenized"]["input_ids"], first_batch["tokenized"]["attention_mask"])
        self.side.synchronize()
        # nxt / cur: ONE buffer each for the batch tensors, the prefetched indices and the prefetched hidden states -- the
        # rotation at the head of the point graph is one copy
        flat0 = _flat(first_batch)
        order = flat0 + warm + ([text_warm] if text_prefetch else [])
        self._pack_next, nv = _packed_like(order)
        self._pack_cur, cv = _packed_like(order)
        for v, t in zip(nv, order):
            v.copy_(t)
        self._pack_cur.copy_(self._pack_next)
        nb, ni = len(flat0), len(warm)
        self._nxt_flat, self._cur_flat = nv[:nb], cv[:nb]
        self.nxt, self.cur = _rebuild(first_batch, self._nxt_flat), _rebuild(first_batch, self._cur_flat)
        self.inds_next, self.inds_cur = nv[nb:nb + ni], cv[nb:nb + ni]
        self.text_prefetch = text_prefetch
        # the text encoder reads the NEXT batch's tokens when it is prefetched, else the current batch's
        tok = (self.nxt if text_prefetch else self.cur)["tokenized"]

        # ---- side stream: sampling (+ optionally all coordinate-only geometry) and text encoder of the NEXT batch ----
        self.g_fps, self.g_text = None, torch.cuda.CUDAGraph()
        if prefetch is not None:
            self.g_fps = torch.cuda.CUDAGraph()
            # the sampler as a prefetch: lean polling, so that its ~100 resident workgroups leave the memory system to the
            # step's kernels (include/eda_hip.h: eda_fps_set_background; a launch parameter, so the captured nodes keep it)
            from . import _lib
            _lib.check(_lib.lib().eda_fps_set_background(1), "eda_fps_set_background")
            try:
                with torch.cuda.graph(self.g_fps, stream=self.side, **mode):
                    xyz_next = self.nxt["point_clouds"][..., 0:3].contiguous()
                    if prefetch_geometry:
                        outs = list(model.backbone_net.geometry(xyz_next).values())
                    else:
                        outs = [pointnet2_utils.furthest_point_sample(xyz_next, sa1_samples)]
                    for v, t in zip(self.inds_next, outs):     # (side stream: off the critical path)
                        v.copy_(t)
            finally:
                _lib.check(_lib.lib().eda_fps_set_background(0), "eda_fps_set_background")
        with torch.cuda.graph(self.g_text, stream=self.side, **mode):
            hidden = model.encode_text_frozen(tok["input_ids"], tok["attention_mask"])
            if text_prefetch:
                self.text_next = nv[-1]
                self.text_next.copy_(hidden)
            else:
                self.text_next = hidden
        self.side.synchronize()
        if self.g_fps is not None:
            self.g_fps.replay()
        self.g_text.replay()
        torch.cuda.synchronize()
        self._pack_cur.copy_(self._pack_next)
        self.text_cur = cv[-1] if self.text_prefetch else self.text_next
        inputs_h = dict(self.cur)
        inputs_h["text_hidden"] = self.text_cur
        if prefetch_geometry:
            inputs_h["backbone_geometry"] = dict(zip(geo_keys, self.inds_cur))
        elif prefetch is not None:
            inputs_h["sa1_inds"] = self.inds_cur[0]
        self.inputs = inputs_h

        # ---- main stream: point graph | rest graph (| update graph) -----------------------------------------------------
        self.g_pts, self.g_rest = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        self.g_up = torch.cuda.CUDAGraph() if split_update else None
        pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(self.g_pts, pool=pool, stream=self.main, **mode):
            # rotate: the batch prefetched during the previous step (its tensors, sampling indices, hidden states) becomes
            # the batch of this step -- one copy of the packed buffer
            self._pack_cur.copy_(self._pack_next)
            attention.advance_dropout_state(dev)
            ep_static = model.forward_point_backbone(inputs_h)
        with torch.cuda.graph(self.g_rest, pool=pool, stream=self.main, **mode):
            self.loss = loss_fn(model.forward_rest(inputs_h, ep_static), self.cur)
            backward_fn(self.loss)
            if after_loss is not None:
                after_loss(self.loss)
            if self.g_up is None:
                update_fn()
        self.post = []
        for cap_fn, eager_fn in (post_stages or []):
            g = None
            if cap_fn is not None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, stream=self.main, **mode):
                    cap_fn()
            self.post.append((g, eager_fn))
        if self.g_up is not None:
            with torch.cuda.graph(self.g_up, pool=pool, stream=self.main, **mode):
                update_fn()
        self.ev_pts, self.ev_fps, self.ev_done, self.ev_text = (torch.cuda.Event() for _ in range(4))
        self.ev_fps.record(self.side)
        self.ev_done.record(self.main)

    def _feed(self, batch):
        """Copy `batch` (same layout as the first one) into the nxt buffers, on the side stream."""
        src = _flat(batch)
        for v, t in zip(self._nxt_flat, src):
            v.copy_(t)
        for t in src:                       # the copy runs on the side stream: keep the caller's memory alive until it is done
            if t.is_cuda:                   # (host / pinned sources have no stream bookkeeping: record_stream raises)
                t.record_stream(self.side)

    def step(self, next_batch=None):
        """Train on the batch fed by the previous call (the first batch initially); start the sampling / text encoding
        of `next_batch` underneath.  Returns the (static) loss tensor of this step."""
        caller = torch.cuda.current_stream()
        if caller != self.main:
            # the graphs were captured on self.main and the events order against it: replay there whatever the caller's
            # current stream is -- AFTER the work the caller has queued on its own stream, which is what
            # produced `next_batch` (the side stream's copies read it, and side only ever waits for main)
            self.main.wait_stream(caller)
            with torch.cuda.stream(self.main):
                return self.step(next_batch)
        cur = self.main
        # the point graph first: a replay call returns when its last node has been queued, which for the long graph is
        # close to its end on the GPU -- whatever the host issues before the point graph is time the main queue idles
        cur.wait_event(self.ev_fps)                # this batch's sampling / hidden states / data are in the nxt buffers
        self.g_pts.replay()
        self.ev_pts.record(cur)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ev_pts)      # the rotation has taken its copies: nxt may be overwritten
            if not self.text_prefetch:
                # the encoder works for THIS step (tokens of cur, rotated by the point graph), underneath the SA stack
                self.side.wait_event(self.ev_done)     # the previous step has consumed the hidden states
                self.g_text.replay()
                self.ev_text.record(self.side)
            if next_batch is not None:
                self._feed(next_batch)
            if self.g_fps is not None and "fps" not in _TIMING_SKIP:
                self.g_fps.replay()
            if self.text_prefetch and "text" not in _TIMING_SKIP:
                self.g_text.replay()
            self.ev_fps.record(self.side)          # (one event: the next point graph waits for all of it)
        if not self.text_prefetch:
            cur.wait_event(self.ev_text)
        self.g_rest.replay()
        self.ev_done.record(cur)
        if self.g_up is not None:
            if self.all_reduce is not None:
                self.all_reduce()
            for g, eager_fn in self.post:
                if g is not None:
                    g.replay()
                if eager_fn is not None:
                    eager_fn()
            self.g_up.replay()
        return self.loss

    def check(self):
        """Host-side health check of everything in the step that gives up instead of hanging (synchronises the device: call it
        once per epoch, or per step while debugging): raises if the multi-workgroup sampler ever gave up its spin (the indices
        of that step were not the FPS result) or an in-kernel BatchNorm statistics exchange timed out (the statistics are
        invalid from that point on, eda_amd/sync_bn.py)."""
        from . import sync_bn
        n = self.fps_status()
        if n:
            raise RuntimeError(f"furthest point sampling gave up its inter-workgroup spin in {n} workspace(s)")
        sync_bn.check()

    def fps_status(self):
        """Sticky give-up flag of the multi-workgroup sampler over all steps so far (0 = every sampling completed)."""
        return ext.fps_status(self.nxt["point_clouds"].device)
