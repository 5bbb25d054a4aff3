"""Data-parallel gradient exchange for the EDITOR hot path (kernel row C1).

One process per GPU; the only collective per step is a SUM all-reduce of the trainable gradients
(reference: DistributedDataParallel(..., find_unused_parameters=True), engine/processor.py:47-50).
MI355X-first: xGMI is point-to-point, so the payload (475.7 MB fp32) is sent as a few LARGE flat buckets
(default 64 MiB, far above NCCL-on-NVSwitch's 25 MB habit) launched on RCCL's own stream the moment the
last gradient of a bucket is produced - the block-granular autograd nodes of editor_amd.functional make
gradients ready block by block, so the exchange overlaps the remaining backward.

Unused parameters (BACKBONE.base.fc, the head not selected by cfg.MODEL.AL) never produce a gradient; the
first step discovers the set and order of live parameters instead of traversing the graph every step.
Works with any backend (`nccl` == RCCL on ROCm, `gloo` in the CPU tests).
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, module, bucket_bytes=64 << 20, process_group=None, force=False):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())   # force: run the exchange on 1 rank too
        self.bucket_bytes = bucket_bytes
        self.params = [p for p in module.parameters() if p.requires_grad]
        self._order = []                 # discovery: params in the order their grads became ready
        self._seen = set()
        self._buckets = None             # list of dict(params, flat, pending, handle)
        self._slot = {}                  # id(param) -> (bucket index)
        self._handles = []
        # RCCL averages in the collective itself; gloo (CPU tests) only sums
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)

    # ------------------------------------------------------------------------------------------
    def _hook(self, p):
        if not self.active:
            return
        if self._buckets is None:
            if id(p) not in self._seen:
                self._seen.add(id(p))
                self._order.append(p)
            return
        bi = self._slot.get(id(p))
        if bi is None:
            return
        b = self._buckets[bi]
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def _launch(self, b):
        torch._foreach_copy_(b["views"], [p.grad for p in b["params"]])
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        b["handle"] = dist.all_reduce(b["flat"], op=op, group=self.group, async_op=True)
        self._handles.append(b)

    def _build(self):
        buckets, cur, size = [], [], 0
        for p in self._order:
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > self.bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            buckets.append(cur)
        self._buckets = []
        for bi, ps in enumerate(buckets):
            flat = torch.empty(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
            views, off = [], 0
            for p in ps:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
                self._slot[id(p)] = bi
            self._buckets.append(dict(params=ps, flat=flat, views=views, pending=len(ps), handle=None))

    # ------------------------------------------------------------------------------------------
    def finalize(self):
        """Call after loss.backward(): waits for the in-flight buckets and leaves AVERAGED grads in .grad."""
        if not self.active:
            return
        if self._buckets is None:
            # discovery step: nothing was overlapped; build the buckets and reduce them now
            self._build()
            for b in self._buckets:
                self._launch(b)
        else:
            for b in self._buckets:            # a bucket whose params did not all fire (should not happen)
                if b["pending"] != 0 and b["handle"] is None and b["pending"] < len(b["params"]):
                    raise RuntimeError("GradReducer: a bucket received only part of its gradients")
        inv = 1.0 / self.world
        for b in self._handles:
            b["handle"].wait()
            if not self._avg:
                torch._foreach_mul_(b["views"], inv)
            torch._foreach_copy_([p.grad for p in b["params"]], b["views"])
            b["handle"] = None
            b["pending"] = len(b["params"])
        self._handles = []

    def broadcast_parameters(self, src=0):
        """Initial sync of parameters and buffers (what DDP's constructor does)."""
        if not self.active:
            return
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            dist.broadcast(t.data, src=src, group=self.group)


class FlatAllReduce:
    """Gradient exchange for a step whose forward + backward were captured into a hipGraph (bench.py, N > 1): the
    gradients then live at fixed addresses, so they are packed once-and-for-all into a few flat buckets (views built at
    the first call) and every step is: gather into the buckets, all-reduce (AVG on RCCL), scatter back.  No autograd
    hooks, hence no overlap with the backward - the price of taking the ~1100 launches of the step off the host."""

    def __init__(self, module, bucket_bytes=64 << 20, process_group=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.bucket_bytes = bucket_bytes
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self._avg = dist.get_backend(process_group) == "nccl"
        self._buckets = None

    def _build(self):
        live = [p for p in self.params if p.grad is not None]
        self._buckets, cur, size = [], [], 0
        for p in live + [None]:
            nbytes = 0 if p is None else p.grad.numel() * p.grad.element_size()
            if cur and (p is None or size + nbytes > self.bucket_bytes):
                flat = torch.empty(sum(q.grad.numel() for q in cur), dtype=cur[0].grad.dtype, device=cur[0].grad.device)
                views, off = [], 0
                for q in cur:
                    views.append(flat[off:off + q.grad.numel()].view_as(q.grad))
                    off += q.grad.numel()
                self._buckets.append((flat, views, [q.grad for q in cur]))
                cur, size = [], 0
            if p is not None:
                cur.append(p)
                size += nbytes

    @torch.no_grad()
    def __call__(self):
        if self._buckets is None:
            self._build()
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        works = []
        for flat, views, grads in self._buckets:
            torch._foreach_copy_(views, grads)
            works.append(dist.all_reduce(flat, op=op, group=self.group, async_op=True))
        for w, (flat, views, grads) in zip(works, self._buckets):
            w.wait()
            if not self._avg:
                flat.mul_(1.0 / self.world)
            torch._foreach_copy_(grads, views)



# ------------------------------------------------------------------------------------------------------------------
# Gradient buckets WRITTEN IN PLACE by the backward, all-reduced from inside it (eager or captured in a hipGraph)
# ------------------------------------------------------------------------------------------------------------------
class BlockSink:
    """Where one transformer block's backward (editor_amd.functional.TransformerBlockFn) writes its 12 parameter
    gradients: views of a flat bucket, in forward-argument order (None for absent biases)."""

    def __init__(self, owner, seg, views, params=None):
        self.owner, self.seg, self.views = owner, seg, views
        self.params = params if params is not None else [None] * len(views)

    def ln_pair(self, i):
        """(2, D) view over the adjacent [weight | bias] gradient slots of a LayerNorm (slots i, i + 1)."""
        w, b = self.views[i], self.views[i + 1]
        if w is None or b is None:
            return None
        assert b.data_ptr() == w.data_ptr() + w.numel() * w.element_size()
        return torch.as_strided(w, (2, w.numel()), (w.numel(), 1))

    def done(self):
        # The backward WROTE the slots (it hands autograd None for these parameters): make sure every parameter's .grad
        # still IS its slot - nn.Module.zero_grad() / `p.grad = None` by foreign code would otherwise leave .grad None and
        # the optimizer would silently skip 99 % of the weights.
        for p, v in zip(self.params, self.views):
            if p is not None and (p.grad is None or p.grad.data_ptr() != v.data_ptr()):
                p.grad = v
        self.owner.segment_done(self.seg)


class GradBuckets:
    """Data-parallel gradient exchange of the EDITOR training step (one all-reduce of the 118.9 M
    trainable gradients per step, overlapped with the backward; reference: DistributedDataParallel's reducer,
    engine/processor.py:47-50).

    * The parameters of every transformer block (16 blocks = 99 % of the bytes) have their `.grad` set ONCE to a view of
      a flat bucket; the block's backward writes dW / db / dgamma / dbeta straight into those views (no per-step
      allocation, no pack / unpack copies) and reports `segment_done`.  Segments are laid out in gradient-ready order
      (joint HMA block first, backbone block 0 last) and grouped into few LARGE buckets (xGMI is point-to-point: ring
      collectives are per-link bound, so 64 MiB and up, not NVSwitch-sized 25 MB).
    * When the last segment of a bucket is done its all-reduce is issued (async, on the process group's own stream) -
      from INSIDE the backward, so it overlaps the remaining blocks.  The same calls are capturable: bench.py captures
      forward + backward + these collectives + the fused SGD into one hipGraph per rank.
    * The remaining small parameters (heads, BatchNorm, REDUCE, embeddings, final norms: ~2 MB) come out of autograd as
      usual and go through one packed tail bucket in `finish()`, which also waits for everything in flight.
    Backends: nccl (= RCCL, AVG in the collective) and gloo (SUM + scale; CPU tests)."""

    def __init__(self, segments, tail_params, bucket_bytes=64 << 20, process_group=None, force=False, wire_dtype=None):
        """segments: list of (name, [12 parameters or None]) in gradient-READY order.  tail_params: the rest.
        wire_dtype: None / torch.float32 = the buckets travel as they are (fp32, 475.7 MB per step for EDITOR); torch.bfloat16 =
        16-bit exchange ("flat bucketed grads (bf16 or fp32)"; what DDP's bf16_compress_hook does): a ready
        bucket is cast into a persistent bf16 twin on the comm stream, the twin is all-reduced (half the bytes on the xGMI
        ring: 237.8 MB) and cast back into the fp32 slots the optimizer reads.  bfloat16 whatever the compute mode: fp32's
        range, so unscaled gradient sums cannot overflow on the wire.  The averaged gradient then equals the fp32 exchange to
        bf16 rounding (2^-9 relative per element; tests/test_ddp_gloo.py)."""
        if wire_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("GradBuckets: wire_dtype must be None / torch.float32 / torch.bfloat16")
        self.wire_dtype = torch.bfloat16 if wire_dtype == torch.bfloat16 else None
        self.group = process_group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if inited else 1
        self.active = inited and (self.world > 1 or force)
        self._avg = inited and dist.get_backend(process_group) == "nccl"
        self.segments = segments
        self.tail_params = [p for p in tail_params if p.requires_grad]
        # ---- bucket plan: consecutive segments until bucket_bytes is reached
        self.buckets = []                 # dict(flat, segs, pending, work)
        self.seg_bucket = {}
        cur, size = [], 0
        plan = []
        for si, (name, params) in enumerate(segments):
            nbytes = sum(p.numel() * 4 for p in params if p is not None)
            if cur and size + nbytes > bucket_bytes:
                plan.append(cur)
                cur, size = [], 0
            cur.append(si)
            size += nbytes
        if cur:
            plan.append(cur)
        self.sinks = {}
        for bi, segs in enumerate(plan):
            ps = [p for si in segs for p in segments[si][1] if p is not None]
            flat = torch.zeros(sum(p.numel() for p in ps), dtype=torch.float32, device=ps[0].device)
            off = 0
            for si in segs:
                views = []
                for p in segments[si][1]:
                    if p is None:
                        views.append(None)
                        continue
                    v = flat[off:off + p.numel()].view_as(p)
                    off += p.numel()
                    p.grad = v                       # the parameter's gradient IS the bucket slot, permanently
                    p._grad_sink = v                 # (FusedSGD.zero_grad restores it instead of dropping it)
                    p._grad_owner = self             # (... and tells these buckets that a new step begins)
                    views.append(v)
                self.sinks[si] = BlockSink(self, si, views, list(segments[si][1]))
                self.seg_bucket[si] = bi
            self.buckets.append(dict(flat=flat, segs=segs, pending=len(segs), work=None))
        self._tail = None                 # (flat, views, params, key) of the tail parameters that carry a gradient
        self._inflight = []
        self._written = set()             # segments whose slots were written since the last finish()

    def sink(self, seg_index):
        return self.sinks[seg_index]

    def reset_step(self):
        """A new step begins: forget which segments the previous backward wrote.  finish() does this; FusedSGD.zero_grad()
        does it too, so that a backward that raised half-way (out of memory, a NaN assert) or a loop that uses the in-place
        slots with a plain optimizer and never calls finish() does not poison every later step with the 'written twice'
        error.  Collectives already in flight are waited for and dropped."""
        self._written.clear()
        for b in self.buckets:
            b["pending"] = len(b["segs"])
        for b in self._inflight:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
        self._inflight = []

    def segment_done(self, si):
        # The slots are OVERWRITTEN by a backward, not accumulated into: a second backward before finish() / the optimizer
        # step (gradient accumulation, two forwards per step) would silently keep only the last micro-batch.
        if si in self._written:
            raise RuntimeError("GradBuckets: segment %r was written twice before finish() - gradient accumulation is not "
                               "supported with in-place gradient buckets (call finish() + optimizer.step() per backward)"
                               % (self.segments[si][0],))
        self._written.add(si)
        b = self.buckets[self.seg_bucket[si]]
        b["pending"] -= 1
        if b["pending"] == 0:
            b["pending"] = len(b["segs"])
            if self.active:
                self._launch(b)

    def _wire(self, b):
        """The tensor that travels: the bucket itself, or its persistent 16-bit twin (allocated once: a captured graph replays
        launches that hold its address)."""
        if self.wire_dtype is None:
            return b["flat"]
        w = b.get("wire")
        if w is None or w.numel() != b["flat"].numel():
            w = b["wire"] = torch.empty(b["flat"].numel(), dtype=self.wire_dtype, device=b["flat"].device)
        w.copy_(b["flat"])                   # fp32 -> bf16, on the stream the collective is issued from
        return w

    def _launch(self, b):
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if b["flat"].is_cuda:
            # The bucket's slots are written on TWO streams: LayerNorm / bias gradients on the main stream, the blocks' grouped
            # weight gradients on the side stream (editor_amd.functional).  The collective is issued from a third stream that
            # waits for both - the main stream itself never waits for the side stream here (joining it at every bucket
            # boundary stalled the backward by ~0.35 ms per bucket: 49.3 instead of 45.9 ms per step with a 1-rank group)
            from . import functional
            dev = b["flat"].device
            comm = self._comm_stream(dev)
            comm.wait_stream(torch.cuda.current_stream(dev))
            side = functional._SIDE.get(dev.index)
            if side is not None:
                comm.wait_stream(side)
            with torch.cuda.stream(comm):
                b["work"] = dist.all_reduce(self._wire(b), op=op, group=self.group, async_op=True)
        else:
            b["work"] = dist.all_reduce(self._wire(b), op=op, group=self.group, async_op=True)
        self._inflight.append(b)

    def _comm_stream(self, dev):
        st = getattr(self, "_comm", None)
        if st is None:
            st = self._comm = torch.cuda.Stream(device=dev)
        return st

    @torch.no_grad()
    def finish(self):
        """After loss.backward(): exchange the tail parameters and wait for every bucket (stream-level wait on RCCL)."""
        self._written.clear()
        if self.buckets and self.buckets[0]["flat"].is_cuda:
            from . import functional
            functional.join_side_stream(self.buckets[0]["flat"].device)    # deferred weight-gradient launches of the last blocks
        if not self.active:
            return
        # tail = the small parameters that carry a gradient THIS step (never-used ones - BACKBONE.base.fc, the head cfg.MODEL.AL
        # does not select - have none).  The set is re-checked every step outside a capture, so a parameter that starts
        # receiving gradients later is exchanged too; inside a captured graph the set is fixed by construction.
        capturing = self.tail_params and self.tail_params[0].is_cuda and torch.cuda.is_current_stream_capturing()
        live = [p for p in self.tail_params if p.grad is not None]
        key = tuple(id(p) for p in live)
        if self._tail is None or (not capturing and self._tail[3] != key):
            if capturing and self._tail is None:
                raise RuntimeError("GradBuckets.finish(): run one eager step before capturing (the tail bucket is built there)")
            dev = live[0].device if live else self.buckets[0]["flat"].device
            flat = torch.zeros(sum(p.numel() for p in live), dtype=torch.float32, device=dev)
            views, off = [], 0
            for p in live:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self._tail = (flat, views, live, key)
            self._tail_bucket = dict(flat=flat, work=None)        # persistent, like the block buckets: its 16-bit wire twin is
                                                                  # allocated ONCE (a fresh dict per step re-allocated it)
        if capturing and self._tail[3] != key:
            # the captured graph would copy from the EAGER step's tail set: parameters that have since gained / lost a gradient
            # would be exchanged from stale or missing tensors
            raise RuntimeError("GradBuckets.finish(): the set of small parameters that carry a gradient differs from the one the "
                               "last eager step built the tail bucket for; run one eager step with the final configuration "
                               "before capturing")
        flat, views, live, _ = self._tail
        if live:
            torch._foreach_copy_(views, [p.grad for p in live])
            self._launch(self._tail_bucket)
        inv = 1.0 / self.world
        if getattr(self, "_comm", None) is not None:
            torch.cuda.current_stream(self._comm.device).wait_stream(self._comm)     # (re-joins the issuing stream: capture)
        for b in self._inflight:
            b["work"].wait()
            if self.wire_dtype is not None:
                b["flat"].copy_(b["wire"])          # bf16 -> the fp32 slots the optimizer reads
            if not self._avg:
                b["flat"].mul_(inv)
            b["work"] = None
        self._inflight = []
        if live:
            torch._foreach_copy_([p.grad for p in live], views)

    def _broadcast_coalesced(self, tensors, src):
        """One broadcast per dtype of a flat copy (the ~440 parameters + buffers are a handful of collectives, not 440)."""
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        for dt, ts in by_dtype.items():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=self.group)
            off = 0
            for t in ts:
                t.data.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()

    @torch.no_grad()
    def broadcast_parameters(self, module, src=0):
        """Initial sync of parameters and buffers from rank `src` (what DDP's constructor does), coalesced; the 16-bit operand
        copies cached by an earlier forward are invalidated (the version counters do not move under `.data` writes)."""
        if not self.active:
            return
        self._broadcast_coalesced(list(module.parameters()) + list(module.buffers()), src)
        from . import functional
        functional.invalidate_weight_cache()

    @torch.no_grad()
    def broadcast_buffers(self, module, src=0):
        """DDP's per-forward `broadcast_buffers=True` (the reference constructs DDP with the default, train_net.py:63-64):
        BatchNorm running statistics follow rank `src`.  One small coalesced collective per dtype; capturable."""
        if not self.active:
            return
        bufs = [b for b in module.buffers() if b.numel()]
        if bufs:
            self._broadcast_coalesced(bufs, src)

    def describe(self):
        esz = 2 if self.wire_dtype is not None else 4
        return {"buckets": len(self.buckets) + 1, "bucket_mib": [round(b["flat"].numel() * 4 / 2 ** 20, 1) for b in self.buckets],
                "segments": len(self.segments), "wire_dtype": "bf16" if self.wire_dtype is not None else "f32",
                "wire_mb_per_step": round(sum(b["flat"].numel() for b in self.buckets) * esz / 1e6, 1)}


def graph_capture_kwargs(settle=0.3):
    """Keyword arguments for `torch.cuda.graph(...)` when the captured step contains RCCL collectives, after giving the process
    group's watchdog thread time to retire the work of the eager warm-up steps.

    ProcessGroupNCCL's watchdog polls the completion events of enqueued collectives from ITS thread (`hipEventQuery`).  Under
    the default capture mode ('global') any such call made by ANY thread while a capture is open fails with
    hipErrorStreamCaptureUnsupported, the watchdog rethrows and the process aborts - intermittently: it needs a warm-up
    collective that the watchdog has not reaped yet when the capture starts (measured: 1 run in 6 of tools/ddp_selfcheck.py).
    'thread_local' restricts the check to the capturing thread; collectives issued INSIDE the capture are not handed to the
    watchdog at all.  No process group: nothing to do."""
    import time
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {}
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    time.sleep(settle)                      # (the watchdog's polling interval is 100 ms)
    return {"capture_error_mode": "thread_local"}

