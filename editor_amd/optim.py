"""Fused SGD for the EDITOR training step (row N4 of SURVEY.md 8(f)): the per-parameter groups of
solver/make_optimizer.py:4-29 (SGD, momentum 0.9, weight decay 1e-4, lr x2 for biases) applied to all ~200 parameter
tensors in ONE HIP launch (editor_sgd_multi) instead of torch's three foreach passes.

The object quacks like the torch.optim.SGD the reference builds: `param_groups` is one dict per trainable parameter
(make_optimizer.py:19 appends a group per parameter) carrying 'params', 'lr', 'weight_decay', 'momentum', so the
reference's scheduler (solver/scheduler.py:76-80 writes param_group['lr']) and amp.GradScaler.step (iterates
param_groups) drive it unchanged; `state_dict()` / `load_state_dict()` use torch.optim.SGD's layout
({'state': {i: {'momentum_buffer'}}, 'param_groups'}).

hipGraph-replay safety: the kernel reads lr / weight decay from per-tensor DEVICE tables.  Hyper-parameter changes
are pushed into those tables in place by `sync_param_groups()` (called by step() outside capture, and by
editor_amd.solver's scheduler), so a captured step replays with the new values; a change made while a stream is
capturing is refused instead of being baked into the graph."""
import torch

from . import _lib


class FusedSGD:
    def __init__(self, named_params, base_lr=1e-3, weight_decay=1e-4, bias_lr_factor=2.0, weight_decay_bias=1e-4,
                 momentum=0.9, shadow_bf16=True, shadow_dtype=None, split_pairs=False, check_overflow=None):
        self.params = [(n, p) for n, p in named_params if p.requires_grad]
        if not self.params:
            raise ValueError("FusedSGD: no trainable parameters")
        dev = self.params[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("FusedSGD updates parameters with a HIP kernel: move the model to the GPU first")
        self.device = dev
        self.momentum = float(momentum)
        self.defaults = dict(lr=base_lr, momentum=momentum, weight_decay=weight_decay, dampening=0, nesterov=False)
        chunk = _lib.lib().cdll.editor_sgd_chunk_elems()
        numel, chunk_t, chunk_o = [], [], []
        self.param_groups = []
        for i, (name, p) in enumerate(self.params):
            is_bias = "bias" in name                                   # make_optimizer.py:12-14
            self.param_groups.append({"params": [p], "name": name,
                                      "lr": base_lr * bias_lr_factor if is_bias else base_lr,
                                      "weight_decay": weight_decay_bias if is_bias else weight_decay,
                                      "momentum": self.momentum, "dampening": 0, "nesterov": False})
            numel.append(p.numel())
            for off in range(0, p.numel(), chunk):
                chunk_t.append(i)
                chunk_o.append(off)
        n = len(self.params)
        self.bufs = [torch.zeros_like(p, memory_format=torch.contiguous_format) for _, p in self.params]
        self.lr = torch.zeros(n, dtype=torch.float32, device=dev)
        self.wd = torch.zeros(n, dtype=torch.float32, device=dev)
        self._hyper_host = torch.zeros(2, n, dtype=torch.float32).pin_memory()
        self._hyper_cached = None
        self.sync_param_groups()
        self.numel = torch.tensor(numel, dtype=torch.int64, device=dev)
        self.chunk_t = torch.tensor(chunk_t, dtype=torch.int32, device=dev)
        self.chunk_o = torch.tensor(chunk_o, dtype=torch.int64, device=dev)
        self.nchunks = len(chunk_t)
        # 16-bit shadows of the GEMM weights (>= 2-D parameters): written by the update kernel itself and handed to the
        # operand cache of editor_amd.functional, instead of one cast launch per weight at the next forward
        if shadow_dtype is None:
            shadow_dtype = torch.bfloat16 if shadow_bf16 else None
        self.shadow_dtype = shadow_dtype
        self.shadows = [torch.empty(p.shape, dtype=shadow_dtype, device=dev) if (shadow_dtype is not None and p.dim() >= 2)
                        else None for _, p in self.params]
        self.h_ptrs = torch.tensor([0 if h is None else h.data_ptr() for h in self.shadows], dtype=torch.int64, device=dev)
        # k-major (transposed) shadows of the 2-D GEMM weights for the dgrad products, refreshed by one multi-tensor
        # transpose launch after the update (editor_transpose_multi)
        self.shadows_t = [torch.empty(p.shape[1], p.shape[0], dtype=shadow_dtype, device=dev)
                          if (h is not None and p.dim() == 2 and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0
                              and p.numel() >= (1 << 18)) else None for (_, p), h in zip(self.params, self.shadows)]
        tt = [i for i, h in enumerate(self.shadows_t) if h is not None]
        self._tr = None
        if tt:
            tile_t, tile_r, tile_c = [], [], []
            for j, i in enumerate(tt):
                r, c = self.params[i][1].shape
                for a in range(r // 64):
                    for b_ in range(c // 64):
                        tile_t.append(j); tile_r.append(a); tile_c.append(b_)
            i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
            self._tr = dict(src=torch.tensor([self.shadows[i].data_ptr() for i in tt], dtype=torch.int64, device=dev),
                            dst=torch.tensor([self.shadows_t[i].data_ptr() for i in tt], dtype=torch.int64, device=dev),
                            rows=i32([self.params[i][1].shape[0] for i in tt]), cols=i32([self.params[i][1].shape[1] for i in tt]),
                            tile_t=i32(tile_t), tile_r=i32(tile_r), tile_c=i32(tile_c), n=len(tile_t))
        # split-precision weight pairs (COMPUTE_DTYPE 'f16x2'): hi / lo halves of w * ops.SPLIT_WSCALE for the forward products,
        # refreshed by ONE launch after the update (editor_split_multi over the same chunk tables)
        self.pairs = None
        if split_pairs:
            self.pairs = [(torch.empty(p.shape, dtype=torch.float16, device=dev), torch.empty(p.shape, dtype=torch.float16, device=dev))
                          if p.dim() >= 2 else None for _, p in self.params]
            self.hi_ptrs = torch.tensor([0 if pr is None else pr[0].data_ptr() for pr in self.pairs], dtype=torch.int64, device=dev)
            self.lo_ptrs = torch.tensor([0 if pr is None else pr[1].data_ptr() for pr in self.pairs], dtype=torch.int64, device=dev)
        # overflow protocol of amp.GradScaler.step (engine/processor.py:94-96): one pass over the gradients sets `_found`
        # BEFORE the update, and the update kernel does nothing when it is set - parameters, momentum and shadows keep their
        # values.  Default: on in f16 modes (half gradients can overflow), off for bf16 / fp32 (fp32 exponent range).
        self.check_overflow = (shadow_dtype == torch.float16) if check_overflow is None else bool(check_overflow)
        self._found = torch.zeros(1, dtype=torch.int32, device=dev)
        self.grad_scaler = None              # DeviceGradScaler.step() attaches itself: gradients then carry its loss scale
        self.p_ptrs = torch.tensor([p.data_ptr() for _, p in self.params], dtype=torch.int64, device=dev)
        self.m_ptrs = torch.tensor([b.data_ptr() for b in self.bufs], dtype=torch.int64, device=dev)
        # gradient tensors are new every step: their addresses go to the device through double-buffered pinned staging
        # (an event per buffer keeps the host from overwriting a table whose async copy has not executed yet)
        self._g_host = [torch.zeros(n, dtype=torch.int64).pin_memory() for _ in range(2)]
        self._g_dev = [torch.zeros(n, dtype=torch.int64, device=dev) for _ in range(2)]
        self._g_host_cap = torch.zeros(n, dtype=torch.int64).pin_memory()   # table of a captured step
        self._g_dev_cap = torch.zeros(n, dtype=torch.int64, device=dev)
        self._ev = [None, None]
        self._slot = 0
        self._keep = [None, None]
        # set by the update kernel when it meets a non-finite gradient element (f16 mode: the static loss scale of
        # editor_amd.functional overflowed half somewhere in the backward); read with found_inf()
        self._nonfinite = torch.zeros(1, dtype=torch.int32, device=dev)

    # -- torch.optim.Optimizer surface --------------------------------------------------------------------
    def zero_grad(self, set_to_none=True):
        owners = []
        for _, p in self.params:
            sink = getattr(p, "_grad_sink", None)
            if sink is not None:
                p.grad = sink                    # gradient lives in an all-reduce bucket slot that the backward overwrites
                owner = getattr(p, "_grad_owner", None)
                if owner is not None and all(owner is not o for o in owners):
                    owners.append(owner)
                    owner.reset_step()           # a new step: a backward that raised half-way must not poison this one
            elif set_to_none or p.grad is None:
                p.grad = None
            else:
                p.grad.zero_()

    def sync_param_groups(self):
        """Push param_groups' lr / weight_decay into the device tables the kernel reads (in place: replay safe)."""
        vals = ([float(g["lr"]) for g in self.param_groups], [float(g["weight_decay"]) for g in self.param_groups])
        if vals == self._hyper_cached:
            return False
        for g in self.param_groups:
            if float(g.get("momentum", self.momentum)) != self.momentum or g.get("nesterov") or g.get("dampening"):
                raise NotImplementedError("FusedSGD: one momentum for all groups, no dampening / nesterov "
                                          "(solver/make_optimizer.py:24 uses neither)")
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FusedSGD: learning rate / weight decay changed during hipGraph capture; "
                               "change it between replays (the tables are read from device memory)")
        torch.cuda.current_stream(self.device).synchronize()     # the pinned staging row may still be in flight
        self._hyper_host[0] = torch.tensor(vals[0], dtype=torch.float32)
        self._hyper_host[1] = torch.tensor(vals[1], dtype=torch.float32)
        self.lr.copy_(self._hyper_host[0], non_blocking=True)
        self.wd.copy_(self._hyper_host[1], non_blocking=True)
        self._hyper_cached = vals
        return True

    def found_inf(self, reset=True):
        """True if any step since the last reset saw an inf / nan gradient (synchronises: call it every N steps, as the
        reference's GradScaler bookkeeping does once per step, engine/processor.py:95-96).

        Whether the offending step was APPLIED depends on the mode: with `check_overflow` on (the f16 / f16x2 default, and
        whenever a DeviceGradScaler drives the step) the gradients are checked before the update and the update kernel wrote
        nothing - parameters, momentum and 16-bit shadows are those of the previous step.  With `check_overflow` off (bf16 / f32
        default) the flag is raised by the update kernel ITSELF while it applies the inf / nan gradient: the weights are
        poisoned and the caller must restore a checkpoint (or construct the optimizer with check_overflow=True).
        `self.check_overflow` says which.  With the static in-backward scale (cfg.MODEL.GRAD_SCALE) halve it on a hit; a
        DeviceGradScaler backs its own scale off without the host."""
        bad = bool(self._nonfinite.item())
        if reset:
            self._nonfinite.zero_()
        return bad

    def set_lr(self, lr):
        """lr: one float for every group, or one value per group (same order as param_groups)."""
        if not isinstance(lr, (list, tuple)):
            lr = [lr] * len(self.param_groups)
        for g, v in zip(self.param_groups, lr):
            g["lr"] = float(v)
        self.sync_param_groups()

    def state_dict(self):
        groups = []
        for i, g in enumerate(self.param_groups):
            d = {k: v for k, v in g.items() if k != "params"}
            d["params"] = [i]
            groups.append(d)
        return {"state": {i: {"momentum_buffer": b.detach().clone()} for i, b in enumerate(self.bufs)},
                "param_groups": groups}

    def load_state_dict(self, sd):
        if len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("FusedSGD.load_state_dict: parameter group count differs")
        for g, s in zip(self.param_groups, sd["param_groups"]):
            for k, v in s.items():
                if k != "params":
                    g[k] = v
        for i, b in enumerate(self.bufs):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is not None and st.get("momentum_buffer") is not None:
                b.copy_(st["momentum_buffer"])
            else:
                b.zero_()
        self.sync_param_groups()

    @torch.no_grad()
    def step(self):
        from . import functional
        functional.join_side_stream(self.device)     # grouped weight gradients of the last block(s) (side stream)
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            # hipGraph capture of the whole step: the gradients live at fixed addresses of the graph's memory pool, so the
            # pointer table is written once into a dedicated pinned buffer whose (captured) upload every replay repeats;
            # no events, no host waits
            host, dev_tab, k = self._g_host_cap, self._g_dev_cap, None    # (allocated up front: pinning is not capturable)
        else:
            self.sync_param_groups()
            k = self._slot
            self._slot ^= 1
            if self._ev[k] is not None:
                self._ev[k].synchronize()                                # the copy issued two steps ago has executed
            host, dev_tab = self._g_host[k], self._g_dev[k]
        grads = []
        for i, (_, p) in enumerate(self.params):
            g = p.grad
            if g is None:
                host[i] = 0
            else:
                if not g.is_contiguous():
                    g = g.contiguous()
                grads.append(g)
                host[i] = g.data_ptr()
        if capturing:
            self._keep_cap = grads
        else:
            self._keep[k] = grads                                        # keep the tensors alive until the kernel ran
        dev_tab.copy_(host, non_blocking=True)
        # momentum buffers start at zero, so mu*0 + g' == g' reproduces torch's first-step `buf = clone(g')` exactly
        # without a "first step" flag (a flag passed by value would be baked into a captured graph)
        sc = self.grad_scaler
        check = self.check_overflow or sc is not None
        with torch.cuda.device(self.device):
            if check:
                found = sc._found if sc is not None else self._found
                if sc is None:
                    self._found.zero_()              # (the scaler's update kernel clears its own flag)
                _lib.call("editor_grad_check_multi", dev_tab, self.chunk_t, self.chunk_o, self.numel, self.nchunks, found,
                          self._nonfinite)
            self._launch_update(dev_tab, None if check else self._nonfinite, sc._inv_scale if sc is not None else None,
                                found if check else None)
            if self.pairs is not None:
                from . import ops
                _lib.call("editor_split_multi", self.p_ptrs, self.hi_ptrs, self.lo_ptrs, self.chunk_t, self.chunk_o, self.numel,
                          self.nchunks, float(ops.SPLIT_WSCALE))
        if not capturing:
            self._ev[k] = torch.cuda.Event()
            self._ev[k].record()
        # the kernel wrote the parameters behind autograd's back: tell the operand cache (version counters did not
        # move) and hand it the shadows of the weights that just got a gradient
        from . import functional
        functional.invalidate_weight_cache()
        if self.shadow_dtype is not None:
            if self._tr is not None:
                tr = self._tr
                with torch.cuda.device(self.device):
                    _lib.call("editor_transpose_multi", tr["src"], tr["dst"], tr["rows"], tr["cols"], tr["tile_t"], tr["tile_r"],
                              tr["tile_c"], tr["n"])
            functional.install_weight_copies((p, h) for (_, p), h in zip(self.params, self.shadows)
                                             if h is not None and p.grad is not None)
            functional.install_weight_copies(((p, h) for (_, p), h in zip(self.params, self.shadows_t)
                                              if h is not None and p.grad is not None), transposed=True)
        if self.pairs is not None:
            functional.install_weight_pairs((p, pr[0], pr[1]) for (_, p), pr in zip(self.params, self.pairs) if pr is not None)


    def _launch_update(self, g_tab, nonfinite, inv_scale, skip):
        _lib.call("editor_sgd_multi", self.p_ptrs, g_tab, self.m_ptrs, self.chunk_t, self.chunk_o, self.numel,
                  self.lr, self.wd, float(self.momentum), self.nchunks, self.h_ptrs,
                  2 if self.shadow_dtype == torch.float16 else 1, nonfinite, inv_scale, skip)


class FusedAdamW(FusedSGD):
    """torch.optim.AdamW as solver/make_optimizer.py:23-24 builds it (OPTIMIZER_NAME 'AdamW': the per-parameter groups keep their own
    lr / weight decay, betas (0.9, 0.999), eps 1e-8, decoupled weight decay) in ONE HIP launch (editor_adamw_multi) - everything else
    (gradient buckets, 16-bit shadows, overflow protocol, hipGraph-replay safety: the step count lives on the device) is FusedSGD's.
    `state_dict()` uses torch.optim.AdamW's layout: {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups'}."""

    def __init__(self, named_params, base_lr=1e-3, weight_decay=1e-2, bias_lr_factor=1.0, weight_decay_bias=None, betas=(0.9, 0.999),
                 eps=1e-8, **kw):
        super().__init__(named_params, base_lr=base_lr, weight_decay=weight_decay, bias_lr_factor=bias_lr_factor,
                         weight_decay_bias=weight_decay if weight_decay_bias is None else weight_decay_bias, momentum=0.0, **kw)
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self.defaults = dict(lr=base_lr, betas=self.betas, eps=self.eps, weight_decay=weight_decay, amsgrad=False)
        for g in self.param_groups:
            for k in ("momentum", "dampening", "nesterov"):
                g.pop(k, None)
            g.update(betas=self.betas, eps=self.eps, amsgrad=False)
        self.bufs_sq = [torch.zeros_like(b) for b in self.bufs]                       # exp_avg_sq (self.bufs = exp_avg)
        self.v_ptrs = torch.tensor([b.data_ptr() for b in self.bufs_sq], dtype=torch.int64, device=self.device)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=self.device)

    def _launch_update(self, g_tab, nonfinite, inv_scale, skip):
        _lib.call("editor_adamw_multi", self.p_ptrs, g_tab, self.m_ptrs, self.v_ptrs, self.chunk_t, self.chunk_o, self.numel,
                  self.lr, self.wd, self.betas[0], self.betas[1], self.eps, self.step_count, self.nchunks, self.h_ptrs,
                  2 if self.shadow_dtype == torch.float16 else 1, nonfinite, inv_scale, skip)

    def state_dict(self):
        sd = super().state_dict()
        t = self.step_count.detach().clone().cpu().view(())
        sd["state"] = {i: {"step": t.clone(), "exp_avg": m.detach().clone(), "exp_avg_sq": v.detach().clone()}
                       for i, (m, v) in enumerate(zip(self.bufs, self.bufs_sq))}
        return sd

    def load_state_dict(self, sd):
        state = sd["state"]
        super().load_state_dict({"state": {}, "param_groups": sd["param_groups"]})
        t = 0.0
        for i, (m, v) in enumerate(zip(self.bufs, self.bufs_sq)):
            st = state.get(i, state.get(str(i)))
            if st is None:
                m.zero_(); v.zero_()
                continue
            m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
            t = max(t, float(st["step"]))
        self.step_count.fill_(t)


class DeviceGradScaler:
    """torch.cuda.amp.GradScaler (engine/processor.py:60,94-96) with its whole state on the device, so that a step using it
    can be captured into a hipGraph and replayed: no `.item()`, no host branch.

        scaler = DeviceGradScaler(device)            # init_scale 2**15, growth 2, backoff 0.5, growth_interval 2000
        scaler.scale(loss).backward()                # loss * scale (a device scalar): gradients come out scaled
        scaler.step(optimizer)                       # FusedSGD: overflow check -> unscale + update, or nothing
        scaler.update()                              # overflow: scale *= backoff; else every growth_interval steps *= growth

    Use it with cfg.MODEL.GRAD_SCALE = 1 (the static in-backward scale of editor_amd.functional off): the loss scale then
    rides through the backward exactly as the reference's does."""

    def __init__(self, device, init_scale=2.0 ** 15, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.device = torch.device(device)
        self._scale = torch.full((1,), float(init_scale), dtype=torch.float32, device=self.device)
        self._inv_scale = torch.full((1,), 1.0 / float(init_scale), dtype=torch.float32, device=self.device)
        self._tracker = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._found = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.enabled = enabled

    def scale(self, loss):
        return loss * self._scale.to(loss.dtype).view(()) if self.enabled else loss

    def step(self, optimizer):
        if not self.enabled:
            return optimizer.step()
        if not isinstance(optimizer, FusedSGD):
            raise TypeError("DeviceGradScaler drives editor_amd.optim.FusedSGD (its update kernel unscales and skips on the device)")
        optimizer.grad_scaler = self
        try:
            optimizer.step()
        finally:
            optimizer.grad_scaler = None

    def update(self):
        if self.enabled:
            with torch.cuda.device(self.device):
                _lib.call("editor_scaler_update", self._scale, self._inv_scale, self._tracker, self._found, self.growth_factor,
                          self.backoff_factor, self.growth_interval)

    def get_scale(self):
        """Host copy of the scale (synchronises; logging only)."""
        return float(self._scale.item())

    def state_dict(self):
        return {"scale": self.get_scale(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(self._tracker.item())}

    def load_state_dict(self, sd):
        self._scale.fill_(float(sd["scale"]))
        self._inv_scale.fill_(1.0 / float(sd["scale"]))
        self._tracker.fill_(int(sd.get("_growth_tracker", 0)))
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])
