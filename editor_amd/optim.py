"""Fused SGD for the EDITOR training step (row N4 of SURVEY.md 8(f)): the per-parameter groups of
solver/make_optimizer.py:4-29 (SGD, momentum 0.9, weight decay 1e-4, lr x2 for biases) applied to all ~200 parameter
tensors in ONE HIP launch (editor_sgd_multi) instead of torch's three foreach passes."""
import torch

from . import _lib


class FusedSGD:
    def __init__(self, named_params, base_lr=1e-3, weight_decay=1e-4, bias_lr_factor=2.0, weight_decay_bias=1e-4,
                 momentum=0.9, shadow_bf16=True):
        self.params = [(n, p) for n, p in named_params if p.requires_grad]
        dev = self.params[0][1].device
        self.device = dev
        self.momentum = momentum
        self.first = True
        chunk = _lib.lib().cdll.editor_sgd_chunk_elems()
        lrs, wds, numel, chunk_t, chunk_o = [], [], [], [], []
        for i, (name, p) in enumerate(self.params):
            is_bias = "bias" in name                                   # make_optimizer.py:12-14
            lrs.append(base_lr * bias_lr_factor if is_bias else base_lr)
            wds.append(weight_decay_bias if is_bias else weight_decay)
            numel.append(p.numel())
            for off in range(0, p.numel(), chunk):
                chunk_t.append(i)
                chunk_o.append(off)
        self.bufs = [torch.zeros_like(p, memory_format=torch.contiguous_format) for _, p in self.params]
        self.lr = torch.tensor(lrs, dtype=torch.float32, device=dev)
        self.wd = torch.tensor(wds, dtype=torch.float32, device=dev)
        self.numel = torch.tensor(numel, dtype=torch.int64, device=dev)
        self.chunk_t = torch.tensor(chunk_t, dtype=torch.int32, device=dev)
        self.chunk_o = torch.tensor(chunk_o, dtype=torch.int64, device=dev)
        self.nchunks = len(chunk_t)
        # bf16 shadows of the GEMM weights (>= 2-D parameters): written by the update kernel itself and handed to the
        # operand cache of editor_amd.functional, instead of one cast launch per weight at the next forward
        self.shadows = [torch.empty(p.shape, dtype=torch.bfloat16, device=dev) if (shadow_bf16 and p.dim() >= 2) else None
                        for _, p in self.params]
        self.h_ptrs = torch.tensor([0 if h is None else h.data_ptr() for h in self.shadows], dtype=torch.int64, device=dev)
        self.p_ptrs = torch.tensor([p.data_ptr() for _, p in self.params], dtype=torch.int64, device=dev)
        self.m_ptrs = torch.tensor([b.data_ptr() for b in self.bufs], dtype=torch.int64, device=dev)
        # gradient tensors are new every step: their addresses go to the device through double-buffered pinned staging
        # (an event per buffer keeps the host from overwriting a table whose async copy has not executed yet)
        self._g_host = [torch.zeros(len(self.params), dtype=torch.int64).pin_memory() for _ in range(2)]
        self._g_dev = [torch.zeros(len(self.params), dtype=torch.int64, device=dev) for _ in range(2)]
        self._g_host_cap = torch.zeros(len(self.params), dtype=torch.int64).pin_memory()   # table of a captured step
        self._g_dev_cap = torch.zeros(len(self.params), dtype=torch.int64, device=dev)
        self._ev = [None, None]
        self._slot = 0
        self._keep = [None, None]

    def zero_grad(self, set_to_none=True):
        for _, p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            # hipGraph capture of the whole step: the gradients live at fixed addresses of the graph's memory pool, so the
            # pointer table is written once into a dedicated pinned buffer whose (captured) upload every replay repeats;
            # no events, no host waits
            host, dev_tab, k = self._g_host_cap, self._g_dev_cap, None    # (allocated up front: pinning is not capturable)
        else:
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
        _lib.call("editor_sgd_multi", self.p_ptrs, dev_tab, self.m_ptrs, self.chunk_t, self.chunk_o, self.numel,
                  self.lr, self.wd, float(self.momentum), 1 if self.first else 0, self.nchunks, self.h_ptrs)
        if not capturing:
            self._ev[k] = torch.cuda.Event()
            self._ev[k].record()
        self.first = False
        # the kernel wrote the parameters behind autograd's back: tell the bf16 operand cache (version counters did not
        # move) and hand it the shadows of the weights that just got a gradient
        from . import functional
        functional.invalidate_weight_cache()
        functional.install_weight_copies((p, h) for (_, p), h in zip(self.params, self.shadows)
                                         if h is not None and p.grad is not None)
