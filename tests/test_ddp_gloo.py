"""world_size-2 gloo test of the gradient exchange (editor_amd.ddp.GradReducer) on CPU: bucketing,
discovery of unused parameters, averaging, overlap path == reference all-reduce of the same grads."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(64, 128)
        self.b = nn.Linear(128, 256)
        self.c = nn.Linear(256, 32)
        self.unused = nn.Linear(8, 8)            # like BACKBONE.base.fc: never in the graph

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from editor_amd.ddp import GradReducer
    torch.manual_seed(0)
    m = _Toy()
    red = GradReducer(m, bucket_bytes=64 * 1024)        # several buckets
    red.broadcast_parameters()
    ok = True
    for step in range(3):
        g = torch.Generator().manual_seed(100 * step + rank)
        x = torch.randn(16, 64, generator=g)
        m.zero_grad(set_to_none=True)
        m(x).pow(2).mean().backward()
        local = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        red.finalize()
        for n, p in m.named_parameters():
            if p.grad is None:
                ok &= n.startswith("unused")
                continue
            ref = local[n].clone()
            dist.all_reduce(ref)
            ok &= torch.allclose(p.grad, ref / world, atol=1e-6)
        ok &= len(red._buckets) >= 2
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_grad_reducer_world2():
    world = 2
    port = _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))


def _worker_flat(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from editor_amd.ddp import FlatAllReduce
    torch.manual_seed(0)
    m = _Toy()
    for p in m.parameters():
        dist.broadcast(p.data, src=0)
    flat = FlatAllReduce(m, bucket_bytes=64 * 1024)
    ok = True
    # static gradient tensors, as after a graph capture: allocate once, refill in place every step
    x = torch.randn(16, 64, generator=torch.Generator().manual_seed(rank))
    m(x).pow(2).mean().backward()
    for step in range(3):
        g = torch.Generator().manual_seed(100 * step + rank)
        for p in m.parameters():
            if p.grad is not None:
                p.grad.copy_(torch.randn(p.grad.shape, generator=g))
        local = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        flat()
        for n, p in m.named_parameters():
            if p.grad is None:
                ok &= n.startswith("unused")
                continue
            ref = local[n].clone()
            dist.all_reduce(ref)
            ok &= torch.allclose(p.grad, ref / world, atol=1e-6)
        ok &= len(flat._buckets) >= 2
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_all_reduce_world2():
    """The exchange used after a hipGraph replay of forward+backward (bench.py, N > 1): in-place gradients, flat buckets."""
    world = 2
    port = _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker_flat, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))


def _worker_buckets(rank, world, port, ret):
    """GradBuckets (the in-backward, graph-capturable exchange bench.py times) on the REAL EDITOR parameter list: bucket
    plan in gradient-ready order, in-place gradient slots, launch when a bucket's last block reports, tail bucket with
    never-used parameters skipped, averaging."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import contextlib
    import io
    from editor_amd import config
    from editor_amd.modeling import make_model
    cfg, c, cams = config.preset("RGBNT100")              # AL = 0: AL_* absent, BACKBONE_HEAD / BN in use
    with contextlib.redirect_stdout(io.StringIO()):
        m = make_model(cfg, c, cams)
    gb = m.enable_grad_buckets(bucket_bytes=64 << 20)
    ok = gb.active
    segs, tail = m.grad_segments()
    names = {id(p): n for n, p in m.named_parameters()}
    ok &= [s[0] for s in segs][:5] == ["hma.joint", "hma.T", "hma.N", "hma.R", "backbone.11"] and segs[-1][0] == "backbone.0"
    ok &= 6 <= len(gb.buckets) <= 9                                   # 16 blocks x 28.3 MB in <= 64 MiB buckets
    total = sum(b["flat"].numel() for b in gb.buckets) + sum(p.numel() for p in tail if p.requires_grad)
    ok &= total == sum(p.numel() for p in m.parameters() if p.requires_grad)
    launched = []
    orig = gb._launch
    gb._launch = lambda b: (launched.append(id(b)), orig(b))[1]
    for step in range(2):
        # "backward": blocks report in ready order after writing rank-dependent gradients into their slots
        for si, (name, params) in enumerate(segs):
            sink = gb.sink(si)
            for j, v in enumerate(sink.views):
                if v is not None:
                    v.fill_(float((rank + 1) * (si + 1) + j + step))
            before = len(launched)
            sink.done()
            last_of_bucket = si == gb.buckets[gb.seg_bucket[si]]["segs"][-1]
            ok &= (len(launched) == before + 1) == last_of_bucket     # collective issued exactly when the bucket is complete
        for p in tail:                                                 # small parameters: ordinary .grad tensors
            n = names[id(p)]
            used = p.requires_grad and not n.startswith("BACKBONE.base.fc")
            p.grad = torch.full_like(p, float(rank + 1 + step)) if used else None
        gb.finish()
        for si, (name, params) in enumerate(segs):
            for j, p in enumerate(params):
                if p is not None:
                    want = sum((r + 1) * (si + 1) + j + step for r in range(world)) / world
                    ok &= bool((p.grad == want).all()) and p.grad.data_ptr() == gb.sink(si).views[j].data_ptr()
        for p in tail:
            if p.grad is not None:
                ok &= bool(torch.allclose(p.grad, torch.full_like(p, sum(r + 1 + step for r in range(world)) / world)))
        ln = gb.sink(0).ln_pair(0)                                     # [dgamma; dbeta] of a LayerNorm are adjacent slots
        ok &= ln.shape == (2, 768) and ln.data_ptr() == segs[0][1][0].grad.data_ptr()
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_grad_buckets_real_parameter_list_world2():
    world = 2
    port = _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker_buckets, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))
