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
