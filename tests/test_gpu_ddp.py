"""The data-parallel step on ONE GPU with a real RCCL group (world 1, forced): GradBuckets' in-backward gradient sinks,
the bucket all-reduces launched from inside the backward, the per-step buffer broadcast and the whole thing captured into
a hipGraph - against the same step without any of it.  A 1-rank AVG all-reduce is the identity, so parameters must come
out BIT-identical: an aliasing / ordering bug in the sink path (a column-sum partial landing in the wrong slot, a
collective reading a slot before the side-stream weight gradient wrote it) shows up as a difference.
SURVEY.md 8(e); reference: DistributedDataParallel, engine/processor.py:47-50 / train_net.py:63-64."""
import contextlib
import io
import os
import socket

import pytest
import torch
import torch.distributed as dist

from editor_amd import config, synth

pytestmark = pytest.mark.gpu


class _Quiet:
    def add_scalar(self, *a, **k):
        pass


@pytest.fixture(scope="module")
def rccl_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def _build(dtype, buckets):
    from editor_amd.modeling import make_model
    from editor_amd import solver
    torch.manual_seed(77)
    cfg, c, cams = config.preset("RGBNT201", compute_dtype=dtype, drop_path=0.1)
    with contextlib.redirect_stdout(io.StringIO()):
        m = make_model(cfg, c, cams)
    synth.fill_state_dict_(m.state_dict(), 31)
    m = m.cuda().train()
    gb = m.enable_grad_buckets(force=True) if buckets else None
    if gb is not None:
        gb.broadcast_parameters(m)
    opt, _ = solver.make_optimizer(cfg, m, None)
    return m, opt, gb, cams


def _step_fn(m, opt, gb, batch):
    from editor_amd import losses
    img, label, cam, view = batch

    def step():
        opt.zero_grad(set_to_none=True)
        if gb is not None:
            gb.broadcast_buffers(m)
        out = m(img, label=label, cam_label=cam, view_label=view, img_path=None, writer=_Quiet(), epoch=1)
        loss = losses.loss_pairs(out, label)
        loss.backward()
        if gb is not None:
            gb.finish()
        opt.step()
        return loss
    return step


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_grad_bucket_sinks_equal_plain_autograd(dtype, rccl_group):
    """Every parameter gradient of one step: in-place bucket slots (+ the 1-rank collectives) == tensors returned through
    autograd, bit for bit; then three eager steps and a captured + replayed step leave identical parameters."""
    b = 64
    m0, opt0, _, cams = _build(dtype, False)
    m1, opt1, gb, _ = _build(dtype, True)
    assert gb.active and len(gb.buckets) >= 6
    img, label, cam, view = synth.make_batch(5, b, 256, 128, cams, instances=8)
    batch = ({k: v.cuda() for k, v in img.items()}, label.cuda(), cam.cuda(), view.cuda())
    s0, s1 = _step_fn(m0, opt0, None, batch), _step_fn(m1, opt1, gb, batch)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        # gradients of the first step, before the optimizers touch anything else
        from editor_amd import losses
        for m, g_ in ((m0, None), (m1, gb)):
            m.zero_grad(set_to_none=True) if g_ is None else opt1.zero_grad()
        out0 = m0(batch[0], label=batch[1], cam_label=batch[2], view_label=batch[3], writer=_Quiet(), epoch=1)
        losses.loss_pairs(out0, batch[1]).backward()
        m1._drop_state = None if m1._drop_state is None else m1._drop_state          # (same seed -> same drop-path draws)
        out1 = m1(batch[0], label=batch[1], cam_label=batch[2], view_label=batch[3], writer=_Quiet(), epoch=1)
        losses.loss_pairs(out1, batch[1]).backward()
        gb.finish()
        torch.cuda.synchronize()
        n0, n1 = dict(m0.named_parameters()), dict(m1.named_parameters())
        bad = [k for k in n0 if (n0[k].grad is None) != (n1[k].grad is None) or
               (n0[k].grad is not None and not torch.equal(n0[k].grad, n1[k].grad))]
        assert not bad, bad[:8]
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    # fresh models (the manual step above advanced drop-path state and BN statistics identically, but keep it simple)
    m0, opt0, _, _ = _build(dtype, False)
    m1, opt1, gb, _ = _build(dtype, True)
    s0, s1 = _step_fn(m0, opt0, None, batch), _step_fn(m1, opt1, gb, batch)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(5):
            s0()
        for _ in range(2):
            s1()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    opt1.zero_grad(set_to_none=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = s1()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(static_loss).item()
    sd0, sd1 = m0.state_dict(), m1.state_dict()
    diff = [k for k in sd0 if not torch.equal(sd0[k], sd1[k])]
    assert not diff, diff[:8]
