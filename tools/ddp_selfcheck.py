"""1-rank RCCL self-check of the data-parallel step (run by tests/test_gpu_ddp.py in a CHILD process, so that the process
group - its watchdog thread, its destructor - never shares a process with the other GPU tests' hipGraph captures):
    python tools/ddp_selfcheck.py bf16|f16
GradBuckets' in-backward gradient sinks + bucket all-reduces + per-step buffer broadcast, eager and captured, against the
same step without any of it.  A 1-rank AVG all-reduce is the identity, so everything must be BIT-identical."""
import contextlib
import io
import os
import socket
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from editor_amd import config, losses, synth  # noqa: E402


class _Quiet:
    def add_scalar(self, *a, **k):
        pass


WIRE = None          # torch.bfloat16: the 16-bit gradient exchange (python tools/ddp_selfcheck.py bf16 wire16)


def _build(dtype, buckets):
    from editor_amd.modeling import make_model
    from editor_amd import solver
    torch.manual_seed(77)
    cfg, c, cams = config.preset("RGBNT201", compute_dtype=dtype, drop_path=0.1)
    with contextlib.redirect_stdout(io.StringIO()):
        m = make_model(cfg, c, cams)
    synth.fill_state_dict_(m.state_dict(), 31)
    m = m.cuda().train()
    gb = m.enable_grad_buckets(force=True, wire_dtype=WIRE) if buckets else None
    if gb is not None:
        gb.broadcast_parameters(m)
    opt, _ = solver.make_optimizer(cfg, m, None)
    return m, opt, gb, cams


def _step_fn(m, opt, gb, batch):
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


def main(dtype):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    b = 64
    m0, opt0, _, cams = _build(dtype, False)
    m1, opt1, gb, _ = _build(dtype, True)
    assert gb.active and len(gb.buckets) >= 6
    img, label, cam, view = synth.make_batch(5, b, 256, 128, cams, instances=8)
    batch = ({k: v.cuda() for k, v in img.items()}, label.cuda(), cam.cuda(), view.cuda())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        # (1) every parameter gradient of one step: bucket slots (+ the collectives) == tensors returned through autograd
        m0.zero_grad(set_to_none=True)
        opt1.zero_grad()
        out0 = m0(batch[0], label=batch[1], cam_label=batch[2], view_label=batch[3], writer=_Quiet(), epoch=1)
        losses.loss_pairs(out0, batch[1]).backward()
        out1 = m1(batch[0], label=batch[1], cam_label=batch[2], view_label=batch[3], writer=_Quiet(), epoch=1)
        losses.loss_pairs(out1, batch[1]).backward()
        gb.finish()
        torch.cuda.synchronize()
        n0, n1 = dict(m0.named_parameters()), dict(m1.named_parameters())
        # (16-bit wire: a 1-rank AVG all-reduce of the bf16 twin is the identity on it, so every exchanged gradient is EXACTLY the
        #  plain gradient rounded to bf16 once - bucket slots and tail alike)
        want = (lambda g: g.to(WIRE).float()) if WIRE is not None else (lambda g: g)
        bad = [k for k in n0 if (n0[k].grad is None) != (n1[k].grad is None) or
               (n0[k].grad is not None and not torch.equal(want(n0[k].grad), n1[k].grad))]
        assert not bad, ("gradients differ", bad[:8])
        if WIRE is not None:
            assert all(p.grad.dtype == torch.float32 for p in m1.parameters() if p.grad is not None)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    # (2) five eager steps without the exchange == two eager + one captured step replayed three times with it
    m0, opt0, gb0, _ = _build(dtype, WIRE is not None)     # (16-bit wire: the eager reference exchanges in 16 bits too)
    m1, opt1, gb, _ = _build(dtype, True)
    s0, s1 = _step_fn(m0, opt0, gb0, batch), _step_fn(m1, opt1, gb, batch)
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
    from editor_amd.ddp import graph_capture_kwargs
    with torch.cuda.graph(graph, **graph_capture_kwargs()):
        static_loss = s1()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(static_loss).item()
    sd0, sd1 = m0.state_dict(), m1.state_dict()
    diff = [k for k in sd0 if not torch.equal(sd0[k], sd1[k])]
    assert not diff, ("parameters differ after capture + replay", diff[:8])
    print("DDP-SELFCHECK-OK", dtype, "buckets", len(gb.buckets), "wire", gb.describe()["wire_dtype"], flush=True)
    sys.stdout.flush()
    os._exit(0)          # (RCCL's destructors at interpreter exit are not part of the check)


if __name__ == "__main__":
    if "wire16" in sys.argv[2:]:
        WIRE = torch.bfloat16
    main(sys.argv[1] if len(sys.argv) > 1 else "bf16")
