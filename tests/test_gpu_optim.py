"""Row N4 on the device: FusedSGD driven through the torch.optim surface the reference's harness uses (param_groups
written by the scheduler, state_dict round trip), pinned to the reference's make_optimizer / create_scheduler golden."""
import json
import os

import pytest
import torch

from conftest import GOLDEN, rel_err
from editor_amd import config, synth

pytestmark = pytest.mark.gpu


def _toy(seed=3):
    g = torch.Generator().manual_seed(seed)
    names = ["a.weight", "a.bias", "b.weight", "cls_token", "c.bias"]
    shapes = [(64, 48), (64,), (33, 64), (1, 1, 48), (33,)]
    return names, [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]


def test_make_optimizer_groups_and_scheduler_drive_fused_sgd():
    """make_optimizer -> FusedSGD with the reference's group table; create_scheduler writes lr into the DEVICE table the
    kernel reads; three epochs of steps equal torch.optim.SGD driven by the same per-epoch learning rates."""
    from editor_amd import solver
    from editor_amd.modeling import make_model
    gold = json.load(open(os.path.join(GOLDEN, "f10_solver.json")))
    cfg, c, cams = config.preset("RGBNT201")
    m = make_model(cfg, c, cams).cuda()
    opt, opt_center = solver.make_optimizer(cfg, m, torch.nn.Linear(2, 2))
    assert [[g["name"], g["lr"], g["weight_decay"]] for g in opt.param_groups] == gold["table"]
    assert isinstance(opt_center, torch.optim.SGD)
    sched = solver.create_scheduler(cfg, opt)
    names = [g["name"] for g in opt.param_groups]
    i_w, i_b = names.index("BACKBONE.base.blocks.0.attn.qkv.weight"), names.index("BACKBONE.base.blocks.0.attn.qkv.bias")
    for epoch in (0, 1, 5, 10, 11, 40, 69, 70, 75):
        sched.step(epoch)
        lr_dev = opt.lr.cpu()
        want = gold["lrs"][epoch]
        assert abs(lr_dev[i_w].item() - want[0]) <= 1e-7 * want[0] + 1e-12
        assert abs(lr_dev[i_b].item() - want[1]) <= 1e-7 * want[1] + 1e-12


def test_fused_sgd_follows_param_groups_and_checkpoints():
    from editor_amd.optim import FusedSGD
    names, ps = _toy()
    ref = [p.detach().clone().requires_grad_(True) for p in ps]
    fopt = FusedSGD(list(zip(names, ps)), base_lr=1e-2, weight_decay=1e-4, bias_lr_factor=2.0, weight_decay_bias=1e-4,
                    momentum=0.9)
    topt = torch.optim.SGD([{"params": [r], "lr": g["lr"], "weight_decay": g["weight_decay"]}
                            for r, g in zip(ref, fopt.param_groups)], momentum=0.9)
    g = torch.Generator().manual_seed(9)

    def one_step(fo, to, params, refs):
        for p, r in zip(params, refs):
            gr = torch.randn(p.shape, generator=g).cuda()
            p.grad, r.grad = gr.clone(), gr.clone()
        fo.step()
        to.step()

    for step in range(4):
        if step == 2:                                   # a scheduler writes param_groups, nothing else
            for fg, tg in zip(fopt.param_groups, topt.param_groups):
                fg["lr"] = tg["lr"] = fg["lr"] * 0.3
        one_step(fopt, topt, ps, ref)
        for p, r in zip(ps, ref):
            assert rel_err(p.detach().cpu(), r.detach().cpu()) < 1e-6
    fopt.set_lr(5e-3)
    assert torch.allclose(fopt.lr.cpu(), torch.full((len(ps),), 5e-3))
    assert not fopt.found_inf()
    # checkpoint round trip: momentum buffers + groups restore an optimizer that continues identically
    sd = fopt.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == len(ps)
    assert torch.equal(sd["state"][0]["momentum_buffer"], topt.state[ref[0]]["momentum_buffer"]) or \
        rel_err(sd["state"][0]["momentum_buffer"].cpu(), topt.state[ref[0]]["momentum_buffer"].cpu()) < 1e-6
    ps2 = [p.detach().clone().requires_grad_(True) for p in ps]
    fopt2 = FusedSGD(list(zip(names, ps2)), base_lr=1.0, momentum=0.9)
    fopt2.load_state_dict(sd)
    for p, q in zip(ps, ps2):
        gr = torch.randn(p.shape, generator=g).cuda()
        p.grad, q.grad = gr.clone(), gr.clone()
    fopt.step()
    fopt2.step()
    for p, q in zip(ps, ps2):
        assert torch.equal(p.detach(), q.detach())


def test_fused_sgd_first_step_under_capture_keeps_momentum():
    """the first step() may run under hipGraph capture (no 'first step' flag is baked in): replays keep
    accumulating momentum exactly like eager steps."""
    from editor_amd.optim import FusedSGD
    names, ps = _toy(5)
    ps2 = [p.detach().clone().requires_grad_(True) for p in ps]
    grads = [torch.randn(p.shape, generator=torch.Generator().manual_seed(11)).cuda() for p in ps]
    eager = FusedSGD(list(zip(names, ps2)), base_lr=1e-2, momentum=0.9)
    for _ in range(3):
        for q, gr in zip(ps2, grads):
            q.grad = gr
        eager.step()
    cap = FusedSGD(list(zip(names, ps)), base_lr=1e-2, momentum=0.9)
    for p, gr in zip(ps, grads):
        p.grad = gr
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            cap.step()                                   # FIRST step of this optimizer is the captured one
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for p, q in zip(ps, ps2):
        assert torch.equal(p.detach(), q.detach())
    # a learning-rate change between replays is honoured by the captured launch (device-resident table)
    cap.set_lr(0.0)
    before = [p.detach().clone() for p in ps]
    graph.replay()
    torch.cuda.synchronize()
    for p, b in zip(ps, before):
        assert torch.equal(p.detach(), b)


def test_fused_sgd_flags_nonfinite_gradients():
    from editor_amd.optim import FusedSGD
    names, ps = _toy(13)
    opt = FusedSGD(list(zip(names, ps)), base_lr=1e-2, momentum=0.9)
    for p in ps:
        p.grad = torch.randn_like(p)
    opt.step()
    assert not opt.found_inf()
    ps[2].grad[5, 7] = float("inf")
    opt.step()
    assert opt.found_inf() and not opt.found_inf()          # reported once, then reset


def test_fused_sgd_f16_shadows():
    from editor_amd import functional as fnc
    from editor_amd.optim import FusedSGD
    names, ps = _toy(7)
    names = names + ["big.weight", "wide.weight"]
    ps = ps + [torch.randn(768, 1024).cuda().requires_grad_(True), torch.randn(256, 3072).cuda().requires_grad_(True)]
    opt = FusedSGD(list(zip(names, ps)), base_lr=1e-2, momentum=0.9, shadow_dtype=torch.float16)
    for p in ps:
        p.grad = torch.randn_like(p)
    opt.step()
    for p, h in zip(ps, opt.shadows):
        if h is not None:
            assert h.dtype == torch.float16 and torch.equal(h, p.detach().half())
            assert fnc.act_weight(p, torch.float16).data_ptr() == h.data_ptr()
    # k-major (transposed) copies of the big 2-D weights, written by the multi-tensor transpose launch of the same step
    nt = 0
    for p, ht in zip(ps, opt.shadows_t):
        if ht is not None:
            nt += 1
            assert ht.shape == (p.shape[1], p.shape[0]) and torch.equal(ht, p.detach().half().t())
            assert fnc.act_weight_t(p, torch.float16).data_ptr() == ht.data_ptr()
    assert nt == 2
    # without an optimizer in the loop the cache builds the transpose itself
    q = torch.randn(128, 256).cuda().requires_grad_(True)
    assert torch.equal(fnc.act_weight_t(q, torch.bfloat16), q.detach().bfloat16().t())


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    names = ["a.weight", "a.bias", "b.weight"]
    ps = [torch.nn.Parameter((torch.randn(s, generator=g) * 0.1).cuda()) for s in ((64, 128), (64,), (300, 64))]
    return names, ps


def test_fused_sgd_skips_the_step_on_overflow():
    """GradScaler.step semantics (engine/processor.py:94-96): a step whose gradients hold an inf / nan writes
    NOTHING - parameters, momentum buffers and 16-bit shadows keep their bits - and is reported; the next clean step applies."""
    from editor_amd.optim import FusedSGD
    names, ps = _params()
    opt = FusedSGD(list(zip(names, ps)), base_lr=1e-2, momentum=0.9, shadow_dtype=torch.float16)
    assert opt.check_overflow
    g = torch.Generator().manual_seed(1)
    for p in ps:
        p.grad = torch.randn(p.shape, generator=g).cuda()
    opt.step()                                               # a clean step: momentum buffers and shadows now hold values
    before = [(p.detach().clone(), b.clone(), None if h is None else h.clone()) for p, b, h in zip(ps, opt.bufs, opt.shadows)]
    assert not opt.found_inf()
    for p in ps:
        p.grad = torch.randn(p.shape, generator=g).cuda()
    ps[2].grad[17, 3] = float("inf")
    opt.step()
    for p, b, h, (p0, b0, h0) in zip(ps, opt.bufs, opt.shadows, before):
        assert torch.equal(p.detach(), p0) and torch.equal(b, b0) and (h is None or torch.equal(h, h0))
    assert opt.found_inf() and not opt.found_inf()           # reported once, then reset
    ps[2].grad[17, 3] = 0.5
    opt.step()
    assert not torch.equal(ps[0].detach(), before[0][0]) and not opt.found_inf()
    # bf16 / fp32 modes do not pay for the check pass
    assert not FusedSGD(list(zip(names, ps)), base_lr=1e-2).check_overflow


def test_device_grad_scaler_matches_torch_sgd_and_backs_off():
    """DeviceGradScaler: scale(loss).backward() -> step (unscale + update on the device) -> update(); an injected overflow
    leaves the weights unchanged and halves the scale; growth after `growth_interval` clean steps; the whole sequence
    replays from a hipGraph."""
    from editor_amd.optim import DeviceGradScaler, FusedSGD
    names, ps = _params(3)
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = FusedSGD(list(zip(names, ps)), base_lr=1e-2, momentum=0.9, weight_decay=0.0, weight_decay_bias=0.0, bias_lr_factor=1.0,
                   shadow_dtype=torch.float16)
    topt = torch.optim.SGD(ref, lr=1e-2, momentum=0.9)
    sc = DeviceGradScaler("cuda", init_scale=2.0 ** 10, growth_interval=3)
    x = torch.randn(32, 128, generator=torch.Generator().manual_seed(9)).cuda()

    def loss_of(params, poison=0.0):
        y = torch.tanh(x @ params[0].t() + params[1]) @ params[2].t()
        return y.pow(2).mean() * (1.0 + poison)

    for step in range(3):
        opt.zero_grad(); topt.zero_grad()
        sc.scale(loss_of(ps)).backward()
        assert abs(float(ps[0].grad.abs().max()) / 2.0 ** 10) > 0          # gradients carry the scale
        sc.step(opt); sc.update()
        loss_of(ref).backward(); topt.step()
        for p, r in zip(ps, ref):
            assert torch.allclose(p.detach(), r.detach(), atol=1e-6, rtol=1e-5)
    assert sc.get_scale() == 2.0 ** 11                                     # three clean steps: grown once
    keep = [p.detach().clone() for p in ps]
    opt.zero_grad()
    sc.scale(loss_of(ps, poison=float("inf"))).backward()                 # overflow in the backward
    sc.step(opt); sc.update()
    assert all(torch.equal(p.detach(), k) for p, k in zip(ps, keep)) and sc.get_scale() == 2.0 ** 10
    # captured: the same three calls replay with the device-resident scale
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            opt.zero_grad()
            sc.scale(loss_of(ps)).backward()
            sc.step(opt); sc.update()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    opt.zero_grad()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sc.scale(loss_of(ps)).backward()
        sc.step(opt); sc.update()
    s0 = sc.get_scale()
    w0 = ps[0].detach().clone()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert not torch.equal(ps[0].detach(), w0) and sc.get_scale() >= s0


def test_fused_adamw_matches_torch_adamw_and_checkpoints():
    """OPTIMIZER_NAME 'AdamW' (solver/make_optimizer.py:23-24): editor_adamw_multi against torch.optim.AdamW (CPU, fp32) fed the same
    gradients through per-parameter groups with their own lr / weight decay - five steps, a state_dict round trip into a fresh
    optimizer (torch's layout: step / exp_avg / exp_avg_sq), three more; then a skipped (overflow) step leaves everything as it was."""
    from editor_amd.optim import FusedAdamW
    names, ps = _toy(5)
    ref = [p.detach().cpu().clone().requires_grad_(True) for p in ps]
    fopt = FusedAdamW(list(zip(names, ps)), base_lr=3e-3, weight_decay=5e-2, bias_lr_factor=2.0, weight_decay_bias=1e-3,
                      shadow_dtype=torch.bfloat16)
    topt = torch.optim.AdamW([{"params": [r], "lr": g["lr"], "weight_decay": g["weight_decay"]} for r, g in zip(ref, fopt.param_groups)],
                             lr=3e-3, weight_decay=5e-2)
    assert fopt.param_groups[1]["lr"] == 6e-3 and fopt.param_groups[1]["weight_decay"] == 1e-3
    g = torch.Generator().manual_seed(13)

    def steps(fo, params, n):
        for _ in range(n):
            for p, r in zip(params, ref):
                gr = torch.randn(p.shape, generator=g)
                p.grad, r.grad = gr.cuda(), gr.clone()
            fo.step()
            topt.step()
    steps(fopt, ps, 5)
    for p, r in zip(ps, ref):
        assert rel_err(p.detach().cpu(), r.detach()) < 2e-6
    sd = fopt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 5.0
    assert rel_err(sd["state"][2]["exp_avg_sq"].cpu(), topt.state_dict()["state"][2]["exp_avg_sq"]) < 2e-6
    ps2 = [p.detach().clone().requires_grad_(True) for p in ps]
    fopt2 = FusedAdamW(list(zip(names, ps2)), base_lr=1.0, weight_decay=0.0, shadow_dtype=torch.bfloat16)
    fopt2.load_state_dict(sd)
    assert fopt2.param_groups[1]["lr"] == 6e-3
    steps(fopt2, ps2, 3)
    for p, r in zip(ps2, ref):
        assert rel_err(p.detach().cpu(), r.detach()) < 3e-6
    assert torch.equal(fopt2.shadows[0], ps2[0].detach().to(torch.bfloat16))                 # 16-bit operand copy written by the update
    # overflow protocol (GradScaler.step): an inf gradient -> nothing moves, the step count stays
    fopt2.check_overflow = True
    before = [p.detach().clone() for p in ps2]
    t0 = float(fopt2.step_count.item())
    for p in ps2:
        p.grad = torch.full_like(p, float("inf"))
    fopt2.step()
    assert all(torch.equal(a, b) for a, b in zip(before, [p.detach() for p in ps2])) and float(fopt2.step_count.item()) == t0
    assert fopt2.found_inf()


def test_make_optimizer_builds_adamw():
    from editor_amd import solver
    from editor_amd.optim import FusedAdamW
    from editor_amd.modeling import make_model
    cfg, c, cams = config.preset("RGBNT201")
    cfg.SOLVER.OPTIMIZER_NAME = "AdamW"
    m = make_model(cfg, c, cams).cuda()
    opt, _ = solver.make_optimizer(cfg, m, None)
    assert isinstance(opt, FusedAdamW)
    gold = json.load(open(os.path.join(GOLDEN, "f10_solver.json")))
    assert [[g["name"], g["lr"], g["weight_decay"]] for g in opt.param_groups] == gold["table"]      # the same group rule (:6-19)
