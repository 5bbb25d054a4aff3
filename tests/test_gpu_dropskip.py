"""Stochastic-depth compaction (round 6): the MLP branch of a backbone block runs on the samples its drop-path draw kept
(vit_pytorch.py:52-69,217-218: a dropped sample's branch is multiplied by 0 and gets no gradient through it).  The kernels
that carry it - editor_droppath_plan, editor_layernorm_fwd_perm, editor_gemm_h16_rows, the *_perm_parts backward forms, the
per-problem live counts of the grouped weight gradients - against their dense counterparts, then the whole training step
with the skipping on against the step with it off: forward outputs and input gradients BIT-identical (a live row's
arithmetic does not depend on where the row sits), weight gradients equal up to the fp32 summation order of the reduction
over token rows (the dropped rows contributed exact zeros)."""
import pytest
import torch

from conftest import rel_err
from editor_amd import config, functional as fn, ops, synth

pytestmark = pytest.mark.gpu


def _scales(depth=12, b=24, t=129, seed=11, rate=0.3):
    rates = torch.linspace(0, rate, depth, device="cuda")
    return ops.droppath_scales(rates, b, t, seed), rates


def test_droppath_plan_orders_live_samples_first():
    depth, b, t = 12, 24, 129
    sc, rates = _scales(depth, b, t)
    perm, inv, live = ops.droppath_plan(sc, depth, b, t)
    keep = sc.view(depth, 2, b, t)[..., 0] != 0                                   # (depth, 2, B)
    assert torch.equal(live.long(), keep.sum(-1) * t)
    assert bool((keep[1:].float().mean() < 0.97)) and bool(keep[0].all())           # something was dropped; block 0 never drops
    ar = torch.arange(b * t, device="cuda")
    for l in range(depth):
        for br in range(2):
            p, q = perm[l, br].long(), inv[l, br].long()
            assert torch.equal(q[p], ar) and torch.equal(p[q], ar)                 # a permutation and its inverse
            k = keep[l, br]
            pos_live = torch.cumsum(k.long(), 0) - 1
            pos_dead = int(k.sum()) + torch.cumsum((~k).long(), 0) - 1
            slot = torch.where(k, pos_live, pos_dead)                              # live samples first, order kept
            want = (slot[:, None] * t + torch.arange(t, device="cuda")[None]).reshape(-1)
            assert torch.equal(p, want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_layernorm_fwd_perm_equals_dense_rows(dtype):
    b, t, d = 24, 129, 768
    sc, _ = _scales(12, b, t)
    perm, inv, live = ops.droppath_plan(sc, 12, b, t)
    rs, p = sc[7, 1].contiguous(), perm[7, 1].contiguous()
    x = synth.normal(3, "x", (b * t, d), 1.0).cuda()
    g_, b_ = synth.normal(3, "g", (d,), 1.0).cuda(), synth.normal(3, "b", (d,), 1.0).cuda()
    y0, m0, r0 = ops.layernorm_fwd(x, g_, b_, 1e-6, dtype)
    copy = torch.full_like(x, float("nan"))
    y1, m1, r1 = ops.layernorm_fwd_perm(x, g_, b_, 1e-6, dtype, p, rs, copy)
    assert torch.equal(m0, m1) and torch.equal(r0, r1)
    dead = rs == 0
    nl = int(live[7, 1])
    assert torch.equal(y1[p.long()][~dead], y0[~dead])                             # live rows: the same bits, at their slots
    assert bool((y1[nl:] == 0).all()) and int(dead.sum()) == b * t - nl            # dropped rows: zeros behind the live prefix
    assert torch.equal(copy[dead], x[dead]) and bool(torch.isnan(copy[~dead]).all())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fc2_on_compacted_rows_scatters_the_same_bits(dtype):
    """x2 = x1 + rs * (g W^T + b): dense over all rows vs the live prefix of the compacted rows + row map."""
    b, t, d, hidden = 24, 129, 768, 3072
    m = b * t
    sc, _ = _scales(12, b, t)
    perm, inv, live = ops.droppath_plan(sc, 12, b, t)
    rs, p, q, lv = sc[9, 1].contiguous(), perm[9, 1].contiguous(), inv[9, 1].contiguous(), live[9, 1:2]
    x1 = synth.normal(5, "x1", (m, d), 1.0).cuda()
    g = synth.normal(5, "g", (m, hidden), 1.0).cuda().to(dtype)
    w = synth.normal(5, "w", (d, hidden), 0.02).cuda().to(dtype)
    bias = synth.normal(5, "bias", (d,), 0.1).cuda()
    dense = torch.empty_like(x1)
    # (the dense product on the same 256x256 ping-pong kernel the path's M = 49 536 takes: at these few rows the shape heuristic
    #  would pick the 256x128 three-stage kernel, whose K-loop rounds differently in the last place)
    ops.gemm(g, w, dense, m, d, hidden, hidden, hidden, d, 0, 0, bias=bias, rowscale=rs, epilogue=ops.EPI_RESIDUAL | ops.EPI_FORCE_PP,
             aux=x1)
    gc = torch.zeros_like(g)
    gc[p.long()] = g                                                               # compacted operand (dropped rows behind the prefix)
    gc[int(lv):] = float("nan")              # behind the live prefix: whatever the allocator left (fc1 skips those tiles) - must not be read into a result
    out = torch.full_like(x1, float("nan"))
    dead = rs == 0
    out[dead] = x1[dead]                                                           # what LayerNorm-2 leaves for the dropped rows
    ops.gemm(gc, w, out, m, d, hidden, hidden, hidden, d, 0, 0, bias=bias, rowscale=rs, epilogue=ops.EPI_RESIDUAL, aux=x1,
             m_live=lv, live_dense=True, rowmap=q)
    assert torch.equal(out[~dead], dense[~dead])
    assert torch.equal(out[dead], x1[dead]) and torch.equal(dense[dead], x1[dead])


def _step(skip, dtype, b=64, seed=17, preset="RGBNT100"):
    from editor_amd.modeling import make_model
    from editor_amd import losses
    cfg, c, cams = config.preset(preset, compute_dtype=dtype, drop_path=0.3)   # a high rate: ~15 % of the MLP units dropped
    cfg.MODEL.DROP_SKIP = skip
    m = make_model(cfg, c, cams)
    synth.fill_state_dict_(m.state_dict(), seed)
    m = m.cuda().train()
    buckets = m.enable_grad_buckets()
    h, w = cfg.INPUT.SIZE_TRAIN
    img, label, cam, view = synth.make_batch(seed + 1, b, h, w, cams, instances=16, keys=config.MODALITY_KEYS[:m.nmod])
    gimg = {k: v.cuda().requires_grad_(k == "RGB") for k, v in img.items()}
    m._drop_state = torch.full((1,), 4242, dtype=torch.int64, device="cuda")

    class W:
        def add_scalar(self, *a, **k):
            pass
    out = m(gimg, label=label.cuda(), cam_label=cam.cuda(), view_label=view.cuda(), writer=W(), epoch=1)
    loss = losses.loss_pairs(out, label.cuda())
    loss.backward()
    buckets.finish()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    return [o.detach().clone() for o in out], loss.detach().clone(), grads, m.last_drop_scales.clone(), m.last_aux["index"].clone()


@pytest.mark.parametrize("dtype,preset,b", [("bf16", "RGBNT100", 64), ("f16", "RGBNT100", 64), ("f16x2", "RGBNT100", 64),
                                            ("f16x2s", "RGBNT100", 64),
                                            ("bf16", "MSVR310", 64),          # T = 193: 37 056 token rows, 13-key-tile attention
                                            ("bf16", "SYNTH4L", 16)])         # 4-modal ViT-L: D = 1024, hidden 4096, T = 513, 24 blocks
def test_training_step_with_skipping_equals_the_step_without(dtype, preset, b):
    assert fn.DROP_SKIP
    out0, loss0, g0, sc0, idx0 = _step(False, dtype, b, preset=preset)
    out1, loss1, g1, sc1, idx1 = _step(True, dtype, b, preset=preset)
    torch.cuda.empty_cache()
    assert torch.equal(sc0, sc1) and bool((sc1[1:, 1] == 0).float().mean() > 0.05)
    assert torch.equal(idx0, idx1)
    for a, b_ in zip(out0, out1):
        assert torch.equal(a, b_)                       # forward: bit-identical (live rows keep their arithmetic, dropped rows add exact 0)
    assert torch.equal(loss0, loss1)
    assert set(g0) == set(g1)
    worst = 0.0
    for k in g0:
        if g0[k].dim() == 2 and (".mlp.fc" in k or ".attn." in k) and "BACKBONE" in k:
            e = rel_err(g1[k], g0[k])                   # weight gradients: the reduction over token rows regrouped -> fp32 summation order
            worst = max(worst, e)
            assert e < 2e-5, (k, e)
        elif "BACKBONE" in k and ".norm" not in k and "bias" not in k:
            assert rel_err(g1[k], g0[k]) < 2e-5, k
        else:
            assert rel_err(g1[k], g0[k]) < 1e-4, k      # (bias / LayerNorm gradients: partial-row folds regrouped)
    print(dtype, "skip vs dense: worst weight-gradient rel diff %.2e" % worst)
