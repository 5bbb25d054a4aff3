"""End-to-end parity AT THE BENCHMARKED SIZE (B = 128 per GPU, BASELINE.json configs 2 and 3): the HIP model through
libeditor_hip.so against the oracle run on this host's cores.  At B = 128 the token-row count is M = 3*128*129 = 49 536,
so the GEMMs take the tile / split-K paths the bench times (the goldens of test_gpu_model.py are B <= 16).

f32 parity mode: frequency mask, per-modality attention masks and `index` bit-exact, features <= 1e-3 relative.
16-bit modes: selection agreement reported (asserted as a rate), features / gradients with the oracle's selection
teacher-forced; bounds = measured on MI355X x 1.5 (profiles/r03_parity_table.txt, first measured in round 2)."""
import pytest
import torch

from conftest import rel_err
from editor_amd import config, synth

pytestmark = pytest.mark.gpu

B = 128
TOL = {
    "f32": dict(cls4t=1e-3, loss=1e-4, grad=2e-3),
    # measured (profiles/r03_parity_table.txt; unchanged since round 2): f16 agree .9996 cls4t 8.4e-4 loss 2.4e-6..6.8e-6 grad 3.6e-3..5.9e-3 / 1.7e-2..2.3e-2
    # (patch embed); bf16 agree .9971 cls4t 6.5e-3 loss 1.2e-4 grad 1.85e-2 / 7.1e-2.  The gradient figures move by up to 1.6x
    # between builds whose attention outputs differ by ONE unit in the last place (bit-compared, tools/attn_bitcmp.py): the
    # reference's loss mines the hardest positive / negative per anchor (triplet_loss.py:84-85), a discrete choice that
    # near-ties flip, and one flipped pair shifts every upstream gradient together.  Bounds = worst observed x 1.5.
    "f16": dict(agree=0.999, cls4t=1.0e-3, loss=1.2e-5, grad=9e-3, grad_pe=3.5e-2),
    # split-precision forward (hi.hi + hi.lo + lo.hi on the half matrix cores): selection as the f32 mode (bit-exact except
    # verified fp32 ties), features <= 1e-4; its backward is the f16 mode's, so the gradient bounds are f16's
    "f16x2": dict(cls4t=1e-4, loss=1.2e-5, grad=9e-3, grad_pe=3.5e-2),
    # round 4, cfg.MODEL.SPLIT_SCOPE = 'selection': split precision only where the token selection depends on it (backbone up to
    # the last block's attention map); the last block's projection + MLP and the HMA head as the f16 mode -> selection as the f32
    # mode, features within the north star's 1e-3 (f16-class), loss / gradients f16's
    "f16x2s": dict(cls4t=1.0e-3, loss=1.2e-5, grad=9e-3, grad_pe=3.5e-2),
    "bf16": dict(agree=0.995, cls4t=1.0e-2, loss=3e-4, grad=2.8e-2, grad_pe=0.11),
}


class _Writer:
    def add_scalar(self, *a, **k):
        pass


MEASURED = {}                      # dtype -> cls4t rel err at B = 128 (filled by test_config2_eval_b128_vs_oracle)
NORTH_STAR_FEATURE_TOL = 1e-3      # BASELINE.json north_star: "within 1e-3 rel for bf16 features"


def _check_selection_f32(aux, oaux, nmod=3, k=2, tie=2e-5, max_rows=8):
    """f32 parity mode at full size: the selection must equal the oracle's bit for bit EXCEPT on (sample, head) rows
    whose k-th / (k+1)-th rollout scores are tied within fp32 rounding in the oracle itself - 4608 rows per batch with a
    measured near-tie density of ~40 rows per unit of relative gap make one such row per few
    batches unavoidable for ANY fp32 implementation that does not sum in the oracle's order.  Every disagreement is
    verified to be such a tie; their number is bounded.  Returns the number of tie rows."""
    assert torch.equal(aux["mask_fre"].cpu().bool(), oaux["mask_fre"])                  # integer path: always exact
    b = oaux["index"].shape[0]
    mine = aux["scores"].view(nmod, b, -1, aux["scores"].shape[-1]).cpu()
    ties = 0
    for i in range(nmod):
        sc = oaux["scores"][i]                                                           # (B, heads, N) oracle scores
        assert rel_err(mine[i], sc) < 1e-4
        got, want = aux["attn_masks"][i].cpu().bool(), oaux["attn_masks"][i]
        for bb in (got != want).any(dim=1).nonzero().flatten().tolist():
            top = sc[bb].topk(k + 1, dim=-1)
            gap = ((top.values[:, k - 1] - top.values[:, k]) / top.values[:, k - 1]).abs()
            mtop = mine[i, bb].topk(k, dim=-1).indices.sort(-1).values
            differs = (mtop != top.indices[:, :k].sort(-1).values).any(-1)
            assert differs.any(), "mask differs but no head's top-k does"
            assert (gap[differs] < tie).all(), ("selection differs on a row that is NOT an fp32 tie", i, bb, gap[differs])
            ties += int(differs.sum())
    print("f32 selection: %d (sample, head) rows decided by an fp32 near-tie (of %d)" % (ties, nmod * b * sc.shape[1]))
    assert ties <= max_rows
    if ties == 0:
        assert torch.equal(aux["index"].cpu().bool(), oaux["index"])
    return ties


def _model(preset, seed, dtype, **over):
    from editor_amd.modeling import make_model
    cfg, c, cams = config.preset(preset, compute_dtype=dtype, **over)
    m = make_model(cfg, c, cams)
    synth.fill_state_dict_(m.state_dict(), seed)
    return m.cuda(), cfg, c, cams


@pytest.fixture(scope="module")
def oracle_eval_c2(oracle):
    """Oracle eval forward of BASELINE config 2 (RGBNT201, 256x128, B=128) on the host cores (~15-40 s)."""
    import os
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    cfg, c, cams = config.preset("RGBNT201", drop_path=0.0)
    from editor_amd.modeling import make_model
    m = make_model(cfg, c, cams)
    synth.fill_state_dict_(m.state_dict(), 61)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    img, label, cam, view = synth.make_batch(62, B, 256, 128, cams, instances=16)
    with torch.no_grad():
        ref, aux = oracle.editor_forward(sd, img, cam, training=False, al=cfg.MODEL.AL, return_aux=True)
    return dict(ref=ref, aux=aux, batch=(img, label, cam, view))


@pytest.mark.parametrize("dtype", ["f32", "f16x2", "f16x2s", "f16", "bf16"])
def test_config2_eval_b128_vs_oracle(dtype, oracle_eval_c2):
    o = oracle_eval_c2
    img, label, cam, view = o["batch"]
    m, cfg, c, cams = _model("RGBNT201", 61, dtype, drop_path=0.0)
    m.eval()
    gimg = {k: v.cuda() for k, v in img.items()}
    with torch.no_grad():
        out = m(gimg, cam_label=cam.cuda(), view_label=view.cuda())
    aux = m.last_aux
    assert torch.equal(aux["mask_fre"].cpu().bool(), o["aux"]["mask_fre"])              # integer path: exact in any mode
    masks = [aux["attn_masks"][i].cpu().bool() for i in range(3)]
    if dtype in ("f32", "f16x2", "f16x2s"):
        ties = _check_selection_f32(aux, o["aux"])
        if dtype == "f16x2s":
            # the mode `value_at_parity` is quoted on - on THIS batch (seeds 61 / 62, the bench's accuracy protocol at
            # the benchmarked size) not one of the 4 608 (sample, head) rows may differ: no verified-tie allowance, every mask equal
            assert ties == 0
            for i in range(3):
                assert torch.equal(masks[i], o["aux"]["attn_masks"][i])
            assert torch.equal(aux["index"].cpu().bool(), o["aux"]["index"])
        if ties:
            m.teacher_index = o["aux"]["index"]
            with torch.no_grad():
                out = m(gimg, cam_label=cam.cuda(), view_label=view.cuda())
        err = rel_err(out.cpu(), o["ref"])
        print(dtype, "B=128 cls4t rel err:", err)
        MEASURED[dtype] = err
        assert err < TOL[dtype]["cls4t"]
        return
    agree = [(masks[i] == o["aux"]["attn_masks"][i]).float().mean().item() for i in range(3)]
    rows = [(masks[i] == o["aux"]["attn_masks"][i]).all(dim=1).float().mean().item() for i in range(3)]
    print(dtype, "B=128 attention-mask agreement (elements):", agree, "(whole rows):", rows)
    assert min(agree) > TOL[dtype]["agree"]
    m.teacher_index = o["aux"]["index"]
    with torch.no_grad():
        out = m(gimg, cam_label=cam.cuda(), view_label=view.cuda())
    err = rel_err(out.cpu(), o["ref"])
    print(dtype, "B=128 cls4t rel err (teacher-forced):", err)
    MEASURED[dtype] = err
    assert err < TOL[dtype]["cls4t"]


def test_feature_error_of_every_mode_against_the_north_star_bar():
    """The north star's feature tolerance (1e-3 relative) against what every compute mode measures at B = 128 - stated, not
    loosened: f32, f16x2 and f16 are inside it; bf16 (8-bit mantissas; the dtype BASELINE names and bench.py's default) is
    NOT (6.5e-3: torch's own CPU bf16 autocast differs from fp32 by 7e-3), which is why the bench line
    carries the accuracy of every mode next to its speed (`modes`).  Runs after test_config2_eval_b128_vs_oracle."""
    if len(MEASURED) < 4:
        pytest.skip("needs the four parametrisations of test_config2_eval_b128_vs_oracle in the same session")
    print("cls4t rel err at B = 128 vs the north-star bar %.0e:" % NORTH_STAR_FEATURE_TOL,
          {k: float("%.3g" % v) for k, v in MEASURED.items()})
    assert MEASURED["f32"] < 1e-4 and MEASURED["f16x2"] < 1e-4            # fp32-class
    if "f16x2s" in MEASURED:                                              # selection-scope split: exact selection, f16-class features
        assert MEASURED["f16x2"] < MEASURED["f16x2s"] < NORTH_STAR_FEATURE_TOL
    assert MEASURED["f16"] < NORTH_STAR_FEATURE_TOL                       # the reference's own autocast dtype meets the bar
    assert NORTH_STAR_FEATURE_TOL < MEASURED["bf16"] < 1e-2               # bf16 does NOT: the documented gap, visible here


GRAD_KEYS = ["BACKBONE.base.blocks.0.attn.qkv.weight", "BACKBONE.base.blocks.0.attn.qkv.bias",
             "BACKBONE.base.blocks.5.attn.proj.weight", "BACKBONE.base.blocks.11.mlp.fc1.weight",
             "BACKBONE.base.blocks.7.mlp.fc2.bias", "BACKBONE.base.patch_embed.proj.weight", "BACKBONE.base.cls_token",
             "BACKBONE.base.pos_embed", "BACKBONE.base.norm.weight", "FUSE_block.attn1.qkv.weight",
             "FUSE_block.mlpN.fc2.weight", "FUSE_block.normT.weight", "RGB_REDUCE.weight", "FUSE_HEAD.weight",
             "BACKBONE_HEAD.weight", "FUSE_BN.weight"]
# (not FUSE_block.out_norm.bias / *_REDUCE.bias: under the real loss their gradient is identically zero - a constant
# shift of cls4t is removed by FUSE_BN's batch statistics and leaves the triplet distances unchanged - so both sides
# hold rounding noise; the projection-loss goldens of test_gpu_model.py cover them)


DROP_SEED = 20260929        # stochastic-depth RNG counter the drop-path variant starts from (device state of EDITOR._drop_state)


def _oracle_train_c3(oracle, drop_path):
    """Oracle training step (forward, the real loss head, backward) of BASELINE config 3 (RGBNT100, 128x256, AL=0,
    B=128) on the host cores (~40-90 s).  drop_path > 0: the keep masks are the ones the PRODUCT's generator draws from
    DROP_SEED (editor_droppath_scales: per (block, branch, stacked sample) - two independent draws per block as the reference's
    two self.drop_path calls, vit_pytorch.py:217-218), handed to the oracle as (modality, block, branch, sample)."""
    import os
    from editor_amd import ops
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    cfg, c, cams = config.preset("RGBNT100", drop_path=drop_path)
    from editor_amd.modeling import make_model
    m = make_model(cfg, c, cams)
    synth.fill_state_dict_(m.state_dict(), 63)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    for k in GRAD_KEYS:
        sd[k].requires_grad_(True)
    img, label, cam, view = synth.make_batch(64, B, 128, 256, cams, instances=16)
    drop, scales = {}, None
    if drop_path > 0:
        rates = m.BACKBONE.base.drop_rates
        t_ = m.BACKBONE.base.num_patches + 1
        scales = ops.droppath_scales(torch.tensor(rates, dtype=torch.float32, device="cuda"), 3 * B, t_, DROP_SEED)
        per = scales.view(len(rates), 2, 3 * B, t_)
        assert bool((per == per[..., :1]).all())                                       # one draw per sample, repeated over its rows
        keep = (per[..., 0] > 0).view(len(rates), 2, 3, B).permute(2, 0, 1, 3).float().cpu()   # (3, depth, 2, B)
        kp = 1.0 - torch.tensor(rates, dtype=torch.float32)
        want = keep.permute(1, 2, 0, 3).reshape(len(rates), 2, 3 * B) / kp.view(-1, 1, 1)
        assert torch.equal(per[..., 0].cpu(), want)                                    # scale = 0 or 1 / keep_prob, exactly
        frac = keep[:, 1:].mean().item()
        assert 0.90 < frac < 0.99 and bool((keep[:, 0] == 1).all()), frac               # rates 0 .. 0.1, block 0 never drops
        assert not torch.equal(keep[:, :, 0], keep[:, :, 1])                           # the two branches draw independently
        drop = dict(drop_keep=keep, drop_rates=rates)
    out, aux = oracle.editor_forward(sd, img, cam, label=label, training=True, al=0, return_aux=True, **drop)
    loss = oracle.loss_pairs(out, label)
    loss.backward()
    return dict(loss=loss.detach(), out=[o.detach() for o in out], aux=aux, grads={k: sd[k].grad.clone() for k in GRAD_KEYS},
                batch=(img, label, cam, view), scales=scales, drop_path=drop_path)


@pytest.fixture(scope="module")
def oracle_train_c3(oracle):
    return _oracle_train_c3(oracle, 0.0)


@pytest.fixture(scope="module")
def oracle_train_c3_dp(oracle):
    return _oracle_train_c3(oracle, 0.1)


def _mined_pairs(feat, label):
    """batch-hard mining of TripletLoss (layers/triplet_loss.py:84-85): per anchor the index of its farthest positive and of its
    nearest negative, from fp32 distances of `feat`."""
    f = feat.detach().float().cpu()
    dist = torch.cdist(f, f)
    same = label.view(-1, 1).eq(label.view(1, -1))
    return dist.masked_fill(~same, -1.0).argmax(1), dist.masked_fill(same, float("inf")).argmin(1)


def _train_step_b128_vs_oracle(dtype, o):
    from editor_amd import losses
    img, label, cam, view = o["batch"]
    dp = o["drop_path"]
    m, cfg, c, cams = _model("RGBNT100", 63, dtype, drop_path=dp)
    gimg = {k: v.cuda() for k, v in img.items()}

    def seed_drop():
        if dp > 0:           # the product's own generator, started where the oracle's masks were drawn from
            m._drop_state = torch.full((1,), DROP_SEED, dtype=torch.int64, device="cuda")

    if dtype in ("f32", "f16x2", "f16x2s"):            # the selection itself
        if dp > 0:           # stochastic depth changes the attention maps: check the TRAINING forward's selection, then restore the state it updated
            sd0 = {k: v.clone() for k, v in m.state_dict().items()}
            m.train()
            seed_drop()
            with torch.no_grad():
                m(gimg, label=label.cuda(), cam_label=cam.cuda(), view_label=view.cuda(), writer=_Writer(), epoch=1)
            m.load_state_dict(sd0)
        else:                # checked in eval mode (no state is updated)
            m.eval()
            with torch.no_grad():
                m(gimg, cam_label=cam.cuda(), view_label=view.cuda())
        _check_selection_f32(m.last_aux, o["aux"])
    m.train()
    m.teacher_index = o["aux"]["index"]
    seed_drop()
    out = m(gimg, label=label.cuda(), cam_label=cam.cuda(), view_label=view.cuda(), writer=_Writer(), epoch=1)
    assert len(out) == 9
    if dp > 0:
        assert torch.equal(m.last_drop_scales, o["scales"])         # the step multiplied by exactly the masks the oracle was given
    loss = losses.loss_pairs(out, label.cuda())
    loss.backward()
    lerr = abs(loss.item() / o["loss"].item() - 1)
    oerr = max(rel_err(a.detach().float().cpu(), b) for a, b in zip(out, o["out"]))
    named = dict(m.named_parameters())
    gerr = {k: rel_err(named[k].grad.cpu(), o["grads"][k]) for k in GRAD_KEYS}
    worst = max(gerr, key=gerr.get)
    print(dtype, "B=128 train step (drop_path %.1f): loss rel err %.2e, worst output %.2e, worst gradient %.2e (%s)" %
          (dp, lerr, oerr, gerr[worst], worst))
    print(dtype, "   per-parameter gradient rel err:", {k.replace("BACKBONE.base.", ""): float("%.2e" % v) for k, v in gerr.items()})
    # The loss mines ONE hardest positive / negative per anchor (triplet_loss.py:84-85): a discrete choice.  Where the product's
    # 16-bit features order two near-tied candidates differently from the oracle's, that anchor's gradient flows through another
    # sample - the loss moves by the (tiny) margin difference, but every gradient upstream of that pair shifts together.  Count them.
    flips = 0
    for i in range(1, 9, 2):
        (pa, na), (pb, nb) = _mined_pairs(out[i], label), _mined_pairs(o["out"][i], label)
        flips += int((pa != pb).sum() + (na != nb).sum())
    print(dtype, "   hard-mining choices that differ from the oracle's (of %d): %d" % (8 * B, flips))
    assert lerr < TOL[dtype]["loss"]
    assert oerr < 10 * TOL[dtype]["cls4t"]       # all 9 outputs (scores, per-modality cls features, aux loss); cls4t itself is held to TOL in the eval tests
    # The patch-embedding weight gradient is ILL-CONDITIONED on this synthetic data: dW = sum_rows dx_row * pixels_row
    # with i.i.d. uniform pixels is mostly cancellation (measured on the oracle: rounding the exact fp32 dx to f16 moves
    # dW by 2e-4, but the 2e-3 error the 16-bit backward accumulates in dx over 12 layers - the same 2e-3 that cls_token /
    # pos_embed show - is amplified ~11x).  It gets its own bound; every other parameter is held to TOL["grad"].
    pe = "BACKBONE.base.patch_embed.proj.weight"
    rest = {k: v for k, v in gerr.items() if k != pe}
    worst = max(rest, key=rest.get)
    # with mined pairs that differ (above) the bound is the flipped pairs' share of the gradient, not rounding: x 2.5 (measured with
    # drop-path 0.1: f16 1.7e-2 on the HMA head's weights with every backbone gradient at 2e-3 .. 5e-3; bf16 4.2e-2) - the parameters the
    # triplet term does NOT reach (classifier heads: CE only) must stay inside the plain bound either way
    slack = 2.5 if flips else 1.0
    assert rest[worst] < slack * TOL[dtype]["grad"], (worst, rest[worst], flips)
    for k in ("FUSE_HEAD.weight", "BACKBONE_HEAD.weight"):
        assert gerr[k] < TOL[dtype]["grad"], (k, gerr[k])
    assert gerr[pe] < slack * TOL[dtype].get("grad_pe", TOL[dtype]["grad"]), gerr[pe]


@pytest.mark.parametrize("dtype", ["f32", "f16x2", "f16x2s", "f16", "bf16"])
def test_config3_train_step_b128_vs_oracle(dtype, oracle_train_c3):
    _train_step_b128_vs_oracle(dtype, oracle_train_c3)


@pytest.mark.parametrize("dtype", ["f32", "f16x2", "f16x2s", "f16", "bf16"])
def test_config3_train_step_b128_drop_path_vs_oracle(dtype, oracle_train_c3_dp):
    """the benchmarked workload runs DROP_PATH = 0.1 (bench.py); this is the same B = 128 training step with
    stochastic depth ON against the oracle (whose drop-path restatement is pinned to the reference by the f4_train_*_dp01
    goldens): the product draws the masks with its own generator, the oracle is handed those masks - a mis-scaled forward
    branch or a gradient leaking through a dropped branch fails loss / outputs / the 16 GRAD_KEYS at the existing TOL table."""
    _train_step_b128_vs_oracle(dtype, oracle_train_c3_dp)


@pytest.fixture(scope="module")
def oracle_eval_c4(oracle):
    """Oracle eval forward of BASELINE config 4 (MSVR310 preset, 384x128 input: N = 192 patches, T = 193, B = 128) on the host
    cores (~30-80 s).  M = 3 * 128 * 193 = 74 112 token rows: the 768-wide products run 290 x 3 tiles, the attention kernels their
    13-key-tile instantiation, the selection its n = 192 rows - none of which the B <= 16 goldens of this geometry reach."""
    import os
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    cfg, c, cams = config.preset("MSVR310", drop_path=0.0)
    from editor_amd.modeling import make_model
    m = make_model(cfg, c, cams)
    synth.fill_state_dict_(m.state_dict(), 65)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    img, label, cam, view = synth.make_batch(66, B, 384, 128, cams, instances=16)
    with torch.no_grad():
        ref, aux = oracle.editor_forward(sd, img, cam, training=False, al=cfg.MODEL.AL, return_aux=True)
    return dict(ref=ref, aux=aux, batch=(img, label, cam, view))


@pytest.mark.parametrize("dtype", ["f32", "f16x2s", "f16", "bf16"])
def test_config4_eval_b128_vs_oracle(dtype, oracle_eval_c4):
    """config 4 at the benchmarked size against the oracle - the protocol of test_config2_eval_b128_vs_oracle."""
    o = oracle_eval_c4
    img, label, cam, view = o["batch"]
    m, cfg, c, cams = _model("MSVR310", 65, dtype, drop_path=0.0)
    m.eval()
    gimg = {k: v.cuda() for k, v in img.items()}
    with torch.no_grad():
        out = m(gimg, cam_label=cam.cuda(), view_label=view.cuda())
    aux = m.last_aux
    assert tuple(aux["mask_fre"].shape) == (B, 192)
    assert torch.equal(aux["mask_fre"].cpu().bool(), o["aux"]["mask_fre"])
    if dtype in ("f32", "f16x2s"):
        forced = _check_selection_f32(aux, o["aux"])
    else:
        masks = [aux["attn_masks"][i].cpu().bool() for i in range(3)]
        agree = [(masks[i] == o["aux"]["attn_masks"][i]).float().mean().item() for i in range(3)]
        print(dtype, "config 4 B=128 attention-mask agreement (elements):", agree)
        assert min(agree) > TOL[dtype]["agree"]
        forced = True
    if forced:
        m.teacher_index = o["aux"]["index"]
        with torch.no_grad():
            out = m(gimg, cam_label=cam.cuda(), view_label=view.cuda())
    err = rel_err(out.cpu(), o["ref"])
    print(dtype, "config 4 (384x128) B=128 cls4t rel err:", err)
    assert err < TOL[dtype]["cls4t"]
