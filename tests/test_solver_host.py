"""Row N4 / A9 host logic against goldens captured from the reference (CPU): optimizer group table, warm-up-cosine
schedule, Trans.load_param + resize_pos_embed.  The device side of the optimizer is in tests/test_gpu_optim.py."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from editor_amd import config, synth


def _gold():
    return json.load(open(os.path.join(GOLDEN, "f10_solver.json")))


def test_param_group_table_matches_reference_make_optimizer():
    """solver/make_optimizer.py:4-29 run on the reference model: same (name, lr, weight_decay) per trainable parameter."""
    from editor_amd import solver
    from editor_amd.modeling import make_model
    g = _gold()
    cfg, c, cams = config.preset("RGBNT201")
    m = make_model(cfg, c, cams)
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    table = solver.param_group_table(cfg, names)
    assert [list(t) for t in table] == g["table"]
    assert g["momentum"] == cfg.SOLVER.MOMENTUM
    for k, v in g["solver"].items():                      # the cfg defaults are the reference's (config/defaults.py)
        assert getattr(cfg.SOLVER, k) == v, k


class _Opt:                       # any object with param_groups drives the schedule (the reference's torch SGD does)
    def __init__(self, lrs):
        self.param_groups = [{"lr": v} for v in lrs]


def test_warmup_cosine_matches_reference_scheduler():
    """solver/scheduler_factory.py:7-31 + cosine_lr.py:67-94: lr of a weight and a bias group for epochs 0..80."""
    from editor_amd import solver
    g = _gold()
    cfg, _, _ = config.preset("RGBNT201")
    opt = _Opt([cfg.SOLVER.BASE_LR, cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR])
    sched = solver.create_scheduler(cfg, opt)
    assert [gp["lr"] for gp in opt.param_groups] == g["after_init"]     # constructor writes the warm-up start value
    for epoch, want in enumerate(g["lrs"]):
        sched.step(epoch)
        got = [gp["lr"] for gp in opt.param_groups]
        assert got == want, (epoch, got, want)                          # same float arithmetic: bit-equal doubles
    assert sched.get_epoch_values(80) == [1e-6, 1e-6]
    sd = sched.state_dict()
    s2 = solver.create_scheduler(cfg, _Opt([1.0, 1.0]))
    s2.load_state_dict(sd)
    assert s2._get_lr(33) == sched._get_lr(33)


def test_trans_load_param_and_resize_pos_embed_match_reference(tmp_path):
    """vit_pytorch.py:646-690 on a seeded ImageNet-style checkpoint: pos-embed 14x14 -> 16x8, flattened patch weight,
    head / dist keys skipped, wrong-shape tensor skipped."""
    from editor_amd.modeling.make_model import Trans, resize_pos_embed
    g = load_golden("f11_load_param")
    seed, d = int(g["seed"]), 64
    ck = {
        "pos_embed": synth.normal(seed, "ck/pos", (1, 197, d), 0.02),
        "cls_token": synth.normal(seed, "ck/cls", (1, 1, d), 0.02),
        "patch_embed.proj.weight": synth.normal(seed, "ck/pe", (d, 768), 0.05),
        "patch_embed.proj.bias": synth.normal(seed, "ck/peb", (d,), 0.05),
        "blocks.0.attn.qkv.weight": synth.normal(seed, "ck/qkv", (3 * d, d), 0.02),
        "blocks.0.mlp.fc1.weight": synth.normal(seed, "ck/bad", (7, 5), 1.0),
        "norm.weight": synth.normal(seed, "ck/norm", (d,), 1.0),
        "head.weight": synth.normal(seed, "ck/head", (1000, d), 1.0),
        "dist_token": synth.normal(seed, "ck/dist", (1, 1, d), 1.0),
    }
    path = str(tmp_path / "jx_vit_small_p16_224.pth")
    torch.save({"model": ck}, path)
    t = Trans((256, 128), d, 1, 2, 4.0, True, 4, 3.0, 0.0)
    fc1_before = t.state_dict()["blocks.0.mlp.fc1.weight"].clone()
    fc_before = t.state_dict()["fc.weight"].clone()
    t.load_param(path)
    sd = t.state_dict()
    assert torch.equal(sd["pos_embed"], torch.from_numpy(g["pos_embed"]))
    assert torch.equal(sd["cls_token"], torch.from_numpy(g["cls_token"]))
    assert torch.equal(sd["patch_embed.proj.weight"][:6], torch.from_numpy(g["pe_weight"]))
    assert torch.equal(sd["patch_embed.proj.bias"], torch.from_numpy(g["pe_bias"]))
    assert torch.equal(sd["blocks.0.attn.qkv.weight"][:16], torch.from_numpy(g["qkv"]))
    assert torch.equal(sd["norm.weight"], torch.from_numpy(g["norm_w"]))
    assert torch.equal(sd["blocks.0.mlp.fc1.weight"], fc1_before) and torch.equal(sd["fc.weight"], fc_before)
    assert torch.equal(resize_pos_embed(ck["pos_embed"], 24, 8), torch.from_numpy(g["resized_24x8"]))


def _ddp_data():
    data = []
    for pid in range(37):
        for k in range(2 + (pid * 5) % 23):
            data.append((f"img_{pid}_{k}.jpg", pid, k % 4, 0))
    return data


@pytest.mark.parametrize("world", [1, 2, 4])
def test_ddp_identity_sampler_matches_reference(world):
    """data/datasets/sampler_ddp.py:111-196 for every rank of world 1 / 2 / 4 with the shared seed fixed."""
    from editor_amd.data import RandomIdentitySampler_DDP
    g = load_golden("f12_sampler_ddp")
    seen = []
    for rank in range(world):
        s = RandomIdentitySampler_DDP(_ddp_data(), 64, 8, rank=rank, world_size=world, seed=int(g["seed"]))
        idx = np.asarray(list(iter(s)), dtype=np.int64)
        assert np.array_equal(idx, g[f"w{world}r{rank}"]), (world, rank)
        assert len(s) == int(g[f"w{world}r{rank}_len"])
        pids = np.asarray([d[1] for d in _ddp_data()])[idx].reshape(-1, 8)
        assert (pids == pids[:, :1]).all()                # K consecutive instances of one identity
        seen.append(idx)
    if world > 1:                                          # ranks take disjoint mini-batches of one global list
        blocks = [tuple(b) for s in seen for b in s.reshape(-1, 64 // world).tolist()]
        assert len(set(blocks)) == len(blocks)
