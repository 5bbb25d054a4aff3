"""BASELINE.json config 5 (synthetic 4-modal ViT-L/16, 512 patch tokens): an extension of the reference, whose factory
has no ViT-L and whose forward hard-codes three modalities (make_model.py:153-155,363-368).  What CAN be pinned to the
reference is pinned: single Block / BlockMask in->out pairs at D = 1024, 16 heads, T = 513 from the reference's own classes
(golden f6_blocks_large).  The 4-modal whole is checked against the oracle's N-modality form (the same
computation with one more term in every per-modality loop, oracle/editor_ref.py MODALITIES4) - "parity unpinned" for the
4th modality by construction, there is no reference for it.  The long-sequence (T > 608) attention kernels that the joint
HMA block of this configuration needs are checked against a plain fp32 softmax attention."""
import pytest
import torch

from conftest import load_golden, rel_err, t
from editor_amd import config, synth

pytestmark = pytest.mark.gpu


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _attn_ref(qkv, lens, heads, hd):
    d = heads * hd
    outs, r0 = [], 0
    for ln in lens:
        q, k, v = (qkv[r0:r0 + ln, i * d:(i + 1) * d].reshape(ln, heads, hd).transpose(0, 1) for i in range(3))
        p = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1)
        outs.append((p @ v).transpose(0, 1).reshape(ln, d))
        r0 += ln
    return torch.cat(outs, 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("lens", [[700], [1539, 513], [2052], [609, 2052, 64, 1, 1000]])
def test_attention_long_sequences(dtype, lens):
    """T > 608: chunked kernels (64 own rows per workgroup, 256-row LDS chunks), dense and packed (cu) forms, fwd + bwd."""
    from editor_amd import ops
    heads, hd = 16, 64
    d = heads * hd
    total = sum(lens)
    dense = len(lens) == 1
    rows = total if dense else (total + 63) // 64 * 64
    g = _g(5)
    qkv = (torch.randn(rows, 3 * d, generator=g) * 1.0).to(dtype)
    do = torch.randn(rows, d, generator=g).to(dtype)
    qkv[total:] = 0
    do[total:] = 0
    qr = qkv.float().requires_grad_(True)
    o_ref = _attn_ref(qr, lens, heads, hd)
    o_ref.backward(do.float()[:total])
    cu = None if dense else torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32).cuda()
    o, lse = ops.attention_fwd(qkv.cuda(), len(lens), max(lens), heads, hd, None, None, cu=cu)
    tol = 1.2e-2 if dtype == torch.bfloat16 else 1.5e-3
    assert rel_err(o[:total].float().cpu(), o_ref.detach()) < tol
    assert float(o[total:].float().abs().max()) == 0.0 if not dense else True
    dqkv = ops.attention_bwd(qkv.cuda(), do.cuda(), len(lens), max(lens), heads, hd, None, lse, o, cu=cu)
    assert rel_err(dqkv[:total].float().cpu(), qr.grad[:total]) < 2 * tol
    assert torch.isfinite(dqkv.float()).all()


def test_attention_513_tokens_key_validity():
    """513..608-token sequences run in the whole-sequence kernels with more than 128 validity bits (a two-word bitmap
    aliased keys >= 512): dense T = 513 and a masked T = 579 row set."""
    from editor_amd import ops
    heads, hd = 16, 64
    d = heads * hd
    for tlen, use_mask in ((513, False), (579, True), (600, False)):
        b = 2
        g = _g(tlen)
        qkv = (torch.randn(b * tlen, 3 * d, generator=g)).bfloat16()
        mask = None
        if use_mask:
            mask = (torch.rand(b, tlen, generator=g) > 0.4).to(torch.uint8)
            mask[:, 0] = 1
        q, k, v = (qkv.float()[:, i * d:(i + 1) * d].reshape(b, tlen, heads, hd).transpose(1, 2) for i in range(3))
        s = (q @ k.transpose(-2, -1)) * hd ** -0.5
        if mask is not None:
            mm = mask.float().view(b, 1, tlen, 1)
            s = s.masked_fill((mm @ mm.transpose(-2, -1)) == 0, -65504.0)
            p = s.softmax(-1) * mm
        else:
            p = s.softmax(-1)
        ref = (p @ v).transpose(1, 2).reshape(b * tlen, d)
        o, _ = ops.attention_fwd(qkv.cuda(), b, tlen, heads, hd, None if mask is None else mask.cuda(), None)
        assert rel_err(o.float().cpu(), ref) < 1.2e-2, tlen


def _large_model(dtype, nmod=3, size=(512, 256), seed=43):
    from editor_amd.modeling import make_model
    cfg, c, cams = config.preset("SYNTH4L", compute_dtype=dtype, drop_path=0.0, num_modalities=nmod, size_train=size)
    m = make_model(cfg, 8 if nmod == 3 else c, cams)
    return m, cfg, cams


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-4), ("f16", 1.5e-3), ("bf16", 1.2e-2)])
def test_block_and_blockmask_d1024_match_reference_golden(dtype, tol):
    """F6 at D = 1024 / 16 heads / T = 513 from the reference's Block and BlockMask classes."""
    from editor_amd import functional as fn
    from editor_amd.modeling.make_model import _block_args
    g = load_golden("f6_blocks_large")
    seed = int(g["seed"])
    m, cfg, cams = _large_model(dtype, nmod=3)
    d, tk = 1024, 513
    blk = m.BACKBONE.base.blocks[0]
    synth.fill_state_dict_(blk.state_dict(), seed)
    synth.fill_state_dict_(m.FUSE_block.state_dict(), seed + 1)
    m = m.cuda().eval()
    act = m.act_dtype
    x = synth.normal(seed, "blkL/x", (2, tk, d), 1.0).cuda()
    with torch.no_grad():
        y = fn.TransformerBlockFn.apply(x, *_block_args(blk.norm1, blk.attn, blk.norm2, blk.mlp), None, None, 16, 1e-6, act,
                                        None, None)
        assert rel_err(y[:, ::64, :64].cpu(), g["block_out"]) < tol
        assert abs(y.norm().item() / float(g["block_out_norm"]) - 1) < tol
        feats = [synth.normal(seed, "hmaL/%d" % i, (2, tk, d), 1.0) for i in range(3)]
        idx = synth.integers(seed, "hmaL/mask", (2, tk - 1), 2).bool()
        fs = torch.stack([torch.cat([f[:, :1], f[:, 1:] * idx.unsqueeze(-1)], 1) for f in feats]).cuda()
        z, _ = m._hma(fs, idx.to(torch.uint8).cuda(), None)            # dense-masked form (1539 joint tokens)
        assert rel_err(z[:, ::96, :64].cpu(), g["hma_out"]) < tol
        assert abs(z.norm().item() / float(g["hma_out_norm"]) - 1) < tol


def test_config5_4modal_vitl_eval_vs_oracle(oracle):
    """4 modalities x ViT-L/16 x 512 patch tokens, eval, B = 2: f32 parity mode against the oracle's 4-modal form (index
    bit-exact, features 1e-3), then f16 / bf16 with the selection teacher-forced (compacted HMA, joint block up to
    2052 tokens -> long-sequence attention kernels)."""
    import os
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    b, seed = 2, 47
    m, cfg, cams = _large_model("f32", nmod=4)
    synth.fill_state_dict_(m.state_dict(), seed)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    img, label, cam, view = synth.make_batch(seed + 1, b, 512, 256, cams, instances=1, keys=config.MODALITY_KEYS)
    with torch.no_grad():
        ref, aux = oracle.editor_forward(sd, img, cam, training=False, al=0, heads=16, hma_heads=16,
                                         modalities=oracle.MODALITIES4, return_aux=True)
    m = m.cuda().eval()
    gimg = {k: v.cuda() for k, v in img.items()}
    with torch.no_grad():
        out = m(gimg, cam_label=cam.cuda(), view_label=view.cuda())
    assert out.shape == (b, 4 * 1024)
    assert torch.equal(m.last_aux["mask_fre"].cpu().bool(), aux["mask_fre"])
    for i in range(4):
        assert torch.equal(m.last_aux["attn_masks"][i].cpu().bool(), aux["attn_masks"][i]), i
    assert torch.equal(m.last_aux["index"].cpu().bool(), aux["index"])
    err = rel_err(out.cpu(), ref)
    print("config5 f32 cls4t rel err:", err)
    assert err < 1e-3
    del m
    for dtype, tol in (("f16", 1.5e-3), ("bf16", 1.5e-2)):
        m2, _, _ = _large_model(dtype, nmod=4)
        m2.load_state_dict(sd)
        m2 = m2.cuda().eval()
        m2.teacher_index = aux["index"]
        with torch.no_grad():
            out2 = m2(gimg, cam_label=cam.cuda(), view_label=view.cuda())
        e2 = rel_err(out2.cpu(), ref)
        print("config5", dtype, "cls4t rel err (teacher-forced):", e2)
        assert e2 < tol
        del m2
    # split-precision forward: FREE-RUNNING - 513-token sequences take the chunked split attention kernel, the joint HMA
    # block up to 2052 tokens; selection identical to the oracle's, features 1e-4
    m3, _, _ = _large_model("f16x2", nmod=4)
    m3.load_state_dict(sd)
    m3 = m3.cuda().eval()
    with torch.no_grad():
        out3 = m3(gimg, cam_label=cam.cuda(), view_label=view.cuda())
    for i in range(4):
        assert torch.equal(m3.last_aux["attn_masks"][i].cpu().bool(), aux["attn_masks"][i]), i
    assert torch.equal(m3.last_aux["index"].cpu().bool(), aux["index"])
    e3 = rel_err(out3.cpu(), ref)
    print("config5 f16x2 cls4t rel err (free-running, selection identical):", e3)
    assert e3 < 1e-4


def test_config5_4modal_vitl_train_step_vs_oracle(oracle):
    """4 modalities x ViT-L/16 training step (forward, real loss head, backward) at a small geometry (256x128 -> 129
    tokens, B = 4) against the oracle's 4-modal form: all 11 outputs, loss, gradients of 10 parameters (f32 parity mode)."""
    import os
    from editor_amd import losses
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    b, seed = 4, 53
    m, cfg, cams = _large_model("f32", nmod=4, size=(256, 128))
    synth.fill_state_dict_(m.state_dict(), seed)
    keys = ["BACKBONE.base.blocks.0.attn.qkv.weight", "BACKBONE.base.blocks.23.mlp.fc2.weight", "BACKBONE.base.cls_token",
            "BACKBONE.base.patch_embed.proj.bias", "FUSE_block.attnM4.qkv.weight", "FUSE_block.mlpM4.fc2.weight",
            "FUSE_block.attn1.proj.weight", "M4_REDUCE.weight", "FUSE_HEAD.weight", "BACKBONE_HEAD.weight"]
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    for k in keys:
        sd[k].requires_grad_(True)
    img, label, cam, view = synth.make_batch(seed + 1, b, 256, 128, cams, instances=2, keys=config.MODALITY_KEYS)
    out_ref = oracle.editor_forward(sd, img, cam, label=label, training=True, al=0, heads=16, hma_heads=16,
                                    modalities=oracle.MODALITIES4)
    loss_ref = oracle.loss_pairs(out_ref, label)
    loss_ref.backward()

    class W:
        def add_scalar(self, *a, **k):
            pass
    m = m.cuda().train()
    out = m({k: v.cuda() for k, v in img.items()}, label=label.cuda(), cam_label=cam.cuda(), view_label=view.cuda(),
            writer=W(), epoch=1)
    assert len(out) == 11
    for i, (a, r) in enumerate(zip(out, out_ref)):
        assert rel_err(a.detach().cpu(), r.detach()) < 1e-3, i
    loss = losses.loss_pairs(out, label.cuda())
    loss.backward()
    assert abs(loss.item() / loss_ref.item() - 1) < 1e-4
    named = dict(m.named_parameters())
    for k in keys:
        assert rel_err(named[k].grad.cpu(), sd[k].grad) < 3e-3, k
    cen = m.FUSE_block.memory_cls.M4_centers[label.unique().cuda()]
    assert rel_err(cen.cpu(), sd["FUSE_block.memory_cls.M4_centers"][label.unique()]) < 1e-4
