"""GPU parity of the token-selection kernels against the oracle + the reference's golden fixtures.
All calls go through the C ABI (editor_amd.ops -> libeditor_hip.so)."""
import ctypes

import pytest
import torch

from conftest import load_golden, rel_err, t
from editor_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from editor_amd import ops
    return ops


@pytest.mark.parametrize("tag,hw", [("256x128", (256, 128)), ("128x256", (128, 256)), ("384x128", (384, 128))])
@pytest.mark.parametrize("kind", ["u8", "smooth"])
def test_frequency_golden(ops, tag, hw, kind):
    g = load_golden(f"f1_freq_{tag}_{kind}")
    img, _, _, _ = synth.make_batch(int(g["seed"]), 128, hw[0], hw[1], 2, smooth=bool(g["smooth"]))
    mask, counts = ops.frequency_mask(img["RGB"].cuda(), img["NI"].cuda(), img["TI"].cuda(), 10)
    assert torch.equal(counts.cpu(), t(g["counts"]))
    assert torch.equal(mask.cpu().bool(), t(g["mask"]))


def test_frequency_two_modalities(ops, oracle):
    img, _, _, _ = synth.make_batch(5, 16, 256, 128, 2)
    counts = ops.freq_counts(img["RGB"].cuda(), img["NI"].cuda(), None)
    ref, _ = oracle.frequency_counts(img["RGB"], img["NI"], None)
    assert torch.equal(counts.cpu(), ref)


def test_frequency_special_values_and_odd_grids(ops, oracle):
    """The 4x4-pixels-per-lane kernel (round 4) divides by the modality count with the three-instruction correctly-rounded form
    and keeps the IEEE sequence for what that form does not cover: flat regions (exact zeros in every detail band), saturated
    pixels, denormal and huge magnitudes, an infinity, and a patch grid whose width is not a multiple of the four patches a wave
    takes (W = 48: three patches per row) must all count exactly as the oracle does; four modalities and two take other
    instantiations."""
    g = torch.Generator().manual_seed(41)
    b, h, w = 6, 64, 48
    base = torch.rand(3, b, 3, h, w, generator=g) * 2 - 1
    base[:, 0, :, :32] = 0.0                                   # flat zero region
    base[:, 1, :, 16:48, 16:32] = 1.0                          # saturated block (all modalities agree: zero detail bands)
    base[:, 2] *= 1e-41                                        # denormal pixels -> denormal coefficients
    base[:, 3] *= 1e30                                         # large magnitudes
    base[1, 4, 1, 5, 7] = float("inf")                         # an infinity in one patch
    base[:, 5, :, ::2] = base[:, 5, :, 1::2]                   # row pairs equal: level-1 row-high bands exactly zero
    r, n_, t_ = base[0].contiguous(), base[1].contiguous(), base[2].contiguous()
    ref, _ = oracle.frequency_counts(r, n_, t_)
    got = ops.freq_counts(r.cuda(), n_.cuda(), t_.cuda())
    assert torch.equal(got.cpu(), ref)
    ref2, _ = oracle.frequency_counts(r, n_, None)
    assert torch.equal(ops.freq_counts(r.cuda(), n_.cuda(), None).cpu(), ref2)
    m4 = (torch.rand(b, 3, h, w, generator=g) * 2 - 1)
    ref4, _ = oracle.frequency_counts(r[:4], n_[:4], t_[:4], extra=(m4[:4],))
    assert torch.equal(ops.freq_counts(r[:4].cuda(), n_[:4].cuda(), t_[:4].cuda(), m4[:4].cuda()).cpu(), ref4)


@pytest.mark.parametrize("n", [128, 192, 512])
@pytest.mark.parametrize("k", [1, 2, 10])
def test_topk_tie_torture(ops, oracle, n, k):
    g = torch.Generator().manual_seed(n + k)
    for x in (torch.randint(90, 110, (1000, n), generator=g, dtype=torch.int32),
              torch.randint(0, 4, (1000, n), generator=g).float(),
              torch.rand(1000, n, generator=g)):
        ref = oracle.topk_mask(x, k)
        got = ops.topk_mask(x.cuda(), k).cpu().bool()
        assert torch.equal(ref, got)


def test_topk_wide_rows_and_nan(ops, oracle):
    """One wave per row (round 4): rows longer than a wave's first pass (n = 2048: the one-row-per-workgroup form), k > 64
    (the mask write loops), NaNs (ordered first by torch.topk) and the introselect depth limit (sorted / organ-pipe rows)."""
    g = torch.Generator().manual_seed(77)
    x = torch.rand(40, 2048, generator=g)
    x[::3, ::7] = float("nan")
    for k in (3, 70, 300):
        assert torch.equal(oracle.topk_mask(x, k), ops.topk_mask(x.cuda(), k).cpu().bool()), k
    asc = torch.arange(512, dtype=torch.int32).repeat(8, 1)
    pipe = torch.cat([torch.arange(256), torch.arange(255, -1, -1)]).int().repeat(8, 1)
    for xx in (asc, asc.flip(1).contiguous(), pipe, torch.zeros(8, 512, dtype=torch.int32)):
        for k in (10, 100):
            assert torch.equal(oracle.topk_mask(xx, k), ops.topk_mask(xx.cuda(), k).cpu().bool())


def test_topk_group_or(ops, oracle):
    g = torch.Generator().manual_seed(3)
    x = torch.rand(24 * 12, 128, generator=g)
    ref = oracle.topk_mask(x, 2).reshape(24, 12, 128).any(1)
    got = ops.topk_mask(x.cuda(), 2, group=12).cpu().bool()
    assert torch.equal(ref, got)


def test_rollout_and_part_attention(ops, oracle):
    g = torch.Generator().manual_seed(9)
    L, B, H, T = 12, 6, 12, 129
    probs = torch.softmax(torch.randn(L, B, H, T, T, generator=g) * 2.0, dim=-1)
    ref = oracle.rollout_scores([probs[i] for i in range(L)])
    got = ops.attn_rollout(probs.cuda())
    assert ((got.cpu() - ref).abs() / ref.abs()).max() < 2e-5
    # selection on the ORACLE's scores is bit-exact (stage-wise protocol)
    m_ref = oracle.part_attention_mask(ref, 2)
    m_got = ops.topk_mask(ref.reshape(B * H, T - 1).cuda(), 2, group=H).cpu().bool()
    assert torch.equal(m_ref, m_got)


def test_probe_tr16(ops):
    """Documents ds_read_b64_tr_b16 semantics on gfx950 (used by the bf16 GEMM's transposed operands)."""
    from editor_amd import _lib
    lane = torch.arange(64)
    i = lane & 15
    # lane i of each 16-lane group points at row (i>>2), 8-byte chunk (i&3) of a [4][16] u16 block (row = 64 B)
    addr = ((lane >> 4) * 512 + (i >> 2) * 64 + (i & 3) * 8).int().cuda()
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    rc = _lib.probe_lib().editor_probe_tr16(ctypes.c_void_p(addr.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    got = out.cpu().view(64, 4).long()
    print("tr16 lane0..17:", got[:18].tolist())
    # hypothesis A: lane i receives column i of the block: elements [r][i], r = 0..3  (u16 index = g*256 + r*32 + i)
    exp = (lane >> 4).view(64, 1) * 256 + torch.arange(4).view(1, 4) * 32 + i.view(64, 1)
    assert torch.equal(got, exp), "ds_read_b64_tr_b16 semantics differ from the assumed map"


def test_probe_mfma16(ops):
    from editor_amd import _lib
    g = torch.Generator().manual_seed(1)
    a = torch.randn(16, 32, generator=g).bfloat16().float()
    b = torch.randn(32, 16, generator=g).bfloat16().float()
    d = torch.zeros(16, 16, device="cuda")
    ag, bg = a.cuda(), b.cuda()
    rc = _lib.probe_lib().editor_probe_mfma16(ctypes.c_void_p(ag.data_ptr()), ctypes.c_void_p(bg.data_ptr()),
                                              ctypes.c_void_p(d.data_ptr()),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    assert torch.allclose(d.cpu(), a @ b, atol=1e-4, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("t", [129, 193])
def test_rollout_recomputed_from_qk_matches_materialised(t):
    """bf16 backbone form: rollout steps that recompute P from (qkv, lse) == the rollout over the probabilities the
    forward writes (same bf16 products; only exp2(s - lse) vs exp2(s - max)/sum differs, by rounding)."""
    from editor_amd import ops
    b, heads, hd, layers = 6, 12, 64, 5
    g = torch.Generator().manual_seed(3)
    ldp = (t + 3) // 4 * 4
    probs = torch.empty(layers, b, heads, t, ldp, device="cuda")
    pairs = []
    for l in range(layers):
        qkv = (torch.randn(b * t, 3 * heads * hd, generator=g) * 0.7).bfloat16().cuda()
        _, lse = ops.attention_fwd(qkv, b, t, heads, hd, None, probs[l])
        pairs.append((qkv, lse))
    ref = ops.attn_rollout(probs)
    got = ops.attn_rollout_qk(pairs, b, t, heads, hd)
    assert got.shape == ref.shape == (b, heads, t - 1)
    assert rel_err(got.cpu(), ref.cpu()) < 2e-5
    k = 2
    same = (got.topk(k, dim=-1).indices.sort(-1).values == ref.topk(k, dim=-1).indices.sort(-1).values).all(-1)
    assert same.float().mean().item() > 0.98                      # selections agree except on near-ties


@pytest.mark.gpu
def test_probe_mfma_f16_keeps_subnormal_operands():
    """The split-precision forward (COMPUTE_DTYPE 'f16x2') keeps low-order parts of its operands in IEEE half, where small
    ones are SUBNORMAL (|x| < 2^-14): pins that v_mfma_f32_16x16x32_f16 multiplies them exactly instead of flushing."""
    from editor_amd import _lib
    a = torch.zeros(16, 32, dtype=torch.float16)
    b = torch.zeros(32, 16, dtype=torch.float16)
    a[:, 0] = 2.0 ** -20                    # subnormal half (min normal 2^-14, min subnormal 2^-24)
    a[:, 1] = 2.0 ** -24
    a[:, 2] = 3 * 2.0 ** -24
    b[0, :] = 1024.0
    b[1, :] = 2.0 ** 14
    b[2, :] = 2.0 ** -24                    # subnormal x subnormal: 3 * 2^-48, representable in the fp32 accumulator
    exp = (a.double() @ b.double()).float()
    assert exp[0, 0] > 0
    d = torch.zeros(16, 16, device="cuda")
    ag, bg = a.view(torch.int16).cuda(), b.view(torch.int16).cuda()
    rc = _lib.probe_lib().editor_probe_mfma16_raw(ctypes.c_void_p(ag.data_ptr()), ctypes.c_void_p(bg.data_ptr()),
                                                  ctypes.c_void_p(d.data_ptr()), 1,
                                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    print("mfma f16 subnormal probe:", d[0, 0].item(), "expected", exp[0, 0].item())
    assert torch.equal(d.cpu(), exp), "the matrix core flushes subnormal half operands"
