"""Row N3: host-side input-pipeline pieces against goldens captured from the reference (CPU), and the device transform
against the oracle (GPU)."""
import random

import numpy as np
import pytest
import torch

from conftest import load_golden
from editor_amd import synth


def _data():
    data = []
    for pid in range(23):
        for k in range(3 + (pid * 7) % 29):
            data.append((f"img_{pid}_{k}.jpg", pid, k % 4, 0))
    return data


def test_identity_sampler_matches_reference():
    from editor_amd.data import RandomIdentitySampler
    g = load_golden("f9_input")
    seed = int(g["seed"])
    random.seed(seed); np.random.seed(seed)
    s = RandomIdentitySampler(_data(), 32, 8)
    order = np.asarray(list(iter(s)), dtype=np.int64)
    assert np.array_equal(order, g["sampler_order"])
    assert len(order) % 32 == 0 and len(s) >= len(order)
    pids = np.asarray([d[1] for d in _data()])[order].reshape(-1, 4, 8)
    assert (pids == pids[:, :, :1]).all()                      # P x K batches: K consecutive instances per identity


def test_erasing_rectangles_match_reference():
    from editor_amd.data import ErasingParams
    g = load_golden("f9_input")
    random.seed(int(g["seed"]) + 1)
    ep = ErasingParams(0.5)
    rects = np.asarray([ep(256, 128) for _ in range(64)], dtype=np.int32)
    assert np.array_equal(rects, g["rects"])
    assert 10 < rects[:, 0].sum() < 54


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,b", [(256, 128, 32), (128, 256, 5), (384, 128, 3)])
def test_device_train_transform_matches_oracle(h, w, b):
    from editor_amd.data import DeviceTrainTransform
    from oracle import augment_ref
    img = synth.integers(5, "aug/img", (b, h, w, 3), 256).to(torch.uint8)
    noise = synth.normal(5, "aug/noise", (b, 3, h, w), 1.0)
    tf = DeviceTrainTransform((h, w), prob=0.5, padding=10, re_prob=0.5)
    random.seed(3); torch.manual_seed(3)
    params = tf.draw(b)
    params[0] = torch.tensor([1, 0, 20, 1, 0, 0, h - 1, w - 1])          # extremes: corner crops, near-full erase
    params[1] = torch.tensor([0, 20, 0, 0, 0, 0, 0, 0])
    ref = augment_ref.train_transform(img, params, 10, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5), noise)
    out = tf(img.cuda(), params, noise.cuda())
    assert torch.equal(out.cpu(), ref)                                      # byte / fp32 arithmetic: bit-exact
    # device-generated fill: untouched pixels identical, erased ones ~ N(0,1)
    out2 = tf(img.cuda(), params, None, seed=11).cpu()
    erased = torch.zeros(b, h, w, dtype=torch.bool)
    for i in range(b):
        _, _, _, e, t, l, eh, ew = [int(v) for v in params[i]]
        if e:
            erased[i, t:t + eh, l:l + ew] = True
    m = erased[:, None].expand(-1, 3, -1, -1)
    assert torch.equal(out2[~m], ref[~m])
    z = out2[m]
    assert abs(z.mean().item()) < 0.02 and abs(z.std().item() - 1) < 0.02
    with pytest.raises(RuntimeError):
        tf(img, params, noise)


def _fpc_cases():
    g = load_golden("f19_flip_pad_crop")
    seed = int(g["seed"])
    for ci in range(int(g["n"])):
        h, w, pad, n = [int(v) for v in g["case%d" % ci]]
        img = synth.integers(seed, "fpc/%d" % ci, (n, h, w, 3), 256).to(torch.uint8)
        p3 = torch.from_numpy(g["params%d" % ci].astype(np.int64))
        params = torch.zeros(n, 8, dtype=torch.int64)
        params[:, :3] = p3                                     # flip, top, left; no erase
        yield ci, g, img, params, pad


def _fpc_check(out, g, ci):
    out = out.double().numpy()
    assert np.array_equal(out[:, :, ::5, ::3].astype(np.float32), g["out%d" % ci])
    assert np.array_equal(out.sum(axis=(1, 2, 3)), g["sum%d" % ci])
    wts = np.arange(out[0].size).reshape(out[0].shape) % 251 + 1
    assert np.array_equal((out * wts).sum(axis=(1, 2, 3)), g["wsum%d" % ci])


def test_flip_pad_crop_oracle_matches_pillow():
    """F19: what the train transform's flip / pad / crop / ToTensor / Normalize do to the pixels for GIVEN draws, pinned to Pillow +
    torch - the implementations torchvision 0.14.1 delegates to for PIL inputs (tests/golden/capture_golden.py f19).  The oracle's
    restatement reproduces every value (subsampled image + two checksums per sample); only the ORDER of the draws stays restated."""
    from oracle import augment_ref
    for ci, g, img, params, pad in _fpc_cases():
        n, h, w, _ = img.shape
        ref = augment_ref.train_transform(img, params, pad, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5), torch.zeros(n, 3, h, w))
        _fpc_check(ref, g, ci)


@pytest.mark.gpu
def test_device_flip_pad_crop_matches_pillow():
    """The device transform (editor_augment_u8) against the same Pillow-made fixture, bit for bit."""
    from editor_amd.data import DeviceTrainTransform
    for ci, g, img, params, pad in _fpc_cases():
        n, h, w, _ = img.shape
        tf = DeviceTrainTransform((h, w), prob=0.5, padding=pad, re_prob=0.5)
        out = tf(img.cuda(), params, torch.zeros(n, 3, h, w, device="cuda"))
        _fpc_check(out.cpu(), g, ci)


# ---- T.Resize (Pillow ImagingResample) ----------------------------------------------------------------------------
def _resize_cases():
    g = load_golden("f13_resize")
    for i in range(int(g["n"])):
        h, w, oh, ow, ip = [int(v) for v in g["case%d" % i]]
        a = synth.integers(int(g["seed"]), "resize/%d" % i, (h, w, 3), 256).numpy().astype(np.uint8)
        yield i, g, a, (oh, ow), ip


def _resize_matches_golden(out, g, i):
    assert np.array_equal(out[::3, ::3], g["out%d" % i])
    assert int(out.astype(np.int64).sum()) == int(g["sum%d" % i])
    assert int(np.bitwise_xor.reduce(out.astype(np.int64).ravel() * (np.arange(out.size) % 251 + 1))) == int(g["xor%d" % i])


def test_resize_oracle_matches_pillow_golden():
    """oracle/resize_ref.py (restated Pillow algorithm) == outputs Pillow produced (fixture), bit for bit."""
    from oracle import resize_ref
    for i, g, a, size, ip in _resize_cases():
        _resize_matches_golden(resize_ref.resize(a, size, ip), g, i)


def test_resize_oracle_matches_pillow_live():
    """...and == Pillow run here, when it is importable (it is in this image), on more shapes."""
    PIL = pytest.importorskip("PIL.Image")
    from oracle import resize_ref
    rng = np.random.default_rng(3)
    for (h, w), (oh, ow) in [((300, 150), (256, 128)), ((40, 500), (128, 256)), ((1000, 400), (384, 128)), ((256, 128), (512, 256))]:
        for ip in (2, 3):
            a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            assert np.array_equal(resize_ref.resize(a, (oh, ow), ip), np.asarray(PIL.fromarray(a).resize((ow, oh), ip)))


def test_resize_coefficient_tables_match_oracle():
    """the product's host-side tap tables (editor_amd.data.resize_coeffs, vectorised) == the oracle's scalar restatement"""
    from editor_amd.data import resize_coeffs
    from oracle import resize_ref
    for n_in, n_out in [(256, 256), (150, 128), (64, 128), (233, 128), (517, 384), (61, 128), (1000, 256), (128, 512)]:
        for ip in (2, 3):
            b0, k0 = resize_ref.precompute_coeffs(n_in, n_out, ip)
            b1, k1 = resize_coeffs(n_in, n_out, ip)
            assert np.array_equal(b0, b1) and np.array_equal(k0, k1), (n_in, n_out, ip)


@pytest.mark.gpu
def test_device_resize_bit_exact():
    """editor_resize_u8 == the oracle == Pillow: golden cases + batched random cases incl. single-axis and identity sizes."""
    from editor_amd.data import DeviceResize
    from oracle import resize_ref
    for i, g, a, size, ip in _resize_cases():
        out = DeviceResize(size, ip)(torch.from_numpy(a)[None].cuda())[0].cpu().numpy()
        _resize_matches_golden(out, g, i)
    rng = np.random.default_rng(5)
    for (h, w), size in [((300, 150), (256, 128)), ((256, 64), (256, 128)), ((100, 128), (256, 128)), ((256, 128), (256, 128)),
                         ((517, 233), (384, 128))]:
        batch = rng.integers(0, 256, (5, h, w, 3), dtype=np.uint8)
        out = DeviceResize(size, 3)(torch.from_numpy(batch).cuda()).cpu().numpy()
        for j in range(5):
            assert np.array_equal(out[j], resize_ref.resize(batch[j], size, 3)), (h, w, size, j)
