"""Deterministic synthetic data for the EDITOR hot path (inputs AND weights).

Counter-based splitmix64 so that (seed, name) -> tensor is identical on every
host, numpy version and device; the golden fixtures under tests/golden store
only (seed, cfg) and the expected OUTPUTS.  Image statistics follow the
reference's input pipeline: uint8 pixels normalised by mean=std=0.5
(/root/reference/data/datasets/make_dataloader.py:251, config/defaults.py:70-72).
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on a uint64 array (wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _stream(seed, name, n):
    """n uint64 words of the stream identified by (seed, name)."""
    tag = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([np.uint64(seed) * np.uint64(0x100000001B3) + tag],
                                    dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + base
    return _splitmix64(idx)


def uniform01(seed, name, shape):
    """float64 uniform in [0,1) with 53 random bits."""
    n = int(np.prod(shape)) if len(shape) else 1
    w = _stream(seed, name, n)
    return ((w >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)).reshape(shape)


def uniform(seed, name, shape, lo=-1.0, hi=1.0):
    return torch.from_numpy((lo + (hi - lo) * uniform01(seed, name, shape)).astype(np.float32))


def normal(seed, name, shape, std=1.0):
    """Box-Muller on two independent uniform streams (float64 -> float32)."""
    u1 = uniform01(seed, name + "#a", shape)
    u2 = uniform01(seed, name + "#b", shape)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy((std * z).astype(np.float32))


def uint8_image(seed, name, shape):
    """Seeded uniform uint8 pixels -> (k/255 - 0.5)/0.5 fp32 NCHW."""
    w = _stream(seed, name, int(np.prod(shape)))
    k = (w >> np.uint64(56)).astype(np.float32).reshape(shape)
    return torch.from_numpy(((k / np.float32(255.0)) - np.float32(0.5)) / np.float32(0.5))


def integers(seed, name, shape, high):
    w = _stream(seed, name, int(np.prod(shape)) if len(shape) else 1)
    return torch.from_numpy((w % np.uint64(high)).astype(np.int64).reshape(shape))


def make_batch(seed, batch, height, width, cams, instances=16, smooth=False, keys=("RGB", "NI", "TI")):
    """Tri-modal synthetic batch in the reference's collate layout
    (/root/reference/engine/processor.py:73-81): dict of (B,3,H,W) fp32 +
    label / cam_label / view_label int64.  label = P identities x K contiguous
    instances (the layout OCFR assumes, modeling/fusion_part/OCFR.py:33-39)."""
    img = {}
    for key in keys:
        img[key] = uint8_image(seed, "img/" + key, (batch, 3, height, width))
        if smooth:
            # low-pass variant: makes the per-patch positive counts spread out
            # (fewer ties) - used by a few parity cases, never by the bench.
            k = torch.nn.functional.avg_pool2d(img[key], 5, 1, 2)
            img[key] = (k * 3.0).clamp(-1, 1)
    inst = min(instances, batch)
    p = max(batch // inst, 1)
    label = torch.arange(p, dtype=torch.int64).repeat_interleave(inst)[:batch]
    cam = integers(seed, "cam", (batch,), max(cams, 1))
    view = torch.zeros(batch, dtype=torch.int64)
    return img, label, cam, view


def fill_state_dict_(state_dict, seed):
    """Overwrite every floating tensor of a name-compatible state dict in place
    with seeded values, keyed by parameter NAME (so the reference module and
    this repo's module receive identical weights).  Scales are chosen to keep
    activations O(1) like a trained ViT: weights ~ N(0, 0.02) (trunc_normal_
    std of /root/reference/modeling/backbones/vit_pytorch.py:528-534), norm
    gains 1 +- 0.1, biases / embeddings N(0, 0.02)."""
    for name, t in state_dict.items():
        if not torch.is_floating_point(t):
            continue
        shape = tuple(t.shape)
        leaf = name.rsplit(".", 1)[-1]
        if "running_var" in name:
            v = 1.0 + 0.1 * uniform(seed, name, shape, 0.0, 1.0)
        elif "running_mean" in name:
            v = normal(seed, name, shape, 0.02)
        elif leaf == "weight" and t.dim() == 1:      # LayerNorm / BatchNorm gain
            v = 1.0 + 0.1 * uniform(seed, name, shape)
        elif "centers" in name:
            v = normal(seed, name, shape, 0.03)
        elif name.startswith("FREQ_INDEX."):
            continue                                   # Haar taps are constants
        elif "HEAD.weight" in name:
            v = normal(seed, name, shape, 0.02)
        elif "REDUCE.weight" in name:
            v = normal(seed, name, shape, 0.03)
        else:
            v = normal(seed, name, shape, 0.02)
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
    return state_dict
