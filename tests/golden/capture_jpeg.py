"""Writes tests/golden/f14_decode.npz: baseline JPEG files encoded by Pillow (the reference's own decoder library,
data/datasets/bases.py:19 `Image.open(path).convert('RGB')`) from seeded synthetic images, with PILLOW'S decoded pixels as
the expected output - the pin of the JPEG decode (host Huffman decoder + device reconstruction, editor_amd/csrc/jpeg.hip)
and of oracle/jpeg_ref.py.  Run in the build container (Pillow 12.2.0, libjpeg-turbo):  python tests/golden/capture_jpeg.py"""
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [  # (name, W, H, subsampling | 'gray', quality, extra save options)
    ("stitched_420_q75", 768, 128, 2, 75, {}),            # the dataset's layout: three 256-wide modalities side by side
    ("stitched_444_q90", 768, 128, 0, 90, {}),
    ("stitched_422_q85", 768, 128, 1, 85, {}),
    ("odd_420_q60", 250, 131, 2, 60, {}),                 # partial MCUs on both edges
    ("odd_422_q95", 251, 77, 1, 95, {}),
    ("tiny_420_q50", 33, 17, 2, 50, {}),
    ("restart_420_q80", 264, 72, 2, 80, dict(restart_marker_blocks=7)),
    ("opt_444_q80", 264, 72, 0, 80, dict(optimize=True)),
    ("gray_q80", 300, 100, "gray", 80, {}),
]


def synth(rng, w, h, gray):
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 17.0) * np.cos(yy / 11.0), 128 + 90 * np.cos(xx / 29.0 + yy / 7.0),
                     255.0 * (xx + yy) / (w + h)], axis=2)
    base += rng.normal(0, 25, base.shape)
    a = np.clip(base, 0, 255).astype(np.uint8)
    return Image.fromarray(a[..., 0] if gray else a)


def main():
    rng = np.random.default_rng(14)
    out = {}
    for name, w, h, ss, q, kw in CASES:
        im = synth(rng, w, h, ss == "gray")
        bio = io.BytesIO()
        if ss == "gray":
            im.save(bio, "JPEG", quality=q, **kw)
        else:
            im.save(bio, "JPEG", quality=q, subsampling=ss, **kw)
        data = bio.getvalue()
        out[name + ".jpg"] = np.frombuffer(data, dtype=np.uint8)
        out[name + ".rgb"] = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    # a progressive file: must be refused (EDITOR_JPEG_UNSUPPORTED), never mis-decoded
    bio = io.BytesIO()
    synth(rng, 64, 48, False).save(bio, "JPEG", quality=80, progressive=True)
    out["progressive.jpg"] = np.frombuffer(bio.getvalue(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "f14_decode.npz"), **out)
    print("wrote f14_decode.npz:", {k: v.shape for k, v in out.items() if k.endswith(".rgb")})


if __name__ == "__main__":
    main()
