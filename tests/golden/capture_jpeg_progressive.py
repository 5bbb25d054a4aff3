"""Writes tests/golden/f15_decode_progressive.npz: PROGRESSIVE JPEG files (SOF2: spectral selection + successive approximation)
encoded by Pillow from seeded synthetic images, with PILLOW'S decoded pixels as the expected output - the reference opens its
images with `Image.open(path).convert('RGB')` (data/datasets/bases.py:19), which reads progressive files like any other.
Pins the progressive branch of the host entropy decoder (editor_amd/csrc/jpeg.hip) through oracle/jpeg_ref.py's and the device's
reconstruction.  Run in the build container (Pillow 12.2.0, libjpeg-turbo):  python tests/golden/capture_jpeg_progressive.py"""
import io
import os

import numpy as np
from PIL import Image

from capture_jpeg import synth

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [  # (name, W, H, subsampling | 'gray', quality, extra save options)
    ("prog_stitched_420_q75", 768, 128, 2, 75, {}),       # the dataset's layout
    ("prog_444_q92", 200, 96, 0, 92, {}),
    ("prog_odd_422_q85", 251, 77, 1, 85, {}),             # partial MCUs on both edges
    ("prog_odd_420_q40", 250, 131, 2, 40, {}),            # coarse quantisation: long end-of-band runs
    ("prog_tiny_420_q50", 33, 17, 2, 50, {}),
    ("prog_restart_420_q80", 264, 72, 2, 80, dict(restart_marker_blocks=5)),
    ("prog_gray_q80", 300, 100, "gray", 80, {}),
    ("prog_flat_444_q95", 64, 48, 0, 95, dict(flat=True)),   # constant image: every AC scan is one long end-of-band run
]


def main():
    rng = np.random.default_rng(15)
    out = {}
    for name, w, h, ss, q, kw in CASES:
        kw = dict(kw)
        if kw.pop("flat", False):
            im = Image.fromarray(np.full((h, w, 3), 137, dtype=np.uint8))
        else:
            im = synth(rng, w, h, ss == "gray")
        bio = io.BytesIO()
        if ss == "gray":
            im.save(bio, "JPEG", quality=q, progressive=True, **kw)
        else:
            im.save(bio, "JPEG", quality=q, subsampling=ss, progressive=True, **kw)
        data = bio.getvalue()
        assert b"\xff\xc2" in data, name                                   # really SOF2
        out[name + ".jpg"] = np.frombuffer(data, dtype=np.uint8)
        out[name + ".rgb"] = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    np.savez_compressed(os.path.join(HERE, "f15_decode_progressive.npz"), **out)
    print("wrote f15_decode_progressive.npz:", {k: v.shape for k, v in out.items() if k.endswith(".rgb")})


if __name__ == "__main__":
    main()
