"""Row N3, JPEG decode on the device: editor_jpeg_reconstruct (dequantisation + IDCT + chroma upsampling + YCbCr -> RGB +
crop split, editor_amd/csrc/jpeg.hip) fed by the host Huffman decoder, through editor_amd.data.DeviceJpegDecoder, against
Pillow's pixels (tests/golden/f14_decode.npz) - what `Image.open(path).convert('RGB')` and the 256-wide crops of
data/datasets/bases.py:9-41 produce - bit for bit, then on into the device resize (bit-exact with Pillow's resize)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "f14_decode.npz"))


def test_device_decode_equals_pillow(golden):
    from editor_amd.data import DeviceJpegDecoder
    dec = DeviceJpegDecoder(crop_w=0)
    for name in sorted(k[:-4] for k in golden.files if k.endswith(".rgb")):
        out = dec([golden[name + ".jpg"].tobytes()], "cuda")
        want = golden[name + ".rgb"]
        assert tuple(out.shape) == (1, 1) + want.shape
        assert np.array_equal(out[0, 0].cpu().numpy(), want), name


def test_stitched_batch_crops_and_resize(golden):
    """A batch of stitched tri-modal files with MIXED chroma sampling -> (3, B, 128, 256, 3) crops == Pillow's
    img.crop((256 i, 0, 256 (i + 1), 128)); then T.Resize on the device == Pillow's resize of the Pillow-decoded crop."""
    from PIL import Image
    from editor_amd.data import DeviceJpegDecoder, DeviceResize
    names = ["stitched_420_q75", "stitched_444_q90", "stitched_422_q85", "stitched_420_q75"]
    dec = DeviceJpegDecoder(crop_w=256, threads=4)
    crops = dec([golden[n + ".jpg"].tobytes() for n in names], "cuda")
    assert tuple(crops.shape) == (3, 4, 128, 256, 3)
    for b, n in enumerate(names):
        want = golden[n + ".rgb"]
        for i in range(3):
            assert np.array_equal(crops[i, b].cpu().numpy(), want[:, 256 * i:256 * (i + 1)]), (n, i)
    rs = DeviceResize((256, 128), interpolation=3)
    got = rs(crops[1])
    for b, n in enumerate(names):
        ref = Image.fromarray(golden[n + ".rgb"][:, 256:512]).resize((128, 256), resample=3)
        assert np.array_equal(got[b].cpu().numpy(), np.asarray(ref)), n


def test_progressive_files_on_the_device():
    """Round 4: progressive files (tests/golden/f15_decode_progressive.npz) go through the same device reconstruction; a batch
    that MIXES a progressive and a baseline file of the same geometry is one launch group."""
    from editor_amd.data import DeviceJpegDecoder
    g = np.load(os.path.join(HERE, "golden", "f15_decode_progressive.npz"))
    dec = DeviceJpegDecoder(crop_w=0)
    for name in sorted(k[:-4] for k in g.files if k.endswith(".rgb")):
        out = dec([g[name + ".jpg"].tobytes()], "cuda")
        assert np.array_equal(out[0, 0].cpu().numpy(), g[name + ".rgb"]), name
    base = np.load(os.path.join(HERE, "golden", "f14_decode.npz"))
    crops = DeviceJpegDecoder(crop_w=256)([g["prog_stitched_420_q75.jpg"].tobytes(), base["stitched_420_q75.jpg"].tobytes()], "cuda")
    assert tuple(crops.shape) == (3, 2, 128, 256, 3)
    for i in range(3):
        assert np.array_equal(crops[i, 0].cpu().numpy(), g["prog_stitched_420_q75.rgb"][:, 256 * i:256 * (i + 1)])
        assert np.array_equal(crops[i, 1].cpu().numpy(), base["stitched_420_q75.rgb"][:, 256 * i:256 * (i + 1)])


def test_decoder_refuses_what_it_does_not_cover(golden):
    from editor_amd.data import DeviceJpegDecoder
    data = bytearray(golden["tiny_420_q50.jpg"].tobytes())
    i = data.index(b"\xff\xc0")
    data[i + 1] = 0xC9                                            # arithmetic-coded sequential: unsupported, never mis-decoded
    with pytest.raises(ValueError):
        DeviceJpegDecoder()([bytes(data)], "cuda")
    prog = np.load(os.path.join(HERE, "golden", "f15_decode_progressive.npz"))["prog_444_q92.jpg"].tobytes()
    cut = prog.index(b"\xff\xda", prog.index(b"\xff\xda") + 2)   # only the first scan survives: incomplete -> corrupt
    with pytest.raises(ValueError):
        DeviceJpegDecoder()([prog[:cut] + b"\xff\xd9"], "cuda")
