"""Row N3, JPEG decode (data/datasets/bases.py:9-41), the parts that run without a GPU: the host marker parser + Huffman
decoder of libeditor_hip.so (editor_jpeg_parse / editor_jpeg_entropy_decode) and the oracle's restatement of libjpeg's
reconstruction (oracle/jpeg_ref.py), together, against PILLOW'S OWN pixels for Pillow-encoded files
(tests/golden/f14_decode.npz, written by tests/golden/capture_jpeg.py) - bit for bit."""
import ctypes
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "f14_decode.npz"))


def host_decode(data):
    from editor_amd import _lib
    cd = _lib.lib().cdll
    buf = np.ascontiguousarray(data, dtype=np.uint8)
    info = np.zeros(16, dtype=np.int32)
    rc = cd.editor_jpeg_parse(ctypes.c_void_p(buf.ctypes.data), buf.size, ctypes.c_void_p(info.ctypes.data))
    if rc:
        return rc, None, None, info
    coef = np.zeros((int(info[8]), 64), dtype=np.int16)
    qt = np.zeros((3, 64), dtype=np.uint16)
    rc = cd.editor_jpeg_entropy_decode(ctypes.c_void_p(buf.ctypes.data), buf.size, ctypes.c_void_p(coef.ctypes.data),
                                       ctypes.c_long(int(info[8])), ctypes.c_void_p(qt.ctypes.data), ctypes.c_void_p(info.ctypes.data))
    return rc, coef, qt, info


def test_host_decoder_plus_oracle_equal_pillow(golden):
    from oracle import jpeg_ref
    names = sorted(k[:-4] for k in golden.files if k.endswith(".rgb"))
    assert len(names) >= 9
    for name in names:
        rc, coef, qt, info = host_decode(golden[name + ".jpg"])
        assert rc == 0, (name, rc)
        want = golden[name + ".rgb"]
        assert (int(info[1]), int(info[0])) == want.shape[:2]
        got = jpeg_ref.reconstruct(coef, qt, info)
        assert np.array_equal(got, want), (name, int(np.abs(got.astype(int) - want.astype(int)).max()))


def test_unsupported_and_corrupt_files_are_refused(golden):
    rc, _, _, _ = host_decode(golden["progressive.jpg"])
    assert rc == 9002                                            # EDITOR_JPEG_UNSUPPORTED: never mis-decoded
    rc, _, _, _ = host_decode(np.frombuffer(b"not a jpeg at all", dtype=np.uint8))
    assert rc == 9001
    data = golden["tiny_420_q50.jpg"].copy()
    rc, _, _, _ = host_decode(data[: data.size // 3])            # truncated inside the headers / scan: error or zeros, no crash
    assert rc in (0, 9001)


def test_matches_live_pillow_when_available(golden):
    """The committed expectation really is what this image's Pillow decodes (skipped where Pillow is absent)."""
    PIL = pytest.importorskip("PIL.Image")
    import io
    for name in ("stitched_420_q75", "odd_422_q95", "gray_q80"):
        live = np.asarray(PIL.open(io.BytesIO(golden[name + ".jpg"].tobytes())).convert("RGB"))
        assert np.array_equal(live, golden[name + ".rgb"])
