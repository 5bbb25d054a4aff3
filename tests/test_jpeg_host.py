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


@pytest.fixture(scope="module")
def golden_prog():
    return np.load(os.path.join(HERE, "golden", "f15_decode_progressive.npz"))


def test_progressive_files_equal_pillow(golden, golden_prog):
    """Round 4: SOF2 files (spectral selection + successive approximation; DC / AC first and refinement scans, end-of-band runs,
    restart intervals) decode to the SAME coefficient planes the sequential path delivers - pixels bit-identical with Pillow,
    which is what the reference's `Image.open(path).convert('RGB')` returns (data/datasets/bases.py:19)."""
    from oracle import jpeg_ref
    names = sorted(k[:-4] for k in golden_prog.files if k.endswith(".rgb"))
    assert len(names) >= 8
    for name in names:
        rc, coef, qt, info = host_decode(golden_prog[name + ".jpg"])
        assert rc == 0, (name, rc)
        want = golden_prog[name + ".rgb"]
        got = jpeg_ref.reconstruct(coef, qt, info)
        assert np.array_equal(got, want), (name, int(np.abs(got.astype(int) - want.astype(int)).max()))
    # the round-3 fixture that used to be the "refused" example decodes now (checked against this image's Pillow when present)
    rc, coef, qt, info = host_decode(golden["progressive.jpg"])
    assert rc == 0
    PIL = pytest.importorskip("PIL.Image")
    import io
    live = np.asarray(PIL.open(io.BytesIO(golden["progressive.jpg"].tobytes())).convert("RGB"))
    assert np.array_equal(jpeg_ref.reconstruct(coef, qt, info), live)


def test_incomplete_progressive_file_is_refused(golden_prog):
    """A progressive file cut after some scans is where libjpeg smooths blocks from their neighbours' DC values: not restated on the
    device, so such a file is an error - never a silently different image.  Whole scans removed: CORRUPT; cut inside a scan: the
    remaining data decodes as zeros (as libjpeg feeds them) and the missing later scans make it CORRUPT as well."""
    data = bytes(golden_prog["prog_444_q92.jpg"])
    sos = [i for i in range(len(data) - 1) if data[i] == 0xFF and data[i + 1] == 0xDA]
    assert len(sos) >= 6                                         # Pillow's default script: ten scans for three components
    for cut in (sos[1], sos[3], sos[-1], sos[2] + 40):
        rc, _, _, _ = host_decode(np.frombuffer(data[:cut] + b"\xFF\xD9", dtype=np.uint8))
        assert rc == 9001, (cut, rc)
    # scan-parameter rules of jdphuff.c: Ss > Se, a DC scan with Se != 0, an interleaved AC scan, Al != Ah - 1
    for patch in ((lambda h: (h[0], 70, h[2])), (lambda h: (0, 5, h[2])), (lambda h: (h[0], h[1], 0x31))):
        bad = bytearray(data)
        a = sos[1]
        ns = bad[a + 4]
        o = a + 5 + 2 * ns
        bad[o], bad[o + 1], bad[o + 2] = patch((bad[o], bad[o + 1], bad[o + 2]))
        rc, _, _, _ = host_decode(np.frombuffer(bytes(bad), dtype=np.uint8))
        assert rc == 9001, rc


def test_unsupported_and_corrupt_files_are_refused(golden):
    # arithmetic coding (SOF9), lossless (SOF3), hierarchical (SOF5), 12-bit samples: EDITOR_JPEG_UNSUPPORTED, never mis-decoded
    base = bytearray(bytes(golden["tiny_420_q50.jpg"]))
    sof = [s_ for s_ in _segments(base) if s_[0] == 0xC0][0]
    for marker in (0xC9, 0xC3, 0xC5, 0xCA):
        bad = bytearray(base)
        bad[sof[1] + 1] = marker
        assert host_decode(np.frombuffer(bytes(bad), dtype=np.uint8))[0] == 9002, hex(marker)
    bad = bytearray(base)
    bad[sof[1] + 4] = 12                                         # sample precision
    assert host_decode(np.frombuffer(bytes(bad), dtype=np.uint8))[0] == 9002
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


# ---- malformed files: the parser sees dataset bytes directly; every one of these must come back as an error code ----
def _segments(data):
    """[(marker, start, end)] of the marker segments before the first SOS (end exclusive, start at the 0xFF)."""
    d = bytes(data)
    pos, out = 2, []
    while pos + 4 <= len(d):
        assert d[pos] == 0xFF
        m = d[pos + 1]
        ln = (d[pos + 2] << 8) | d[pos + 3]
        out.append((m, pos, pos + 2 + ln))
        if m == 0xDA:
            break
        pos += 2 + ln
    return out


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8)


def test_dht_with_too_many_short_codes_is_refused_not_written():
    # 280-byte file of the advisor's report: SOI + DHT whose bits[1] = 255 (a length-1 code space holds two codes)
    for bits1 in (3, 255):
        cnt = bits1
        body = bytes([0x00, bits1] + [0] * 15) + bytes([i & 0xFF for i in range(cnt)])
        seg = bytes([0xFF, 0xC4]) + (len(body) + 2).to_bytes(2, "big") + body
        rc, _, _, _ = host_decode(_u8(b"\xFF\xD8" + seg + b"\xFF\xD9"))
        assert rc == 9001, (bits1, rc)
    # over-subscribed at a longer length: 2 codes of length 1 (full) + 1 of length 2
    body = bytes([0x10, 2, 1] + [0] * 14) + bytes([1, 2, 3])
    seg = bytes([0xFF, 0xC4]) + (len(body) + 2).to_bytes(2, "big") + body
    rc, _, _, _ = host_decode(_u8(b"\xFF\xD8" + seg + b"\xFF\xD9"))
    assert rc == 9001
    # length-9/10 boundary of the lookahead table: 2^9 codes would need 512 values; > 256 values is refused by the count check
    body = bytes([0x00] + [0] * 8 + [255] + [0] * 7) + bytes(255)
    seg = bytes([0xFF, 0xC4]) + (len(body) + 2).to_bytes(2, "big") + body
    rc, _, _, _ = host_decode(_u8(b"\xFF\xD8" + seg + b"\xFF\xD9"))
    assert rc == 9001                                            # no SOF / scan follows: still an error, and no crash


def test_second_frame_header_is_refused(golden):
    """A large SOF + its scan followed by a small SOF before EOI must not size the buffer from the small one."""
    big = bytes(golden["stitched_420_q75.jpg"])
    small = bytes(golden["tiny_420_q50.jpg"])
    sof_small = [s for s in _segments(small) if s[0] == 0xC0][0]
    assert big[-2:] == b"\xFF\xD9"
    crafted = big[:-2] + small[sof_small[1]:sof_small[2]] + b"\xFF\xD9"
    rc, _, _, _ = host_decode(_u8(crafted))
    assert rc == 9001
    # and directly against the entropy decoder with a buffer sized for the SMALL frame (what a headers-only pass would say)
    from editor_amd import _lib
    cd = _lib.lib().cdll
    info = np.zeros(16, dtype=np.int32)
    sbuf = _u8(small)
    assert cd.editor_jpeg_parse(ctypes.c_void_p(sbuf.ctypes.data), sbuf.size, ctypes.c_void_p(info.ctypes.data)) == 0
    nsmall = int(info[8])
    guard = np.full((nsmall + 64, 64), 0x5A5A, dtype=np.int16)
    qt = np.zeros((3, 64), dtype=np.uint16)
    bbuf = _u8(big)
    rc = cd.editor_jpeg_entropy_decode(ctypes.c_void_p(bbuf.ctypes.data), bbuf.size, ctypes.c_void_p(guard.ctypes.data),
                                       ctypes.c_long(nsmall), ctypes.c_void_p(qt.ctypes.data), ctypes.c_void_p(info.ctypes.data))
    assert rc == 9001 and (guard[nsmall:] == 0x5A5A).all()       # nothing written past what the caller offered


def test_truncations_and_oversized_counts_never_crash(golden):
    data = bytes(golden["odd_422_q95.jpg"])
    segs = _segments(data)
    sos_end = segs[-1][2]
    for cut in list(range(2, sos_end + 40, 7)) + [len(data) - 1, len(data) - 2]:
        rc, _, _, _ = host_decode(_u8(data[:cut]))
        assert rc in (0, 9001), (cut, rc)
    # a segment length that runs past the end of the file
    for m, a, b in segs:
        bad = bytearray(data)
        bad[a + 2], bad[a + 3] = 0xFF, 0xFF
        rc, _, _, _ = host_decode(_u8(bad))
        assert rc in (9001, 9002), (hex(m), rc)
    # SOS naming more components than the frame has / an unknown component id / table ids out of range
    m, a, b = segs[-1]
    for off, val in ((4, 4), (5, 99), (6, 0x44)):
        bad = bytearray(data)
        bad[a + off] = val
        rc, _, _, _ = host_decode(_u8(bad))
        assert rc == 9001, (off, rc)
    # DQT with a table id > 3, SOF with zero sampling factors
    dqt = [s for s in segs if s[0] == 0xDB][0]
    bad = bytearray(data); bad[dqt[1] + 4] = 0x07
    assert host_decode(_u8(bad))[0] == 9001
    sof = [s for s in segs if s[0] == 0xC0][0]
    bad = bytearray(data); bad[sof[1] + 11] = 0x00
    assert host_decode(_u8(bad))[0] == 9001


def test_file_whose_scans_miss_a_component_is_refused():
    """Per-component scans truncated after Y: Cb / Cr would be reconstructed from whatever the buffer held."""
    PIL = pytest.importorskip("PIL.Image")
    import io
    rng = np.random.default_rng(5)
    img = PIL.fromarray(rng.integers(0, 256, (32, 48, 3), dtype=np.uint8))
    buf = io.BytesIO()
    img.save(buf, "JPEG", quality=80, subsampling=0)
    data = buf.getvalue()
    segs = _segments(data)
    m, a, b = segs[-1]
    assert data[a + 4] == 3                                      # interleaved 3-component scan as Pillow writes it
    # rewrite the scan header to claim ONE component (Y): the entropy data then decodes as a Y-only scan (garbage, but legal
    # Huffman), and Cb / Cr never arrive
    hdr = bytes([0xFF, 0xDA, 0x00, 0x08, 0x01, data[a + 5], data[a + 6], 0x00, 0x3F, 0x00])
    crafted = data[:a] + hdr + data[b:]
    rc, _, _, _ = host_decode(_u8(crafted))
    assert rc == 9001


def test_jpeg_entry_points_have_declared_argument_types():
    from editor_amd import _lib
    cd = _lib.lib().cdll
    for name in ("editor_jpeg_parse", "editor_jpeg_entropy_decode", "editor_jpeg_planes_bytes"):
        fn = getattr(cd, name)
        assert fn.argtypes is not None and fn.restype is ctypes.c_int, name
    assert cd.editor_jpeg_parse.argtypes[1] is ctypes.c_long     # `long n`: never passed as a C int
