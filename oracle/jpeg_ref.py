"""TEST INFRASTRUCTURE (oracle): numpy restatement of the DEVICE half of the JPEG decode of editor_amd/csrc/jpeg.hip -
dequantisation + inverse DCT + chroma upsampling + colour conversion from quantised DCT coefficients.

The algorithm lives in a third-party dependency of the reference, not in /root/reference: `Image.open(path).convert('RGB')`
(data/datasets/bases.py:19) = Pillow -> libjpeg(-turbo) with its DEFAULT decompression parameters (Pillow 12.2.0 /
libjpeg-turbo API 6.2 in this image).  Restated here from libjpeg's published sources:
    jidctint.c  jpeg_idct_islow          (JDCT_ISLOW, the default dct_method)
    jdsample.c  h2v1_fancy_upsample / h2v2_fancy_upsample  (do_fancy_upsampling = TRUE, the default)
    jdcolor.c   build_ycc_rgb_table / ycc_rgb_convert
    jdmainct.c  context rows replicate the first / last REAL sample row
PINNED: tests/test_jpeg_host.py decodes Pillow-encoded fixtures (tests/golden/f14_decode.npz, written by
tests/golden/capture_jpeg.py) through the host Huffman decoder + this file and compares with Pillow's own pixels, bit
for bit.  Only tests/ may import this module."""
import numpy as np

_C = dict(f0298=2446, f0390=3196, f0541=4433, f0765=6270, f0899=7373, f1175=9633, f1501=12299, f1847=15137, f1961=16069,
          f2053=16819, f2562=20995, f3072=25172)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(v, shift):
    """v: (..., 8) int64 along the last axis -> (..., 8); one pass of jpeg_idct_islow (jidctint.c)."""
    c = _C
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * c["f0541"]
    tmp2 = z1 + z3 * (-c["f1847"])
    tmp3 = z1 + z2 * c["f0765"]
    z2, z3 = v[..., 0], v[..., 4]
    tmp0, tmp1 = (z2 + z3) << 13, (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * c["f1175"]
    tmp0, tmp1, tmp2, tmp3 = tmp0 * c["f0298"], tmp1 * c["f2053"], tmp2 * c["f3072"], tmp3 * c["f1501"]
    z1, z2, z3, z4 = z1 * -c["f0899"], z2 * -c["f2562"], z3 * -c["f1961"], z4 * -c["f0390"]
    z3, z4 = z3 + z5, z4 + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    out = [tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3]
    return np.stack([_descale(o, shift) for o in out], axis=-1)


def _range_limit_centered(x):
    i = x & 1023
    return np.where(i < 128, i + 128, np.where(i < 512, 255, np.where(i < 896, 0, i - 896))).astype(np.uint8)


def idct_planes(coef, qt, info):
    """coef (blocks, 64) int16 natural order, qt (3, 64) uint16, info (editor_jpeg_parse) -> list of uint8 planes (padded)."""
    w, h, ncomp, hmax, vmax, mcux, mcuy = [int(v) for v in info[:7]]
    planes, off = [], 0
    for c in range(ncomp):
        hs, vs = (hmax, vmax) if c == 0 else (1, 1)
        bw, bh = mcux * hs, mcuy * vs
        blk = coef[off:off + bw * bh].astype(np.int64).reshape(bh, bw, 8, 8) * qt[c].astype(np.int64).reshape(8, 8)
        off += bw * bh
        ws = _idct_1d(blk.transpose(0, 1, 3, 2), 13 - 2).transpose(0, 1, 3, 2)        # pass 1: columns
        px = _range_limit_centered(_idct_1d(ws, 13 + 2 + 3))                          # pass 2: rows
        planes.append(px.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8))
    return planes


def _fancy_h2v1(p):
    """(rows, n) -> (rows, 2n): jdsample.c h2v1_fancy_upsample."""
    p = p.astype(np.int64)
    n = p.shape[1]
    out = np.empty((p.shape[0], 2 * n), dtype=np.int64)
    left = np.concatenate([p[:, :1], p[:, :-1]], axis=1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], axis=1)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out


def _fancy_h2v2(p):
    """(m, n) -> (2m, 2n): jdsample.c h2v2_fancy_upsample with jdmainct.c's replicated context rows."""
    p = p.astype(np.int64)
    m, n = p.shape
    above = np.concatenate([p[:1], p[:-1]], axis=0)
    below = np.concatenate([p[1:], p[-1:]], axis=0)
    out = np.empty((2 * m, 2 * n), dtype=np.int64)
    for v, other in ((0, above), (1, below)):
        cs = 3 * p + other                                   # column sums (this row 3/4, the other 1/4)
        last = np.concatenate([cs[:, :1], cs[:, :-1]], axis=1)
        nxt = np.concatenate([cs[:, 1:], cs[:, -1:]], axis=1)
        even = (cs * 3 + last + 8) >> 4
        odd = (cs * 3 + nxt + 7) >> 4
        even[:, 0] = (cs[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (cs[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out


def reconstruct(coef, qt, info):
    """-> (H, W, 3) uint8, what Pillow's Image.open(...).convert('RGB') returns for the same file."""
    w, h, ncomp, hmax, vmax = [int(v) for v in info[:5]]
    transform = int(info[7])
    planes = idct_planes(coef, qt, info)
    y = planes[0][:h, :w].astype(np.int64)
    if ncomp == 1:
        return np.repeat(y[..., None], 3, axis=2).astype(np.uint8)
    cw, ch = (w + hmax - 1) // hmax, (h + vmax - 1) // vmax
    chroma = []
    for c in (1, 2):
        p = planes[c][:ch, :cw]
        if hmax == 2 and vmax == 2:
            p = _fancy_h2v2(p)
        elif hmax == 2:
            p = _fancy_h2v1(p)
        chroma.append(np.asarray(p, dtype=np.int64)[:h, :w])
    cb, cr = chroma
    if not transform:
        return np.stack([y, cb, cr], axis=2).astype(np.uint8)
    xb, xr = cb - 128, cr - 128
    r = y + ((91881 * xr + 32768) >> 16)
    b = y + ((116130 * xb + 32768) >> 16)
    g = y + ((-22554 * xb + 32768 - 46802 * xr) >> 16)
    return np.clip(np.stack([r, g, b], axis=2), 0, 255).astype(np.uint8)
