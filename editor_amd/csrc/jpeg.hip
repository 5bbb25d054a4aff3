// Baseline-JPEG decode for the input pipeline: what `Image.open(path).convert('RGB')` + the three
// 256-wide crops of data/datasets/bases.py:9-41 produce for the stitched tri-modal images, split the MI355X way:
//   host   (editor_jpeg_parse / editor_jpeg_entropy_decode): marker parsing and the inherently serial Huffman decode of
//          the scan(s) into quantised DCT coefficient blocks - nothing else;
//   device (editor_jpeg_reconstruct): dequantisation + 8x8 inverse DCT, chroma upsampling, YCbCr -> RGB and the crop split,
//          for a whole batch per launch, writing the uint8 (crop, B, H, cw, 3) tensors editor_resize_u8 consumes.
// The arithmetic restates libjpeg's default decompression path - the one Pillow runs (JDCT_ISLOW, do_fancy_upsampling) -
// integer for integer, so the pixels are BIT-IDENTICAL to Pillow's (tests/golden/f14_decode.npz):
//   jidctint.c  jpeg_idct_islow      13-bit constants, two passes, DESCALE rounding, range-limit table
//   jdsample.c  h2v1 / h2v2 fancy    triangle filter (3/4, 1/4), edge replication at the TRUE (unpadded) plane size
//   jdcolor.c   ycc_rgb_convert      16-bit fixed-point tables, ONE_HALF folded into the Cb->G term
// Supported: 8-bit baseline / extended sequential Huffman (SOF0 / SOF1) and PROGRESSIVE Huffman (SOF2, round 4: spectral
// selection + successive approximation, jdphuff.c - the scans only refine the same quantised coefficient planes, which is all
// the device side consumes), 1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0, interleaved or per-component scans, restart intervals,
// JFIF (YCbCr) and Adobe transform 0 / 1.  A progressive file whose scans do not deliver every coefficient at full precision
// (truncated download) is EDITOR_JPEG_CORRUPT: libjpeg would blend neighbouring blocks there (block smoothing), which the
// device kernels do not restate.  Arithmetic-coded, lossless, 12-bit and 4-component files return EDITOR_JPEG_UNSUPPORTED
// (the caller decides; there is no silent fallback here).
#include "common.h"
#include "../../include/editor_hip.h"
#include <string.h>

namespace {

// ------------------------------------------------------------------------------------------------------------------
// host: markers + Huffman
// ------------------------------------------------------------------------------------------------------------------
const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTab {
    bool present = false;
    uint8_t vals[256];
    int maxcode[18];          // largest code of each length (-1: none)
    int valoff[17];           // vals index of the first code of each length minus that code
    uint16_t look[512];       // 9-bit lookahead: (length << 8) | symbol, 0 = longer code
};

bool build_huff(HuffTab& t, const uint8_t* bits /* [1..16] */, const uint8_t* vals, int nvals)
{
    int code = 0, k = 0;
    memset(t.look, 0, sizeof(t.look));
    memcpy(t.vals, vals, nvals);
    for (int l = 1; l <= 16; ++l) {
        t.valoff[l] = k - code;
        for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
            // (checked BEFORE anything is written: a table with more codes of a length than that length can hold - the
            //  Kraft sum libjpeg's jpeg_make_d_derived_tbl rejects - would index past look[512])
            if (k >= nvals || code >= (1 << l)) return false;
            if (l <= 9) {
                const int first = code << (9 - l);
                for (int f = 0; f < (1 << (9 - l)); ++f) t.look[first + f] = (uint16_t)((l << 8) | vals[k]);
            }
        }
        t.maxcode[l] = bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    t.maxcode[17] = 0x7fffffff;
    t.present = true;
    return true;
}

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint64_t acc = 0; int nbits = 0;
    bool marker = false;          // a marker was met: the segment's data is exhausted (zeros are fed, as libjpeg does)
    void fill() {
        while (nbits <= 56) {
            uint8_t b = 0;
            if (!marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;
                    else { marker = true; b = 0; }
                } else ++p;
            } else {
                marker = true;
            }
            acc = (acc << 8) | b;
            nbits += 8;
        }
    }
    int peek(int n) { if (nbits < n) fill(); return (int)((acc >> (nbits - n)) & ((1u << n) - 1)); }
    void drop(int n) { nbits -= n; }
    int get(int n) { if (n == 0) return 0; const int v = peek(n); drop(n); return v; }
    int get1() { const int v = peek(1); drop(1); return v; }
    void restart() { acc = 0; nbits = 0; marker = false; }
};

inline int huff_decode(BitReader& br, const HuffTab& t)
{
    const int pk = br.peek(9);
    const uint16_t e = t.look[pk];
    if (e) { br.drop(e >> 8); return e & 0xff; }
    int code = br.peek(16);                           // slow path: lengths 10..16
    for (int l = 10; l <= 16; ++l) {
        const int c = code >> (16 - l);
        if (c <= t.maxcode[l]) { br.drop(l); return t.vals[(c + t.valoff[l]) & 0xff]; }
    }
    br.drop(16);
    return -1;                                        // corrupt stream
}
inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Comp { int id, hs, vs, tq, td, ta; int bw, bh; long off; };

struct Jpeg {
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    Comp comp[3];
    uint16_t qt[4][64]; bool qt_ok[4] = {false, false, false, false};
    HuffTab dc[4], ac[4];
    int restart = 0;
    bool progressive = false, unsupported = false, have_sof = false;
    int adobe_transform = -1; bool jfif = false;
    long total_blocks = 0;
    bool covered[3] = {false, false, false};      // components some scan has delivered
    int8_t coef_al[3][64];                        // progressive: successive-approximation bit each coefficient has reached (-1: none yet)
};

inline int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// Parses the markers up to (and including the header of) each SOS.  `coef` NULL: headers only; otherwise `coef` holds
// `coef_blocks` blocks of 64 coefficients and every scan is checked against that size before it writes.
int decode(const uint8_t* d, long n, Jpeg& j, int16_t* coef, long coef_blocks)
{
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return EDITOR_JPEG_CORRUPT;
    long pos = 2;
    bool seen_scan = false;
    while (pos + 4 <= n) {
        if (d[pos] != 0xFF) { ++pos; continue; }
        const int m = d[pos + 1];
        if (m == 0xFF) { ++pos; continue; }
        if (m == 0xD9) break;                                             // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { pos += 2; continue; }
        const long len = rd16(d + pos + 2);
        if (len < 2 || pos + 2 + len > n) return EDITOR_JPEG_CORRUPT;
        const uint8_t* s = d + pos + 4;
        const long sl = len - 2;
        if (m == 0xDB) {                                                  // DQT
            long o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                if (tq > 3) return EDITOR_JPEG_CORRUPT;
                ++o;
                if (o + (pq ? 128 : 64) > sl) return EDITOR_JPEG_CORRUPT;
                for (int i = 0; i < 64; ++i) {
                    j.qt[tq][kZigzag[i]] = pq ? (uint16_t)rd16(s + o + 2 * i) : s[o + i];
                }
                o += pq ? 128 : 64;
                j.qt_ok[tq] = true;
            }
        } else if (m == 0xC4) {                                           // DHT
            long o = 0;
            while (o + 17 <= sl) {
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) return EDITOR_JPEG_CORRUPT;
                uint8_t bits[17]; bits[0] = 0;
                int cnt = 0;
                for (int i = 1; i <= 16; ++i) { bits[i] = s[o + i]; cnt += bits[i]; }
                if (cnt > 256 || o + 17 + cnt > sl) return EDITOR_JPEG_CORRUPT;
                if (!build_huff(tc ? j.ac[th] : j.dc[th], bits, s + o + 17, cnt)) return EDITOR_JPEG_CORRUPT;
                o += 17 + cnt;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            if (m != 0xC0 && m != 0xC1 && m != 0xC2) { j.unsupported = true; return EDITOR_JPEG_UNSUPPORTED; }   // lossless / arithmetic / hierarchical
            j.progressive = m == 0xC2;
            memset(j.coef_al, -1, sizeof(j.coef_al));
            if (j.have_sof) return EDITOR_JPEG_CORRUPT;                   // one frame per file (a second SOF would re-size the block grid under the scans)
            if (sl < 6 || s[0] != 8) { j.unsupported = true; return EDITOR_JPEG_UNSUPPORTED; }
            j.H = rd16(s + 1); j.W = rd16(s + 3); j.ncomp = s[5];
            if ((j.ncomp != 1 && j.ncomp != 3) || j.W <= 0 || j.H <= 0 || sl < 6 + 3 * j.ncomp) return EDITOR_JPEG_UNSUPPORTED;
            for (int c = 0; c < j.ncomp; ++c) {
                Comp& k = j.comp[c];
                k.id = s[6 + 3 * c]; k.hs = s[7 + 3 * c] >> 4; k.vs = s[7 + 3 * c] & 15; k.tq = s[8 + 3 * c];
                if (k.hs < 1 || k.vs < 1 || k.tq > 3) return EDITOR_JPEG_CORRUPT;
                j.hmax = k.hs > j.hmax ? k.hs : j.hmax; j.vmax = k.vs > j.vmax ? k.vs : j.vmax;
            }
            if (j.ncomp == 1) { j.comp[0].hs = j.comp[0].vs = 1; j.hmax = j.vmax = 1; }      // (a lone component is never subsampled)
            else {
                // luma carries the full resolution, both chroma planes share one of 1x1 / (hmax)x(vmax) reductions
                if (j.comp[0].hs != j.hmax || j.comp[0].vs != j.vmax || j.comp[1].hs != 1 || j.comp[1].vs != 1 ||
                    j.comp[2].hs != 1 || j.comp[2].vs != 1 || j.hmax > 2 || j.vmax > 2 || (j.hmax == 1 && j.vmax == 2))
                    return EDITOR_JPEG_UNSUPPORTED;
            }
            j.mcux = (j.W + 8 * j.hmax - 1) / (8 * j.hmax); j.mcuy = (j.H + 8 * j.vmax - 1) / (8 * j.vmax);
            long off = 0;
            for (int c = 0; c < j.ncomp; ++c) {
                Comp& k = j.comp[c];
                k.bw = j.mcux * k.hs; k.bh = j.mcuy * k.vs; k.off = off;
                off += (long)k.bw * k.bh;
            }
            j.total_blocks = off;
            j.have_sof = true;
        } else if (m == 0xDD) {
            if (sl < 2) return EDITOR_JPEG_CORRUPT;
            j.restart = rd16(s);
        } else if (m == 0xE0 && sl >= 5 && !memcmp(s, "JFIF", 5)) {
            j.jfif = true;
        } else if (m == 0xEE && sl >= 12 && !memcmp(s, "Adobe", 5)) {
            j.adobe_transform = s[11];
        } else if (m == 0xDA) {                                           // SOS
            if (!j.have_sof || sl < 1) return EDITOR_JPEG_CORRUPT;
            const int ns = s[0];
            if (ns < 1 || ns > j.ncomp || sl < 1 + 2 * ns + 3) return EDITOR_JPEG_CORRUPT;
            int idx[3];
            for (int i = 0; i < ns; ++i) {
                int c = -1;
                for (int q = 0; q < j.ncomp; ++q) if (j.comp[q].id == s[1 + 2 * i]) c = q;
                if (c < 0) return EDITOR_JPEG_CORRUPT;
                idx[i] = c; j.comp[c].td = s[2 + 2 * i] >> 4; j.comp[c].ta = s[2 + 2 * i] & 15;
                if (j.comp[c].td > 3 || j.comp[c].ta > 3) return EDITOR_JPEG_CORRUPT;
            }
            pos += 2 + len;
            seen_scan = true;
            if (!coef) {                                                  // headers only: skip the entropy-coded segment
                while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] != 0x00 && !(d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7))) ++pos;
                continue;
            }
            if (j.total_blocks > coef_blocks) return EDITOR_JPEG_CORRUPT;  // (the frame this scan belongs to must fit the caller's buffer)
            // scan parameters (B.2.3): spectral selection Ss..Se, successive approximation Ah / Al - sequential scans carry 0,63,0,0
            const uint8_t* sp = s + 1 + 2 * ns;
            const int Ss = sp[0], Se = sp[1], Ah = sp[2] >> 4, Al = sp[2] & 15;
            const bool prog = j.progressive;
            if (prog) {
                // jdphuff.c start_pass_phuff_decoder's validity rules
                if (Ss > Se || Se > 63 || Al > 13 || (Ah != 0 && Al != Ah - 1)) return EDITOR_JPEG_CORRUPT;
                if (Ss == 0 ? Se != 0 : ns != 1) return EDITOR_JPEG_CORRUPT;
            }
            for (int i = 0; i < ns; ++i) {
                const Comp& k = j.comp[idx[i]];
                if (!j.qt_ok[k.tq]) return EDITOR_JPEG_CORRUPT;
                if (!prog) {
                    if (!j.dc[k.td].present || !j.ac[k.ta].present) return EDITOR_JPEG_CORRUPT;
                    if (j.covered[idx[i]]) return EDITOR_JPEG_CORRUPT;      // sequential coding: one scan per component
                    j.covered[idx[i]] = true;
                } else {
                    if (Ss == 0 && Ah == 0 && !j.dc[k.td].present) return EDITOR_JPEG_CORRUPT;
                    if (Ss > 0 && !j.ac[k.ta].present) return EDITOR_JPEG_CORRUPT;
                    // a refinement scan refines what a first scan delivered at exactly the bit above; a first scan delivers
                    // coefficients nobody has delivered yet (libjpeg only warns; such files are not decodable unambiguously)
                    for (int kk = Ss; kk <= Se; ++kk) {
                        if (j.coef_al[idx[i]][kk] != (Ah == 0 ? -1 : Ah)) return EDITOR_JPEG_CORRUPT;
                        j.coef_al[idx[i]][kk] = (int8_t)Al;
                    }
                }
            }
            BitReader br{d + pos, d + n};
            int pred[3] = {0, 0, 0};
            int eobrun = 0;                                               // progressive AC scans: blocks still covered by an end-of-band run
            // MCU geometry of this scan: interleaved -> hs x vs blocks per component per MCU over mcux x mcuy MCUs;
            // a single-component scan -> one block per MCU over the component's own (unpadded) block grid
            int nx = j.mcux, ny = j.mcuy;
            if (ns == 1) {
                const Comp& k = j.comp[idx[0]];
                nx = ((j.W * k.hs + j.hmax - 1) / j.hmax + 7) / 8;
                ny = ((j.H * k.vs + j.vmax - 1) / j.vmax + 7) / 8;
            }
            int left = j.restart, rst = 0;
            for (int my = 0; my < ny; ++my)
                for (int mx = 0; mx < nx; ++mx) {
                    if (j.restart && left == 0) {
                        // byte-align, expect RSTn
                        br.restart();
                        const uint8_t* q = br.p;
                        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
                            if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) break;      // some other marker: give up on resync
                            ++q;
                        }
                        if (q + 1 < br.end && q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7) q += 2;
                        br.p = q;
                        (void)rst; ++rst;
                        pred[0] = pred[1] = pred[2] = 0;
                        eobrun = 0;
                        left = j.restart;
                    }
                    for (int i = 0; i < ns; ++i) {
                        const Comp& k = j.comp[idx[i]];
                        const int bh_ = ns == 1 ? 1 : k.vs, bw_ = ns == 1 ? 1 : k.hs;
                        for (int by = 0; by < bh_; ++by)
                            for (int bx = 0; bx < bw_; ++bx) {
                                const int gx = mx * bw_ + bx, gy = my * bh_ + by;
                                int16_t* blk = coef + (k.off + (long)gy * k.bw + gx) * 64;
                                if (!prog) {
                                    memset(blk, 0, 128);
                                    const int sdc = huff_decode(br, j.dc[k.td]);
                                    if (sdc < 0 || sdc > 11) return EDITOR_JPEG_CORRUPT;
                                    const int diff = sdc ? extend(br.get(sdc), sdc) : 0;
                                    pred[idx[i]] += diff;
                                    blk[0] = (int16_t)pred[idx[i]];
                                    for (int kk = 1; kk < 64;) {
                                        const int rs = huff_decode(br, j.ac[k.ta]);
                                        if (rs < 0) return EDITOR_JPEG_CORRUPT;
                                        const int r = rs >> 4, sz = rs & 15;
                                        if (sz == 0) { if (r == 15) { kk += 16; continue; } break; }
                                        kk += r;
                                        if (kk > 63) return EDITOR_JPEG_CORRUPT;
                                        blk[kZigzag[kk]] = (int16_t)extend(br.get(sz), sz);
                                        ++kk;
                                    }
                                } else if (Ss == 0) {
                                    if (Ah == 0) {                        // decode_mcu_DC_first: the DC difference, scaled by 2^Al
                                        const int sdc = huff_decode(br, j.dc[k.td]);
                                        if (sdc < 0 || sdc > 11) return EDITOR_JPEG_CORRUPT;
                                        const int diff = sdc ? extend(br.get(sdc), sdc) : 0;
                                        pred[idx[i]] += diff;
                                        blk[0] = (int16_t)(pred[idx[i]] * (1 << Al));
                                    } else if (br.get1()) {               // decode_mcu_DC_refine: one more bit of every DC value
                                        blk[0] = (int16_t)(blk[0] | (1 << Al));
                                    }
                                } else if (Ah == 0) {                     // decode_mcu_AC_first
                                    if (eobrun > 0) { --eobrun; continue; }
                                    for (int kk = Ss; kk <= Se; ++kk) {
                                        const int rs = huff_decode(br, j.ac[k.ta]);
                                        if (rs < 0) return EDITOR_JPEG_CORRUPT;
                                        const int r = rs >> 4, sz = rs & 15;
                                        if (sz) {
                                            kk += r;
                                            if (kk > 63) return EDITOR_JPEG_CORRUPT;
                                            blk[kZigzag[kk]] = (int16_t)(extend(br.get(sz), sz) * (1 << Al));
                                        } else if (r == 15) {
                                            kk += 15;                      // ZRL: sixteen zeros
                                        } else {                           // EOBr: this block and the next 2^r + bits - 1 end here
                                            eobrun = (1 << r) - 1;
                                            if (r) eobrun += br.get(r);
                                            break;
                                        }
                                    }
                                } else {                                  // decode_mcu_AC_refine
                                    const int p1 = 1 << Al, m1 = -(1 << Al);
                                    int kk = Ss;
                                    if (eobrun == 0) {
                                        for (; kk <= Se; ++kk) {
                                            const int rs = huff_decode(br, j.ac[k.ta]);
                                            if (rs < 0) return EDITOR_JPEG_CORRUPT;
                                            int r = rs >> 4, sv = rs & 15;
                                            if (sv) {
                                                if (sv != 1) return EDITOR_JPEG_CORRUPT;     // (libjpeg warns and goes on with size 1)
                                                sv = br.get1() ? p1 : m1;
                                            } else if (r != 15) {
                                                eobrun = 1 << r;           // EOBr: the run includes this block, whose remaining
                                                if (r) eobrun += br.get(r);  // non-zero coefficients still get their correction bits
                                                break;
                                            }
                                            // skip r still-zero coefficients, refining every already non-zero one on the way
                                            do {
                                                int16_t* c = blk + kZigzag[kk];
                                                if (*c != 0) {
                                                    if (br.get1() && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
                                                } else if (--r < 0) {
                                                    break;
                                                }
                                                ++kk;
                                            } while (kk <= Se);
                                            if (sv) {
                                                if (kk > 63) return EDITOR_JPEG_CORRUPT;
                                                blk[kZigzag[kk]] = (int16_t)sv;
                                            }
                                        }
                                    }
                                    if (eobrun > 0) {
                                        for (; kk <= Se; ++kk) {
                                            int16_t* c = blk + kZigzag[kk];
                                            if (*c != 0 && br.get1() && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
                                        }
                                        --eobrun;
                                    }
                                }
                            }
                    }
                    if (j.restart) --left;
                }
            // continue the marker scan after the entropy-coded data
            pos = br.p - d;
            while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] != 0x00 && !(d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7))) ++pos;
            continue;
        }
        pos += 2 + len;
    }
    if (!j.have_sof || !seen_scan) return EDITOR_JPEG_CORRUPT;
    if (coef)                                                             // every component delivered by some scan, or the planes
        for (int c = 0; c < j.ncomp; ++c) {                               // of the missing ones would be whatever the buffer held
            if (!j.progressive) { if (!j.covered[c]) return EDITOR_JPEG_CORRUPT; continue; }
            // progressive: every coefficient at full precision (an incomplete file is where libjpeg would smooth blocks)
            for (int kk = 0; kk < 64; ++kk) if (j.coef_al[c][kk] != 0) return EDITOR_JPEG_CORRUPT;
        }
    return 0;
}

void fill_info(const Jpeg& j, int* info)
{
    info[0] = j.W; info[1] = j.H; info[2] = j.ncomp; info[3] = j.hmax; info[4] = j.vmax; info[5] = j.mcux; info[6] = j.mcuy;
    // colour transform of the decompressor's default choice (jdapimin.c default_decompress_parms): 1 = YCbCr -> RGB
    int tr = 0;
    if (j.ncomp == 3) {
        if (j.jfif) tr = 1;
        else if (j.adobe_transform >= 0) tr = j.adobe_transform == 1 ? 1 : 0;
        else tr = !(j.comp[0].id == 'R' && j.comp[1].id == 'G' && j.comp[2].id == 'B');
    }
    info[7] = tr;
    info[8] = (int)j.total_blocks;
    for (int c = 0; c < 3; ++c) { info[9 + c] = c < j.ncomp ? j.comp[c].tq : 0; }
}

// ------------------------------------------------------------------------------------------------------------------
// device: reconstruction
// ------------------------------------------------------------------------------------------------------------------
struct JpegGeom {
    int W, H, ncomp, hmax, vmax, mcux, mcuy, transform;
    int bw[3], bh[3]; long off[3];          // block grids (padded to whole MCUs) and first block of each component
    long blocks_per_image;
    long plane_off[3]; int pw[3], ph[3];    // sample planes (padded): pw = 8 bw, ph = 8 bh; byte offsets inside one image's planes
    long plane_bytes;
};

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
// jpeg_idct_islow's output map: sample_range_limit + CENTERJSAMPLE indexed by (x & RANGE_MASK), RANGE_MASK = 1023
__device__ __forceinline__ uint8_t range_limit_centered(int x)
{
    const int i = x & 1023;
    return (uint8_t)(i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896)));
}

#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172

// one 1-D pass of jidctint.c (CONST_BITS 13): in[8] -> out[8] before the final DESCALE (shift given by the caller)
__device__ __forceinline__ void idct_1d(const int (&in)[8], int (&out)[8], int shift)
{
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * FIX_0_541196100;
    int tmp2 = z1 + z3 * (-FIX_1_847759065);
    int tmp3 = z1 + z2 * FIX_0_765366865;
    z2 = in[0]; z3 = in[4];
    int tmp0 = (z2 + z3) << 13, tmp1 = (z2 - z3) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * FIX_1_175875602;
    tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    out[0] = descale(tmp10 + tmp3, shift); out[7] = descale(tmp10 - tmp3, shift);
    out[1] = descale(tmp11 + tmp2, shift); out[6] = descale(tmp11 - tmp2, shift);
    out[2] = descale(tmp12 + tmp1, shift); out[5] = descale(tmp12 - tmp1, shift);
    out[3] = descale(tmp13 + tmp0, shift); out[4] = descale(tmp13 - tmp0, shift);
}

// one thread per 8x8 block: dequantise, columns (-> scaled by 2^PASS1_BITS), rows, range limit, 8 rows of 8 bytes
__global__ __launch_bounds__(128) void jpeg_idct_kernel(const int16_t* __restrict__ coef, const uint16_t* __restrict__ qt, JpegGeom g, int B,
                                                        uint8_t* __restrict__ planes)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * g.blocks_per_image) return;
    const int img = (int)(t / g.blocks_per_image);
    const long bi = t % g.blocks_per_image;
    const int c = (g.ncomp > 1 && bi >= g.off[1]) ? (bi >= g.off[2] ? 2 : 1) : 0;
    const long lb = bi - g.off[c];
    const int by = (int)(lb / g.bw[c]), bx = (int)(lb % g.bw[c]);
    const int16_t* in = coef + t * 64;
    const uint16_t* q = qt + ((long)img * 3 + c) * 64;
    int ws[8][8];
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int v[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = (int)in[r * 8 + col] * (int)q[r * 8 + col];
        idct_1d(v, o, 13 - 2);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r][col] = o[r];
    }
    uint8_t* dst = planes + (long)img * g.plane_bytes + g.plane_off[c] + ((long)by * 8) * g.pw[c] + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int o[8];
        idct_1d(ws[r], o, 13 + 2 + 3);
        uint2 pk;
        pk.x = range_limit_centered(o[0]) | (range_limit_centered(o[1]) << 8) | (range_limit_centered(o[2]) << 16) | ((uint32_t)range_limit_centered(o[3]) << 24);
        pk.y = range_limit_centered(o[4]) | (range_limit_centered(o[5]) << 8) | (range_limit_centered(o[6]) << 16) | ((uint32_t)range_limit_centered(o[7]) << 24);
        *reinterpret_cast<uint2*>(dst + (long)r * g.pw[c]) = pk;
    }
}

__device__ __forceinline__ int clamp255(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }

// chroma sample at full-resolution pixel (x, y) by jdsample.c's fancy upsampling (or the plain value for 1x1)
__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ p, int pw, int cw, int ch, int hmax, int vmax, int x, int y)
{
    if (hmax == 1) return p[(long)y * pw + x];                          // 4:4:4
    const int cx = x >> 1, right = x & 1;
    if (vmax == 1) {                                                    // h2v1_fancy_upsample
        const uint8_t* r = p + (long)y * pw;
        const int v = r[cx];
        if (right) return cx == cw - 1 ? v : (3 * v + r[cx + 1] + 2) >> 2;
        return cx == 0 ? v : (3 * v + r[cx - 1] + 1) >> 2;
    }
    // h2v2_fancy_upsample: nearer row 3/4, further row 1/4 (context rows replicate at the top / bottom of the TRUE plane)
    const int cy = y >> 1, lower = y & 1;
    int oy = lower ? cy + 1 : cy - 1;
    oy = oy < 0 ? 0 : (oy > ch - 1 ? ch - 1 : oy);
    const uint8_t* r0 = p + (long)cy * pw;
    const uint8_t* r1 = p + (long)oy * pw;
    const int thiscol = 3 * r0[cx] + r1[cx];
    if (right) {
        if (cx == cw - 1) return (thiscol * 4 + 7) >> 4;
        return (thiscol * 3 + (3 * r0[cx + 1] + r1[cx + 1]) + 7) >> 4;
    }
    if (cx == 0) return (thiscol * 4 + 8) >> 4;
    return (thiscol * 3 + (3 * r0[cx - 1] + r1[cx - 1]) + 8) >> 4;
}

// one thread per output pixel: (upsampled) Y, Cb, Cr -> RGB (jdcolor.c ycc_rgb_convert), written into crop x / crop_w of
// out (ncrop, B, H, crop_w, 3); pixels right of the last whole crop are dropped (bases.py:19-21: range(W // 256))
__global__ __launch_bounds__(256) void jpeg_color_kernel(const uint8_t* __restrict__ planes, JpegGeom g, int B, int crop_w, int ncrop,
                                                         uint8_t* __restrict__ out)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int wuse = crop_w * ncrop;
    if (t >= (long)B * g.H * wuse) return;
    const int x = (int)(t % wuse), y = (int)((t / wuse) % g.H), img = (int)(t / ((long)wuse * g.H));
    const uint8_t* pl = planes + (long)img * g.plane_bytes;
    const int yv = pl[g.plane_off[0] + (long)y * g.pw[0] + x];
    int r = yv, gg = yv, b = yv;
    if (g.ncomp == 3) {
        const int cw = (g.W + g.hmax - 1) / g.hmax, ch = (g.H + g.vmax - 1) / g.vmax;      // true downsampled size
        const int cb = chroma_at(pl + g.plane_off[1], g.pw[1], cw, ch, g.hmax, g.vmax, x, y);
        const int cr = chroma_at(pl + g.plane_off[2], g.pw[2], cw, ch, g.hmax, g.vmax, x, y);
        if (g.transform) {
            const int xb = cb - 128, xr = cr - 128;
            r = clamp255(yv + ((91881 * xr + 32768) >> 16));
            b = clamp255(yv + ((116130 * xb + 32768) >> 16));
            gg = clamp255(yv + ((-22554 * xb + 32768 - 46802 * xr) >> 16));
        } else {
            r = yv; gg = cb; b = cr;
        }
    }
    const int crop = x / crop_w, xc = x % crop_w;
    uint8_t* o = out + ((((long)crop * B + img) * g.H + y) * crop_w + xc) * 3;
    o[0] = (uint8_t)r; o[1] = (uint8_t)gg; o[2] = (uint8_t)b;
}

bool make_geom(const int* info, JpegGeom& g)
{
    g.W = info[0]; g.H = info[1]; g.ncomp = info[2]; g.hmax = info[3]; g.vmax = info[4]; g.mcux = info[5]; g.mcuy = info[6];
    g.transform = info[7];
    if ((g.ncomp != 1 && g.ncomp != 3) || g.W <= 0 || g.H <= 0 || g.mcux <= 0 || g.mcuy <= 0) return false;
    long off = 0, poff = 0;
    for (int c = 0; c < 3; ++c) {
        const int hs = c == 0 ? g.hmax : 1, vs = c == 0 ? g.vmax : 1;
        g.bw[c] = c < g.ncomp ? g.mcux * hs : 0; g.bh[c] = c < g.ncomp ? g.mcuy * vs : 0;
        g.off[c] = off; off += (long)g.bw[c] * g.bh[c];
        g.pw[c] = g.bw[c] * 8; g.ph[c] = g.bh[c] * 8;
        g.plane_off[c] = poff; poff += (long)g.pw[c] * g.ph[c];
    }
    if (g.ncomp == 1) { g.off[1] = g.off[2] = off; }
    g.blocks_per_image = off;
    g.plane_bytes = poff;
    return off == info[8];
}

}  // namespace

extern "C" int editor_jpeg_parse(const uint8_t* data, long n, int* info)
{
    if (!data || !info) return EDITOR_JPEG_CORRUPT;
    Jpeg j;
    const int rc = decode(data, n, j, nullptr, 0);
    if (rc) return rc;
    fill_info(j, info);
    return 0;
}

extern "C" int editor_jpeg_entropy_decode(const uint8_t* data, long n, int16_t* coef, long coef_blocks, uint16_t* qt, int* info)
{
    if (!data || !coef || !qt || !info) return EDITOR_JPEG_CORRUPT;
    Jpeg j;
    int rc = decode(data, n, j, nullptr, 0);                  // geometry first: the caller's buffer must hold it
    if (rc) return rc;
    if (j.total_blocks > coef_blocks) return EDITOR_JPEG_CORRUPT;
    // blocks no scan covers (the MCU padding of a per-component scan) are zero, never what the caller's buffer held
    memset(coef, 0, (size_t)j.total_blocks * 128);
    Jpeg k;
    rc = decode(data, n, k, coef, coef_blocks);
    if (rc) return rc;
    fill_info(k, info);
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 64; ++i) qt[c * 64 + i] = c < k.ncomp ? k.qt[k.comp[c].tq][i] : 1;
    return 0;
}

extern "C" int editor_jpeg_planes_bytes(const int* info, long* bytes)
{
    JpegGeom g;
    if (!info || !bytes || !make_geom(info, g)) return EDITOR_JPEG_CORRUPT;
    *bytes = g.plane_bytes;
    return 0;
}

extern "C" int editor_jpeg_reconstruct(const int16_t* coef, const uint16_t* qt, const int* info /* host */, int B, uint8_t* planes,
                                       int crop_w, uint8_t* out, hipStream_t stream)
{
    JpegGeom g;
    if (!coef || !qt || !info || !planes || !out || B < 1 || !make_geom(info, g)) return (int)hipErrorInvalidValue;
    if (crop_w <= 0) crop_w = g.W;
    const int ncrop = g.W / crop_w;
    if (ncrop < 1) return (int)hipErrorInvalidValue;
    const long nb = (long)B * g.blocks_per_image;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((nb + 127) / 128)), dim3(128), 0, stream, coef, qt, g, B, planes);
    EDITOR_LAUNCH_CHECK();
    const long npx = (long)B * g.H * crop_w * ncrop;
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, stream, planes, g, B, crop_w, ncrop, out);
    EDITOR_LAUNCH_CHECK();
    return 0;
}
