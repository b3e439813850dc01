// Baseline-JPEG decode for the step BEFORE the path (imcui/hloc/utils/io.py:11-21 `read_image` = cv2.imread,
// imcui/hloc/extract_features.py:120-156): at thousands of images per second and GPU the host's cv2 / PIL decode is the feeder
// bottleneck.  rocJPEG is not in this image, so the decoder is written here, split where the format splits:
//
//   host   imcui_hip_jpeg_info / imcui_hip_jpeg_entropy_decode -- marker parsing and the Huffman bit stream (inherently serial per
//          image, so it runs on host threads: the functions are re-entrant and hold no state) -> quantised DCT coefficients, int16,
//          natural (de-zigzagged) order, [component][block row][block column][64], every component padded to whole MCUs.
//   device imcui_hip_jpeg_reconstruct -- dequantisation + the 8x8 inverse DCT, chroma up-sampling, YCbCr -> RGB (or luma only).
//
// The device arithmetic is libjpeg's DEFAULT decompression path restated integer for integer, because that is what cv2.imread and
// PIL run (both link libjpeg-turbo and leave dct_method = JDCT_ISLOW, do_fancy_upsampling = TRUE):
//   * jidctint.c `jpeg_idct_islow`: two passes of the Loeffler-Ligtenberg-Moschytz butterfly in 32-bit integers, CONST_BITS 13,
//     PASS1_BITS 2, constants FIX(0.298631336) .. FIX(3.072711026), DESCALE rounding, range limit to [0, 255] about 128;
//   * jdsample.c `h2v1_fancy_upsample` (3/4, 1/4 with biases 1, 2), `h2v2_fancy_upsample` (9/16, 3/16, 3/16, 1/16 with biases 8, 7),
//     edge samples replicated -- over the component's REAL size ceil(W h_c / h_max) x ceil(H v_c / v_max), not the MCU padding;
//   * jdcolor.c `ycc_rgb_convert`: R = y + ((91881 cr' + 32768) >> 16), B = y + ((116130 cb' + 32768) >> 16),
//     G = y + ((-22554 cb' - 46802 cr' + 32768) >> 16), cb' = cb - 128, cr' = cr - 128, range limited;
//   * gray output of a YCbCr file (cv2.IMREAD_GRAYSCALE, PIL's draft('L')) = the luma plane as decoded: chroma is not touched.
// Results are BIT-EXACT against PIL on every JPEG of the reference repository (tests/test_jpeg_cpu.py pins the restatement
// oracle/jpeg.py to PIL; tests/test_gpu_jpeg.py the kernels).  Supported: baseline / extended sequential Huffman (SOF0, SOF1), 8 bit,
// 1 or 3 components, sampling 4:4:4, 4:2:2, 4:2:0, restart intervals, interleaved and non-interleaved scans.  Anything else
// (arithmetic coding, CMYK, 12 bit, 4:4:0 and other sampling factors; progressive files are taken since round 5) is reported as IMCUI_ERR_UNSUPPORTED and the caller keeps
// its host decoder for that file.  EXIF orientation is reported in the info record and applied by imcui_hip_orient_u8 (the Python layer
// does that, as cv2.imread does).
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "common.h"
#include "imcui_hip.h"

// info record (plain ints, include/imcui_hip.h): 0 width, 1 height, 2 ncomp, 3 hmax, 4 vmax, 5 mcus_x, 6 mcus_y, 7 restart interval,
// 8 EXIF orientation (1 = upright / absent), 9 + 4 c .. : h_c, v_c, quant table, reserved for component c (c < 3)
#define JI_W 0
#define JI_H 1
#define JI_NC 2
#define JI_HMAX 3
#define JI_VMAX 4
#define JI_MX 5
#define JI_MY 6
#define JI_RST 7
#define JI_ORI 8
#define JI_COMP 9
#define JI_INTS 24

namespace {

struct HuffTable {
    bool present = false;
    unsigned char bits[17];
    unsigned char vals[256];
    // canonical decoding tables (ITU T.81 F.2.2.3)
    int mincode[17], maxcode[18], valptr[17];
    // 9-bit look-ahead: (length << 8) | symbol, 0 = longer than 9 bits
    unsigned short look[512];
    // false: the code lengths do not form a prefix code (over-subscribed: more codes of some length than the code space has left).
    // libjpeg refuses such a table in jpeg_make_d_derived_tbl; without the test `code << (9 - l)` below indexes far past look[].
    bool build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            if (code > (1 << l)) return false;
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        memset(look, 0, sizeof look);
        code = 0;
        k = 0;
        for (int l = 1; l <= 9; ++l) {
            for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                const int lo = code << (9 - l), n = 1 << (9 - l);
                for (int j = 0; j < n; ++j) look[lo + j] = (unsigned short)((l << 8) | vals[k]);
            }
            code <<= 1;
        }
        return true;
    }
};

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int bw = 0, bh = 0;      // blocks per row / rows of the MCU-padded plane
    int rbw = 0, rbh = 0;    // blocks that cover the component's real size (non-interleaved scans walk these)
    size_t off = 0;          // first coefficient of the plane inside ONE buffer holding all components
    short* base = nullptr;   // where the plane's coefficients go (set by the entry points)
};

struct Jpeg {
    int W = 0, H = 0, nc = 0, hmax = 1, vmax = 1, mx = 0, my = 0, rst = 0, orientation = 1, precision = 8;
    bool progressive = false, arithmetic = false, have_sof = false;
    bool jfif = false, adobe = false;  // APP0 'JFIF' / APP14 'Adobe' seen (libjpeg's colour-space rule, finish_geometry)
    int adobe_transform = 0;
    Comp comp[4];
    unsigned short qt[4][64];
    bool have_qt[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
};

const unsigned char ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                  41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                  30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline int be16(const unsigned char* p) { return (p[0] << 8) | p[1]; }

// EXIF orientation (tag 0x0112) of an APP1 segment, 1 when absent / malformed
int exif_orientation(const unsigned char* p, size_t n) {
    if (n < 14 || memcmp(p, "Exif\0\0", 6) != 0) return 1;
    const unsigned char* t = p + 6;
    const size_t tn = n - 6;
    const bool le = t[0] == 'I' && t[1] == 'I';
    if (!le && !(t[0] == 'M' && t[1] == 'M')) return 1;
    auto u16 = [&](size_t o) -> unsigned { return le ? (unsigned)(t[o] | (t[o + 1] << 8)) : (unsigned)((t[o] << 8) | t[o + 1]); };
    auto u32 = [&](size_t o) -> unsigned {
        return le ? (unsigned)(t[o] | (t[o + 1] << 8) | (t[o + 2] << 16) | ((unsigned)t[o + 3] << 24))
                  : (unsigned)(((unsigned)t[o] << 24) | (t[o + 1] << 16) | (t[o + 2] << 8) | t[o + 3]);
    };
    if (tn < 8 || u16(2) != 42) return 1;
    size_t ifd = u32(4);
    if (ifd + 2 > tn) return 1;
    const unsigned cnt = u16(ifd);
    for (unsigned i = 0; i < cnt; ++i) {
        const size_t e = ifd + 2 + 12 * (size_t)i;
        if (e + 12 > tn) return 1;
        if (u16(e) == 0x0112) {
            const unsigned v = u16(e + 8);
            return (v >= 1 && v <= 8) ? (int)v : 1;
        }
    }
    return 1;
}

// one DHT segment body (possibly several tables)
int parse_dht(const unsigned char* p, int pl, Jpeg& j) {
    int o = 0;
    while (o < pl) {
        if (o + 17 > pl) return IMCUI_ERR_ARG;
        const int tc = p[o] >> 4, th = p[o] & 15;
        if (tc > 1 || th > 3) return IMCUI_ERR_ARG;
        HuffTable& t = tc ? j.ac[th] : j.dc[th];
        int cnt = 0;
        t.bits[0] = 0;
        for (int l = 1; l <= 16; ++l) {
            t.bits[l] = p[o + l];
            cnt += t.bits[l];
        }
        if (cnt > 256 || o + 17 + cnt > pl) return IMCUI_ERR_ARG;
        memcpy(t.vals, p + o + 17, cnt);
        t.present = false;
        if (!t.build()) return IMCUI_ERR_ARG;
        t.present = true;
        o += 17 + cnt;
    }
    return IMCUI_OK;
}
// one DQT segment body
int parse_dqt(const unsigned char* p, int pl, Jpeg& j) {
    int o = 0;
    while (o < pl) {
        const int pq = p[o] >> 4, tq = p[o] & 15;
        if (tq > 3 || o + 1 + (pq ? 128 : 64) > pl) return IMCUI_ERR_ARG;
        for (int k = 0; k < 64; ++k) j.qt[tq][ZIGZAG[k]] = pq ? (unsigned short)be16(p + o + 1 + 2 * k) : p[o + 1 + k];
        j.have_qt[tq] = true;
        o += 1 + (pq ? 128 : 64);
    }
    return IMCUI_OK;
}

// parse every segment up to (not including) the first SOS; returns the offset of that SOS marker or a negative status
long parse_headers(const unsigned char* d, size_t n, Jpeg& j) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return IMCUI_ERR_ARG;
    size_t i = 2;
    while (i + 4 <= n) {
        if (d[i] != 0xFF) return IMCUI_ERR_ARG;
        while (i < n && d[i] == 0xFF) ++i;  // fill bytes
        if (i >= n) return IMCUI_ERR_ARG;
        const int m = d[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return IMCUI_ERR_ARG;  // EOI before any scan
        if (i + 2 > n) return IMCUI_ERR_ARG;
        const int L = be16(d + i);
        if (L < 2 || i + L > n) return IMCUI_ERR_ARG;
        const unsigned char* p = d + i + 2;
        const int pl = L - 2;
        if (m == 0xDA) return (long)(i - 2);
        if (m == 0xC0 || m == 0xC1 || m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xC7) || (m >= 0xC9 && m <= 0xCB) || (m >= 0xCD && m <= 0xCF)) {
            if (pl < 6) return IMCUI_ERR_ARG;
            j.progressive = (m == 0xC2 || m == 0xC6 || m == 0xCA || m == 0xCE);
            j.arithmetic = m >= 0xC9;
            if (m != 0xC0 && m != 0xC1 && m != 0xC2) return IMCUI_ERR_UNSUPPORTED;  // lossless / differential / arithmetic coding (SOF2 = progressive Huffman: round 5)
            j.precision = p[0];
            j.H = be16(p + 1);
            j.W = be16(p + 3);
            j.nc = p[5];
            if (j.nc < 1 || j.nc > 4 || pl < 6 + 3 * j.nc) return IMCUI_ERR_ARG;
            for (int c = 0; c < j.nc; ++c) {
                j.comp[c].id = p[6 + 3 * c];
                j.comp[c].h = p[7 + 3 * c] >> 4;
                j.comp[c].v = p[7 + 3 * c] & 15;
                j.comp[c].tq = p[8 + 3 * c] & 3;
            }
            j.have_sof = true;
        } else if (m == 0xDB) {
            const int rc = parse_dqt(p, pl, j);
            if (rc != IMCUI_OK) return rc;
        } else if (m == 0xC4) {
            const int rc = parse_dht(p, pl, j);
            if (rc != IMCUI_OK) return rc;
        } else if (m == 0xDD) {
            if (pl < 2) return IMCUI_ERR_ARG;
            j.rst = be16(p);
        } else if (m == 0xE1) {
            const int ori = exif_orientation(p, pl);
            if (ori != 1) j.orientation = ori;
        } else if (m == 0xE0) {
            if (pl >= 5 && memcmp(p, "JFIF\0", 5) == 0) j.jfif = true;
        } else if (m == 0xEE) {
            if (pl >= 12 && memcmp(p, "Adobe", 5) == 0) {
                j.adobe = true;
                j.adobe_transform = p[11];
            }
        }
        i += L;
    }
    return IMCUI_ERR_ARG;
}

int finish_geometry(Jpeg& j) {
    if (!j.have_sof || j.W <= 0 || j.H <= 0) return IMCUI_ERR_ARG;
    if (j.precision != 8 || (j.nc != 1 && j.nc != 3)) return IMCUI_ERR_UNSUPPORTED;
    if (j.nc == 3 && !j.jfif) {
        // libjpeg's default_decompress_parms: JFIF -> YCbCr; else an Adobe marker decides (transform 0 = RGB stored as is); else the
        // component ids ('R','G','B' = RGB).  Only YCbCr frames are reconstructed here; the others go back to the host reader.
        if (j.adobe) {
            if (j.adobe_transform != 1) return IMCUI_ERR_UNSUPPORTED;
        } else if (j.comp[0].id == 'R' && j.comp[1].id == 'G' && j.comp[2].id == 'B') {
            return IMCUI_ERR_UNSUPPORTED;
        }
    }
    j.hmax = j.vmax = 1;
    for (int c = 0; c < j.nc; ++c) {
        if (j.comp[c].h < 1 || j.comp[c].h > 4 || j.comp[c].v < 1 || j.comp[c].v > 4) return IMCUI_ERR_ARG;
        j.hmax = j.comp[c].h > j.hmax ? j.comp[c].h : j.hmax;
        j.vmax = j.comp[c].v > j.vmax ? j.comp[c].v : j.vmax;
    }
    if (j.nc == 1) {  // a single-component frame is never interleaved: its sampling factors do not matter
        j.comp[0].h = j.comp[0].v = 1;
        j.hmax = j.vmax = 1;
    } else {
        // luma at full resolution, both chroma planes at one common reduction of 1 or 2 per axis (4:4:4, 4:2:2, 4:2:0, 4:4:0)
        if (j.comp[0].h != j.hmax || j.comp[0].v != j.vmax || j.comp[1].h != j.comp[2].h || j.comp[1].v != j.comp[2].v) return IMCUI_ERR_UNSUPPORTED;
        if (j.comp[1].h != 1 || j.comp[1].v != 1 || j.hmax > 2 || j.vmax > 2) return IMCUI_ERR_UNSUPPORTED;
        if (j.hmax == 1 && j.vmax == 2) return IMCUI_ERR_UNSUPPORTED;  // 4:4:0 (h1v2 up-sampling): no encoder here to pin it against
    }
    j.mx = (j.W + 8 * j.hmax - 1) / (8 * j.hmax);
    j.my = (j.H + 8 * j.vmax - 1) / (8 * j.vmax);
    size_t off = 0;
    for (int c = 0; c < j.nc; ++c) {
        Comp& k = j.comp[c];
        k.bw = j.mx * k.h;
        k.bh = j.my * k.v;
        const int cw = (j.W * k.h + j.hmax - 1) / j.hmax, chh = (j.H * k.v + j.vmax - 1) / j.vmax;
        k.rbw = (cw + 7) / 8;
        k.rbh = (chh + 7) / 8;
        k.off = off;
        off += (size_t)k.bw * k.bh * 64;
    }
    return IMCUI_OK;
}

void fill_info(const Jpeg& j, int* info) {
    memset(info, 0, JI_INTS * sizeof(int));
    info[JI_W] = j.W;
    info[JI_H] = j.H;
    info[JI_NC] = j.nc;
    info[JI_HMAX] = j.hmax;
    info[JI_VMAX] = j.vmax;
    info[JI_MX] = j.mx;
    info[JI_MY] = j.my;
    info[JI_RST] = j.rst;
    info[JI_ORI] = j.orientation;
    for (int c = 0; c < j.nc && c < 3; ++c) {
        info[JI_COMP + 4 * c + 0] = j.comp[c].h;
        info[JI_COMP + 4 * c + 1] = j.comp[c].v;
        info[JI_COMP + 4 * c + 2] = j.comp[c].tq;
    }
}

// ---- the entropy-coded segment
struct BitReader {
    const unsigned char* d;
    size_t n, pos;
    unsigned long long acc = 0;  // bits are consumed from the top
    int cnt = 0;
    int marker = 0;  // a marker met while refilling (0xFFxx with xx != 0): no more data bits exist
    BitReader(const unsigned char* d_, size_t n_, size_t p) : d(d_), n(n_), pos(p) {}
    inline void fill() {
        while (cnt <= 56) {
            int b = 0;
            if (!marker && pos < n) {
                b = d[pos];
                if (b == 0xFF) {
                    while (pos + 2 < n && d[pos + 1] == 0xFF) ++pos;  // FF FF ..: fill bytes in front of a marker / a stuffed FF (T.81 B.1.1.2)
                    const int nx = pos + 1 < n ? d[pos + 1] : 0xD9;
                    if (nx == 0) {
                        pos += 2;
                    } else {
                        marker = nx;  // leave the position ON the marker; zeros are fed from here on (libjpeg does the same)
                        b = 0;
                    }
                } else {
                    ++pos;
                }
            }
            acc |= (unsigned long long)b << (56 - cnt);
            cnt += 8;
        }
    }
    inline int peek(int k) { return (int)(acc >> (64 - k)); }
    inline void skip(int k) {
        acc <<= k;
        cnt -= k;
    }
    inline int get(int k) {
        if (k == 0) return 0;
        if (cnt < k) fill();
        const int v = peek(k);
        skip(k);
        return v;
    }
    inline void reset() {
        acc = 0;
        cnt = 0;
        marker = 0;
    }
};

inline int huff_decode(BitReader& br, const HuffTable& t) {
    if (br.cnt < 16) br.fill();
    const unsigned short e = t.look[br.peek(9)];
    if (e) {
        br.skip(e >> 8);
        return e & 255;
    }
    int code = br.peek(10), l = 10;
    // (codes of 10 .. 16 bits: the canonical walk of T.81 F.16)
    for (;; ++l) {
        if (l > 16) return -1;
        if (t.maxcode[l] >= 0 && code <= t.maxcode[l] && code >= t.mincode[l]) break;
        code = br.peek(l + 1);
    }
    br.skip(l);
    return t.vals[t.valptr[l] + code - t.mincode[l]];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

int decode_block(BitReader& br, const HuffTable& dct, const HuffTable& act, int& pred, short* blk) {
    int s = huff_decode(br, dct);
    if (s < 0 || s > 11) return IMCUI_ERR_ARG;
    int diff = 0;
    if (s) diff = extend(br.get(s), s);
    pred += diff;
    blk[0] = (short)pred;
    for (int k = 1; k < 64;) {
        const int rs = huff_decode(br, act);
        if (rs < 0) return IMCUI_ERR_ARG;
        const int r = rs >> 4, sz = rs & 15;
        if (sz == 0) {
            if (r != 15) break;  // EOB
            k += 16;
            continue;
        }
        k += r;
        if (k > 63) return IMCUI_ERR_ARG;
        blk[ZIGZAG[k]] = (short)extend(br.get(sz), sz);
        ++k;
    }
    return IMCUI_OK;
}

// ---- progressive frames (SOF2; ITU T.81 Annex G, the procedure of libjpeg's jdphuff.c): a scan codes a band Ss .. Se of the zig-zag
// sequence at bit position Al; DC scans (Ss = 0) may interleave components, AC scans code one component block by block with end-of-band
// runs; a refinement scan (Ah > 0) adds one bit to what earlier scans left.  The coefficient planes accumulate over the scans; once
// the last scan is in they hold exactly what a sequential file would have, and the device reconstruction is the same.
// `first` = the offset of the entropy-coded data; returns the offset of the next marker or < 0.
long decode_scan_progressive(const unsigned char* d, size_t n, size_t first, Jpeg& j, const int* sc, int ns, int Ss, int Se, int Ah, int Al) {
    if (Ss > Se || Se > 63 || Al > 13 || Ah > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1)) return IMCUI_ERR_ARG;
    if (Ah != 0 && Ah != Al + 1) return IMCUI_ERR_ARG;  // successive approximation adds one bit per pass
    for (int s = 0; s < ns; ++s) {
        const Comp& k = j.comp[sc[s]];
        if (Ss == 0 ? (Ah == 0 && !j.dc[k.td].present) : !j.ac[k.ta].present) return IMCUI_ERR_ARG;
    }
    BitReader br(d, n, first);
    int pred[4] = {0, 0, 0, 0};
    unsigned eobrun = 0;
    const bool inter = ns > 1;
    const int units_x = inter ? j.mx : j.comp[sc[0]].rbw, units_y = inter ? j.my : j.comp[sc[0]].rbh;
    const long total = (long)units_x * units_y;
    int rst_left = j.rst, next_rst = 0;
    const int p1 = 1 << Al, m1 = -(1 << Al);
    for (long u = 0; u < total; ++u) {
        if (j.rst && rst_left == 0) {
            br.reset();
            size_t q = br.pos;
            while (q + 1 < n && !(d[q] == 0xFF && d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7)) {
                if (d[q] == 0xFF && d[q + 1] != 0 && d[q + 1] != 0xFF) return IMCUI_ERR_ARG;
                ++q;
            }
            if (q + 1 >= n || d[q + 1] != 0xD0 + next_rst) return IMCUI_ERR_ARG;
            br.pos = q + 2;
            next_rst = (next_rst + 1) & 7;
            rst_left = j.rst;
            pred[0] = pred[1] = pred[2] = pred[3] = 0;
            eobrun = 0;
        }
        const int ux = (int)(u % units_x), uy = (int)(u / units_x);
        if (Ss == 0) {
            // ---- DC: first pass = the sequential DC difference, shifted; refinement = one raw bit per block
            for (int s = 0; s < ns; ++s) {
                const Comp& k = j.comp[sc[s]];
                const int nby = inter ? k.v : 1, nbx = inter ? k.h : 1;
                for (int by = 0; by < nby; ++by)
                    for (int bx = 0; bx < nbx; ++bx) {
                        short* blk = inter ? k.base + ((size_t)(uy * k.v + by) * k.bw + (ux * k.h + bx)) * 64 : k.base + ((size_t)uy * k.bw + ux) * 64;
                        if (Ah == 0) {
                            const int t = huff_decode(br, j.dc[k.td]);
                            if (t < 0 || t > 15) return IMCUI_ERR_ARG;
                            if (t) pred[sc[s]] += extend(br.get(t), t);
                            blk[0] = (short)(pred[sc[s]] * p1);
                        } else if (br.get(1)) {
                            blk[0] = (short)(blk[0] | p1);
                        }
                    }
            }
        } else {
            const Comp& kc = j.comp[sc[0]];
            short* blk = kc.base + ((size_t)uy * kc.bw + ux) * 64;
            const HuffTable& act = j.ac[kc.ta];
            if (Ah == 0) {
                // ---- AC first pass: (run, size) symbols inside the band; size 0 with run < 15 opens an end-of-band run
                if (eobrun > 0) {
                    --eobrun;
                } else {
                    for (int k = Ss; k <= Se; ++k) {
                        const int rs = huff_decode(br, act);
                        if (rs < 0) return IMCUI_ERR_ARG;
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz) {
                            k += r;
                            if (k > 63) return IMCUI_ERR_ARG;
                            blk[ZIGZAG[k]] = (short)(extend(br.get(sz), sz) * p1);
                        } else if (r == 15) {
                            k += 15;
                        } else {
                            eobrun = 1u << r;
                            if (r) eobrun += (unsigned)br.get(r);
                            --eobrun;  // this block is the first of the run
                            break;
                        }
                    }
                }
            } else {
                // ---- AC refinement: every coefficient that is already non-zero receives one correction bit as the walk passes it; a new
                // coefficient (+-1 << Al) lands after `r` still-zero positions
                int k = Ss;
                if (eobrun == 0) {
                    for (; k <= Se; ++k) {
                        const int rs = huff_decode(br, act);
                        if (rs < 0) return IMCUI_ERR_ARG;
                        int r = rs >> 4;
                        const int sz = rs & 15;
                        int val = 0;
                        if (sz) {
                            val = br.get(1) ? p1 : m1;  // (size must be 1 here; libjpeg warns and carries on for anything else)
                        } else if (r != 15) {
                            eobrun = 1u << r;
                            if (r) eobrun += (unsigned)br.get(r);
                            break;  // the rest of this block is handled by the end-of-band branch below
                        }
                        do {
                            short* cp = blk + ZIGZAG[k];
                            if (*cp != 0) {
                                if (br.get(1) && (*cp & p1) == 0) *cp = (short)(*cp + (*cp >= 0 ? p1 : m1));
                            } else if (--r < 0) {
                                break;
                            }
                            ++k;
                        } while (k <= Se);
                        if (val && k <= 63) blk[ZIGZAG[k]] = (short)val;
                    }
                }
                if (eobrun > 0) {
                    for (; k <= Se; ++k) {
                        short* cp = blk + ZIGZAG[k];
                        if (*cp != 0 && br.get(1) && (*cp & p1) == 0) *cp = (short)(*cp + (*cp >= 0 ? p1 : m1));
                    }
                    --eobrun;
                }
            }
        }
        if (j.rst) --rst_left;
    }
    size_t q = br.pos;
    while (q + 1 < n && !(d[q] == 0xFF && d[q + 1] != 0 && d[q + 1] != 0xFF && !(d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7))) ++q;
    return (long)q;
}

// one scan starting at the SOS marker at `pos`; returns the offset of the next marker (after the entropy-coded data) or < 0
long decode_scan(const unsigned char* d, size_t n, size_t pos, Jpeg& j) {
    if (pos + 4 > n) return IMCUI_ERR_ARG;
    const int L = be16(d + pos + 2);
    if (L < 6 || pos + 2 + L > n) return IMCUI_ERR_ARG;
    const unsigned char* p = d + pos + 4;
    const int ns = p[0];
    if (ns < 1 || ns > j.nc || L != 6 + 2 * ns) return IMCUI_ERR_ARG;
    int sc[4];
    for (int s = 0; s < ns; ++s) {
        int c = 0;
        while (c < j.nc && j.comp[c].id != p[1 + 2 * s]) ++c;
        if (c == j.nc) return IMCUI_ERR_ARG;
        sc[s] = c;
        j.comp[c].td = p[2 + 2 * s] >> 4;
        j.comp[c].ta = p[2 + 2 * s] & 15;
        if (j.comp[c].td > 3 || j.comp[c].ta > 3 || !j.have_qt[j.comp[c].tq]) return IMCUI_ERR_ARG;
        if (!j.progressive && (!j.dc[j.comp[c].td].present || !j.ac[j.comp[c].ta].present)) return IMCUI_ERR_ARG;
    }
    const unsigned char* tail = p + 1 + 2 * ns;
    if (j.progressive) return decode_scan_progressive(d, n, pos + 2 + L, j, sc, ns, tail[0], tail[1], tail[2] >> 4, tail[2] & 15);
    if (tail[0] != 0 || tail[1] != 63 || tail[2] != 0) return IMCUI_ERR_ARG;  // a sequential frame codes whole blocks
    BitReader br(d, n, pos + 2 + L);
    int pred[4] = {0, 0, 0, 0};
    const bool inter = ns > 1;
    const int units_x = inter ? j.mx : j.comp[sc[0]].rbw, units_y = inter ? j.my : j.comp[sc[0]].rbh;
    const long total = (long)units_x * units_y;
    int rst_left = j.rst, next_rst = 0;
    for (long u = 0; u < total; ++u) {
        if (j.rst && rst_left == 0) {
            // a restart marker must follow: drop the partial byte, find RSTn, reset the predictors
            br.reset();
            size_t q = br.pos;
            while (q + 1 < n && !(d[q] == 0xFF && d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7)) {
                if (d[q] == 0xFF && d[q + 1] != 0 && d[q + 1] != 0xFF) return IMCUI_ERR_ARG;  // another marker: truncated scan
                ++q;
            }
            if (q + 1 >= n || d[q + 1] != 0xD0 + next_rst) return IMCUI_ERR_ARG;
            br.pos = q + 2;
            next_rst = (next_rst + 1) & 7;
            rst_left = j.rst;
            pred[0] = pred[1] = pred[2] = pred[3] = 0;
        }
        const int ux = (int)(u % units_x), uy = (int)(u / units_x);
        if (inter) {
            for (int s = 0; s < ns; ++s) {
                const Comp& k = j.comp[sc[s]];
                for (int by = 0; by < k.v; ++by)
                    for (int bx = 0; bx < k.h; ++bx) {
                        short* blk = k.base + ((size_t)(uy * k.v + by) * k.bw + (ux * k.h + bx)) * 64;
                        const int rc = decode_block(br, j.dc[k.td], j.ac[k.ta], pred[sc[s]], blk);
                        if (rc != IMCUI_OK) return rc;
                    }
            }
        } else {
            const Comp& k = j.comp[sc[0]];
            short* blk = k.base + ((size_t)uy * k.bw + ux) * 64;
            const int rc = decode_block(br, j.dc[k.td], j.ac[k.ta], pred[sc[0]], blk);
            if (rc != IMCUI_OK) return rc;
        }
        if (j.rst) --rst_left;
    }
    // position of the next marker
    size_t q = br.marker ? br.pos : br.pos;
    while (q + 1 < n && !(d[q] == 0xFF && d[q + 1] != 0 && d[q + 1] != 0xFF && !(d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7))) ++q;
    return (long)q;
}

int parse_all(const unsigned char* d, size_t n, Jpeg& j, long* sos) {
    const long s = parse_headers(d, n, j);
    if (s < 0) return (int)s;
    const int rc = finish_geometry(j);
    if (rc != IMCUI_OK) return rc;
    *sos = s;
    return IMCUI_OK;
}

}  // namespace

extern "C" int imcui_hip_jpeg_info(const unsigned char* data, size_t n, int* info) {
    if (!data || !info) return IMCUI_ERR_ARG;
    Jpeg j;
    long sos;
    const int rc = parse_all(data, n, j, &sos);
    if (rc != IMCUI_OK) return rc;
    fill_info(j, info);
    return IMCUI_OK;
}

extern "C" size_t imcui_hip_jpeg_coef_count(const int* info) {
    if (!info) return 0;
    size_t t = 0;
    for (int c = 0; c < info[JI_NC] && c < 3; ++c) t += (size_t)info[JI_MX] * info[JI_COMP + 4 * c] * info[JI_MY] * info[JI_COMP + 4 * c + 1] * 64;
    return t;
}

// planes[c] (c < components): destination of component c's coefficients (bw * bh * 64 int16, zeroed here); planes == nullptr: one
// buffer `coef` with the components one behind the other
static int entropy_decode_impl(const unsigned char* data, size_t n, short* coef, short* const* planes, unsigned short* qt) {
    Jpeg j;
    long pos;
    int rc = parse_all(data, n, j, &pos);
    if (rc != IMCUI_OK) return rc;
    for (int c = 0; c < j.nc; ++c) {
        j.comp[c].base = planes ? planes[c] : coef + j.comp[c].off;
        if (!j.comp[c].base) return IMCUI_ERR_ARG;
        memset(j.comp[c].base, 0, (size_t)j.comp[c].bw * j.comp[c].bh * 64 * sizeof(short));
    }
    int done = 0;
    bool seen[4] = {false, false, false, false};
    // sequential: until every component has been coded; progressive: every scan up to EOI (each adds a band or a bit)
    while (done < j.nc || j.progressive) {
        if ((size_t)pos + 4 > n || data[pos] != 0xFF) {
            if (j.progressive && (size_t)pos + 2 <= n && data[pos] == 0xFF && data[pos + 1] == 0xD9 && done == j.nc) break;
            return IMCUI_ERR_ARG;
        }
        const int m = data[pos + 1];
        if (m == 0xDA) {
            const int SL = be16(data + pos + 2);
            if (SL < 6 || (size_t)pos + 2 + SL > n) return IMCUI_ERR_ARG;  // (the selectors read below lie inside the segment)
            const int ns = data[pos + 4];
            if (ns < 1 || ns > 4 || SL != 6 + 2 * ns) return IMCUI_ERR_ARG;
            for (int s = 0; s < ns; ++s)
                for (int c = 0; c < j.nc; ++c)
                    if (j.comp[c].id == data[pos + 5 + 2 * s] && !seen[c]) {
                        seen[c] = true;
                        ++done;
                    }
            const long nx = decode_scan(data, n, (size_t)pos, j);
            if (nx < 0) return (int)nx;
            pos = nx;
        } else if (m == 0xD9) {
            if (j.progressive && done == j.nc) break;  // the last scan is in
            return IMCUI_ERR_ARG;  // EOI before every component was coded
        } else {
            // tables between scans (DHT / DQT / DRI): parse the one segment
            const int L = be16(data + pos + 2);
            if (L < 2 || (size_t)pos + 2 + L > n) return IMCUI_ERR_ARG;
            if (m == 0xC4 || m == 0xDB || m == 0xDD) {
                const unsigned char* p = data + pos + 4;
                const int pl = L - 2;
                if (m == 0xDD) {
                    if (pl < 2) return IMCUI_ERR_ARG;
                    j.rst = be16(p);
                } else {
                    const int rc2 = (m == 0xDB) ? parse_dqt(p, pl, j) : parse_dht(p, pl, j);
                    if (rc2 != IMCUI_OK) return rc2;
                }
            }
            pos += 2 + L;
        }
    }
    for (int c = 0; c < 3; ++c)
        for (int k = 0; k < 64; ++k) qt[c * 64 + k] = c < j.nc ? j.qt[j.comp[c].tq][k] : 0;
    return IMCUI_OK;
}

// coef [host]: imcui_hip_jpeg_coef_count() int16, zeroed here; qt [host]: [3][64] uint16, the quantisation table of every component
// in natural order.  Re-entrant: decode different images on different host threads.
extern "C" int imcui_hip_jpeg_entropy_decode(const unsigned char* data, size_t n, short* coef, unsigned short* qt) {
    if (!data || !coef || !qt) return IMCUI_ERR_ARG;
    return entropy_decode_impl(data, n, coef, nullptr, qt);
}

// `count` files on `threads` host threads of the library's own (no interpreter lock, no per-image allocation: the caller provides the
// destinations, typically slices of ONE pinned staging buffer laid out plane-major so that the luma coefficients of a whole batch
// cross PCIe in one transfer).  planes [3 * count]: destination of component c of file i at planes[3 * i + c] (unused components
// may be NULL); qt [count][3 * 64]; status [count]: the per-file return code (a refused file does not stop the others).
extern "C" int imcui_hip_jpeg_entropy_decode_batch(const unsigned char* const* data, const size_t* sizes, int count, short* const* planes, unsigned short* qt,
                                                   int* status, int threads) {
    if (!data || !sizes || !planes || !qt || !status || count < 0) return IMCUI_ERR_ARG;
    if (threads < 1) threads = 1;
    if (threads > count) threads = count > 0 ? count : 1;
    std::atomic<int> next(0);
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= count) return;
            status[i] = data[i] ? entropy_decode_impl(data[i], sizes[i], nullptr, planes + 3 * (size_t)i, qt + 192 * (size_t)i) : IMCUI_ERR_ARG;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return IMCUI_OK;
}

// ------------------------------------------------------------------------------------------------ device side
#define CONST_BITS 13
#define PASS1_BITS 2
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

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one 1-D pass of jpeg_idct_islow on (d0 .. d7); SHIFT = the final descale, outputs in o[8]
template <int SHIFT>
__device__ __forceinline__ void idct8(const int* d, int* o) {
    int z2 = d[2], z3 = d[6];
    int z1 = (z2 + z3) * FIX_0_541196100;
    int tmp2 = z1 + z3 * (-FIX_1_847759065);
    int tmp3 = z1 + z2 * FIX_0_765366865;
    int tmp0 = (d[0] + d[4]) << CONST_BITS;
    int tmp1 = (d[0] - d[4]) << CONST_BITS;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = d[7];
    tmp1 = d[5];
    tmp2 = d[3];
    tmp3 = d[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * FIX_1_175875602;
    tmp0 *= FIX_0_298631336;
    tmp1 *= FIX_2_053119869;
    tmp2 *= FIX_3_072711026;
    tmp3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223;
    z2 *= -FIX_2_562915447;
    z3 *= -FIX_1_961570560;
    z4 *= -FIX_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    o[0] = descale(tmp10 + tmp3, SHIFT);
    o[7] = descale(tmp10 - tmp3, SHIFT);
    o[1] = descale(tmp11 + tmp2, SHIFT);
    o[6] = descale(tmp11 - tmp2, SHIFT);
    o[2] = descale(tmp12 + tmp1, SHIFT);
    o[5] = descale(tmp12 - tmp1, SHIFT);
    o[3] = descale(tmp13 + tmp0, SHIFT);
    o[4] = descale(tmp13 - tmp0, SHIFT);
}

// A workgroup of 256 threads reconstructs 32 blocks: thread = (block, column) in pass 1 (dequantise + columns, results parked in
// LDS), (block, row) in pass 2 (rows, range limit, one 8-byte store per row).  plane: [bh * 8][bw * 8] uint8.  blockIdx.y = image of
// the batch (coefficients `nblocks * 64` apart, tables 192 apart, planes `nblocks * 64` bytes apart).
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const short* __restrict__ coef, const unsigned short* __restrict__ qt, unsigned char* __restrict__ plane,
                                                        int bw, int nblocks) {
    __shared__ int ws[32][8][9];
    const int tid = threadIdx.x, lb = tid >> 3, k = tid & 7;
    const int blk = blockIdx.x * 32 + lb;
    coef += (size_t)blockIdx.y * nblocks * 64;
    qt += (size_t)blockIdx.y * 192;
    plane += (size_t)blockIdx.y * nblocks * 64;
    if (blk < nblocks) {
        const short* c = coef + (size_t)blk * 64;
        int d[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) d[r] = (int)c[8 * r + k] * (int)qt[8 * r + k];
        idct8<CONST_BITS - PASS1_BITS>(d, o);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[lb][r][k] = o[r];
    }
    __syncthreads();
    if (blk < nblocks) {
        int d[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) d[r] = ws[lb][k][r];
        idct8<CONST_BITS + PASS1_BITS + 3>(d, o);
        const int by = blk / bw, bx = blk - by * bw;
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            lo |= (unsigned)clamp255(o[r] + 128) << (8 * r);
            hi |= (unsigned)clamp255(o[r + 4] + 128) << (8 * r);
        }
        *reinterpret_cast<uint2*>(plane + ((size_t)(by * 8 + k) * bw + bx) * 8) = make_uint2(lo, hi);
    }
}

// gray output: the luma plane cropped to the image (blockIdx.z = image of the batch)
__global__ __launch_bounds__(256) void jpeg_crop_kernel(const unsigned char* __restrict__ plane, int pstride, size_t plane_bytes, unsigned char* __restrict__ out, int W, int H) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < W && y < H) out[((size_t)blockIdx.z * H + y) * W + x] = plane[(size_t)blockIdx.z * plane_bytes + (size_t)y * pstride + x];
}

// chroma sample of output pixel (x, y) by libjpeg's fancy up-sampling; cw x chh = the component's real size, (hs, vs) = (1, 1), (2, 1), (2, 2)
__device__ __forceinline__ int chroma_at(const unsigned char* __restrict__ p, int ps, int cw, int chh, int hs, int vs, int x, int y) {
    if (hs == 1 && vs == 1) return p[(size_t)y * ps + x];
    if (vs == 1) {  // h2v1_fancy_upsample
        const int cx = x >> 1, a = p[(size_t)y * ps + cx];
        if (x & 1) return cx == cw - 1 ? a : (3 * a + p[(size_t)y * ps + cx + 1] + 2) >> 2;
        return cx == 0 ? a : (3 * a + p[(size_t)y * ps + cx - 1] + 1) >> 2;
    }
    // rows: the nearer input row weighs 3, the farther 1; above the first / below the last row the edge row is duplicated
    const int cy = y >> 1;
    int oy = (y & 1) ? cy + 1 : cy - 1;
    oy = oy < 0 ? 0 : (oy > chh - 1 ? chh - 1 : oy);
    // h2v2_fancy_upsample: column sums 3 * near row + far row, then 3 : 1 across columns, biases 8 (even x) and 7 (odd x)
    const int cx = x >> 1;
    const unsigned char *r0 = p + (size_t)cy * ps, *r1 = p + (size_t)oy * ps;
    const int cur = 3 * r0[cx] + r1[cx];
    if (x & 1) {
        if (cx == cw - 1) return (cur * 4 + 7) >> 4;
        return (cur * 3 + 3 * r0[cx + 1] + r1[cx + 1] + 7) >> 4;
    }
    if (cx == 0) return (cur * 4 + 8) >> 4;
    return (cur * 3 + 3 * r0[cx - 1] + r1[cx - 1] + 8) >> 4;
}

// RGB output [H][W][3]: up-sample + ycc_rgb_convert (blockIdx.z = image of the batch)
__global__ __launch_bounds__(256) void jpeg_color_kernel(const unsigned char* __restrict__ Y, int ys, size_t ybytes, const unsigned char* __restrict__ Cb,
                                                         const unsigned char* __restrict__ Cr, int cs, size_t cbytes, int cw, int chh, int hs, int vs,
                                                         unsigned char* __restrict__ out, int W, int H) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    Y += (size_t)blockIdx.z * ybytes;
    const int yy = Y[(size_t)y * ys + x];
    // (a one-component file as RGB: no chroma planes, cb = cr = 128 -> three equal channels)
    int cb = 0, cr = 0;
    if (Cb) {
        cb = chroma_at(Cb + (size_t)blockIdx.z * cbytes, cs, cw, chh, hs, vs, x, y) - 128;
        cr = chroma_at(Cr + (size_t)blockIdx.z * cbytes, cs, cw, chh, hs, vs, x, y) - 128;
    }
    const int r = yy + ((91881 * cr + 32768) >> 16);
    const int g = yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    const int b = yy + ((116130 * cb + 32768) >> 16);
    unsigned char* o = out + (((size_t)blockIdx.z * H + y) * W + x) * 3;
    o[0] = (unsigned char)clamp255(r);
    o[1] = (unsigned char)clamp255(g);
    o[2] = (unsigned char)clamp255(b);
}

extern "C" size_t imcui_hip_jpeg_workspace_bytes_batch(const int* info, int gray, int n) {
    if (!info || n <= 0) return 0;
    size_t t = 256;
    const int nc = (gray || info[JI_NC] == 1) ? 1 : 3;
    for (int c = 0; c < nc; ++c) t += align_up((size_t)n * info[JI_MX] * info[JI_COMP + 4 * c] * 8 * info[JI_MY] * info[JI_COMP + 4 * c + 1] * 8, 256);
    return t;
}
extern "C" size_t imcui_hip_jpeg_workspace_bytes(const int* info, int gray) { return imcui_hip_jpeg_workspace_bytes_batch(info, gray, 1); }

// n files of ONE geometry (equal info records): coef_y / coef_cb / coef_cr [dev]: [n][plane coefficients] per component (the chroma
// pointers may be NULL when gray != 0 or the files have one component), qt [dev]: [n][3 * 64] uint16, info [host]: the common record;
// out [dev]: gray != 0 -> [n][H][W] uint8 (the luma planes: cv2.IMREAD_GRAYSCALE), else [n][H][W][3] RGB (a one-component file is
// replicated, as IMREAD_COLOR does).  Three launches for the whole batch.
extern "C" int imcui_hip_jpeg_reconstruct_batch(imcui_hip_t* h, const short* coef_y, const short* coef_cb, const short* coef_cr, const unsigned short* qt,
                                                const int* info, int n, int gray, unsigned char* out, void* ws, size_t ws_bytes, void* stream_) {
    if (!h || !coef_y || !qt || !info || !out) return imcui_set_err(h, IMCUI_ERR_ARG, "jpeg: null argument");
    if (n <= 0) return IMCUI_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const int W = info[JI_W], H = info[JI_H], nc = info[JI_NC];
    if (W <= 0 || H <= 0 || (nc != 1 && nc != 3) || n > 65535) return imcui_set_err(h, IMCUI_ERR_ARG, "jpeg: bad info record / batch of %d", n);
    const int ncd = (gray || nc == 1) ? 1 : 3;
    if (ncd == 3 && (!coef_cb || !coef_cr)) return imcui_set_err(h, IMCUI_ERR_ARG, "jpeg: RGB output needs the chroma coefficients");
    if (ws_bytes < imcui_hip_jpeg_workspace_bytes_batch(info, gray, n) || !ws) return imcui_set_err(h, IMCUI_ERR_WS, "jpeg: workspace too small");
    WsAlloc a(ws, ws_bytes);
    const short* coef[3] = {coef_y, coef_cb, coef_cr};
    unsigned char* plane[3] = {nullptr, nullptr, nullptr};
    int bw[3] = {0, 0, 0}, bh[3] = {0, 0, 0};
    for (int c = 0; c < ncd; ++c) {
        bw[c] = info[JI_MX] * info[JI_COMP + 4 * c];
        bh[c] = info[JI_MY] * info[JI_COMP + 4 * c + 1];
        const int nb = bw[c] * bh[c];
        plane[c] = a.get<unsigned char>((size_t)n * nb * 64);
        hipLaunchKernelGGL(jpeg_idct_kernel, dim3((nb + 31) / 32, n), dim3(256), 0, stream, coef[c], qt + 64 * c, plane[c], bw[c], nb);
    }
    const dim3 grid((W + 255) / 256, H, n);
    const size_t ybytes = (size_t)bw[0] * bh[0] * 64;
    if (gray) {
        hipLaunchKernelGGL(jpeg_crop_kernel, grid, dim3(256), 0, stream, plane[0], bw[0] * 8, ybytes, out, W, H);
    } else if (nc == 1) {
        hipLaunchKernelGGL(jpeg_color_kernel, grid, dim3(256), 0, stream, plane[0], bw[0] * 8, ybytes, (const unsigned char*)nullptr, (const unsigned char*)nullptr, 0,
                           (size_t)0, 0, 0, 0, 0, out, W, H);
    } else {
        const int hs = info[JI_HMAX] / info[JI_COMP + 4], vs = info[JI_VMAX] / info[JI_COMP + 5];
        const int cw = (W * info[JI_COMP + 4] + info[JI_HMAX] - 1) / info[JI_HMAX], chh = (H * info[JI_COMP + 5] + info[JI_VMAX] - 1) / info[JI_VMAX];
        hipLaunchKernelGGL(jpeg_color_kernel, grid, dim3(256), 0, stream, plane[0], bw[0] * 8, ybytes, plane[1], plane[2], bw[1] * 8, (size_t)bw[1] * bh[1] * 64, cw, chh,
                           hs, vs, out, W, H);
    }
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

// one file: coef [dev] = the single buffer of imcui_hip_jpeg_entropy_decode (components one behind the other), qt [dev] its [3][64] tables
extern "C" int imcui_hip_jpeg_reconstruct(imcui_hip_t* h, const short* coef, const unsigned short* qt, const int* info, int gray, unsigned char* out,
                                          void* ws, size_t ws_bytes, void* stream_) {
    if (!h || !coef || !info) return imcui_set_err(h, IMCUI_ERR_ARG, "jpeg: null argument");
    const size_t ny = (size_t)info[JI_MX] * info[JI_COMP] * info[JI_MY] * info[JI_COMP + 1] * 64;
    const size_t ncb = info[JI_NC] == 3 ? (size_t)info[JI_MX] * info[JI_COMP + 4] * info[JI_MY] * info[JI_COMP + 5] * 64 : 0;
    const bool chroma = info[JI_NC] == 3 && !gray;
    return imcui_hip_jpeg_reconstruct_batch(h, coef, chroma ? coef + ny : nullptr, chroma ? coef + ny + ncb : nullptr, qt, info, 1, gray, out, ws, ws_bytes, stream_);
}

// EXIF orientation (tag 0x0112, values 2..8) applied to a decoded image, as cv2.imread and PIL's ImageOps.exif_transpose do:
// dst [H'][W'][C] with (H', W') = (H, W) for 2, 3, 4 and (W, H) for 5..8.  2 mirror, 3 rotate 180, 4 flip, 5 transpose, 6 rotate 90
// clockwise, 7 transverse, 8 rotate 90 counter-clockwise.
__global__ __launch_bounds__(256) void jpeg_orient_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int H, int W, int C, int ori) {
    const int Wd = ori >= 5 ? H : W, Hd = ori >= 5 ? W : H;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= Wd || y >= Hd) return;
    int sy, sx;
    switch (ori) {
        case 2: sy = y; sx = W - 1 - x; break;
        case 3: sy = H - 1 - y; sx = W - 1 - x; break;
        case 4: sy = H - 1 - y; sx = x; break;
        case 5: sy = x; sx = y; break;
        case 6: sy = H - 1 - x; sx = y; break;
        case 7: sy = H - 1 - x; sx = W - 1 - y; break;
        case 8: sy = x; sx = W - 1 - y; break;
        default: sy = y; sx = x; break;
    }
    const unsigned char* s = src + ((size_t)sy * W + sx) * C;
    unsigned char* d = dst + ((size_t)y * Wd + x) * C;
    for (int c = 0; c < C; ++c) d[c] = s[c];
}

// src [dev, H,W,C] uint8 -> dst [dev]: the image in its EXIF orientation (`orientation` 1..8; 1 copies); dst holds H * W * C bytes
extern "C" int imcui_hip_orient_u8(imcui_hip_t* h, const unsigned char* src, int H, int W, int C, int orientation, unsigned char* dst, void* stream_) {
    if (!h || !src || !dst || H <= 0 || W <= 0 || C <= 0 || orientation < 1 || orientation > 8) return imcui_set_err(h, IMCUI_ERR_ARG, "orient: bad argument");
    const int Wd = orientation >= 5 ? H : W, Hd = orientation >= 5 ? W : H;
    hipLaunchKernelGGL(jpeg_orient_kernel, dim3((Wd + 255) / 256, Hd), dim3(256), 0, (hipStream_t)stream_, src, dst, H, W, C, orientation);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}
