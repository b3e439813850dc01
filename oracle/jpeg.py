"""CPU restatement of the baseline-JPEG decode of `read_image` (imcui/hloc/utils/io.py:11-21: cv2.imread, i.e. libjpeg-turbo with its
default settings) -- TEST INFRASTRUCTURE: the checker of csrc/jpeg.hip, never imported by the product path.

PARITY PINNED: tests/test_jpeg_cpu.py compares `decode()` with PIL's decoder (libjpeg-turbo, the same library family cv2 links, same
defaults: dct_method = JDCT_ISLOW, do_fancy_upsampling = TRUE) on every JPEG of the reference repository (tests/data, imcui/datasets)
and on PIL-encoded files of every supported sampling mode, restart interval and odd size -- bit for bit, RGB and gray.

What it restates (libjpeg-turbo sources, not in /root/reference -- third-party C library behind cv2; the published algorithm):
  * entropy decoding -- ITU T.81 Annex F (Huffman, DC prediction, restart markers), a plain Python bit loop (small images only);
  * jidctint.c `jpeg_idct_islow` -- the 8x8 inverse DCT in 32-bit integers (numpy, vectorised over the blocks);
  * jdsample.c `h2v1_fancy_upsample`, `h2v2_fancy_upsample`;
  * jdcolor.c `ycc_rgb_convert` (16-bit fixed point tables).
"""
from __future__ import annotations

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
                   57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])  # fmt: skip

CONST_BITS, PASS1_BITS = 13, 2
F = dict(f0298=2446, f0390=3196, f0541=4433, f0765=6270, f0899=7373, f1175=9633, f1501=12299, f1847=15137, f1961=16069, f2053=16819,
         f2562=20995, f3072=25172)  # fmt: skip


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(d, shift):
    """One pass of jpeg_idct_islow over the leading axis of `d` ([8, ...] int64 holding 32-bit values)."""
    z2, z3 = d[2], d[6]
    z1 = (z2 + z3) * F["f0541"]
    tmp2 = z1 + z3 * (-F["f1847"])
    tmp3 = z1 + z2 * F["f0765"]
    tmp0 = (d[0] + d[4]) << CONST_BITS
    tmp1 = (d[0] - d[4]) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = d[7], d[5], d[3], d[1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F["f1175"]
    tmp0, tmp1, tmp2, tmp3 = tmp0 * F["f0298"], tmp1 * F["f2053"], tmp2 * F["f3072"], tmp3 * F["f1501"]
    z1, z2, z3, z4 = z1 * -F["f0899"], z2 * -F["f2562"], z3 * -F["f1961"] + z5, z4 * -F["f0390"] + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    out = np.stack([tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3])
    # (32-bit wrap-around of the C code: the products stay below 2^31 for every legal coefficient, the wrap is kept for the rest)
    out = ((out + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)
    return _descale(out, shift)


def idct_islow(coef: np.ndarray, qt: np.ndarray) -> np.ndarray:
    """coef [nby, nbx, 64] int16 (natural order), qt [64] -> samples [nby * 8, nbx * 8] uint8."""
    nby, nbx, _ = coef.shape
    d = (coef.astype(np.int64) * qt.astype(np.int64)[None, None, :]).reshape(nby, nbx, 8, 8)  # [.., row, col]
    ws = _idct_1d(np.moveaxis(d, 2, 0), CONST_BITS - PASS1_BITS)          # columns: axis 0 = frequency row -> spatial row
    out = _idct_1d(np.moveaxis(ws, 3, 0), CONST_BITS + PASS1_BITS + 3)    # rows: axis 0 = column index -> spatial column
    # ws: [row, nby, nbx, col]; out: [col, row, nby, nbx]
    img = np.clip(out + 128, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(img.transpose(2, 1, 3, 0).reshape(nby * 8, nbx * 8))


def upsample_fancy(p: np.ndarray, hs: int, vs: int, W: int, H: int) -> np.ndarray:
    """Chroma plane `p` (its real size [ch, cw]) -> [H, W] by libjpeg's fancy up-sampling for the factors (1, 1), (2, 1), (2, 2)."""
    p = p.astype(np.int32)
    ch, cw = p.shape
    if hs == 1 and vs == 1:
        return p[:H, :W]
    if vs == 2:
        up = np.concatenate((p[:1], p[:-1]), 0)   # row above (edge duplicated)
        dn = np.concatenate((p[1:], p[-1:]), 0)   # row below
        if hs == 1:
            raise NotImplementedError("4:4:0 (h1v2 up-sampling) is not pinned: no encoder for it in this image")
        col = np.empty((2 * ch, cw), np.int32)
        col[0::2] = 3 * p + up
        col[1::2] = 3 * p + dn
        left = np.concatenate((col[:, :1], col[:, :-1]), 1)
        right = np.concatenate((col[:, 1:], col[:, -1:]), 1)
        out = np.empty((2 * ch, 2 * cw), np.int32)
        out[:, 0::2] = (col * 3 + left + 8) >> 4
        out[:, 1::2] = (col * 3 + right + 7) >> 4
        out[:, 0] = (col[:, 0] * 4 + 8) >> 4
        out[:, 2 * cw - 1] = (col[:, -1] * 4 + 7) >> 4
        return out[:H, :W]
    # h2v1
    left = np.concatenate((p[:, :1], p[:, :-1]), 1)
    right = np.concatenate((p[:, 1:], p[:, -1:]), 1)
    out = np.empty((ch, 2 * cw), np.int32)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, 2 * cw - 1] = p[:, -1]
    return out[:H, :W]


def ycc_to_rgb(y, cb, cr) -> np.ndarray:
    y, cb, cr = y.astype(np.int32), cb.astype(np.int32) - 128, cr.astype(np.int32) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack((r, g, b), -1), 0, 255).astype(np.uint8)


# ---- the bit stream (slow, pure Python: the checker of the C++ entropy decoder on small files) ---------------------------------------
class _Huff:
    def __init__(self, bits, vals):
        self.lut = {}
        code, k = 0, 0
        for ln in range(1, 17):
            for _ in range(bits[ln - 1]):
                self.lut[(ln, code)] = vals[k]
                code += 1
                k += 1
            code <<= 1


def parse(data: bytes) -> dict:
    """Headers and the entropy-coded scans -> {'W','H','comps': [{h, v, tq, coef [bh, bw, 64]}], 'qt', 'orientation'}."""
    assert data[:2] == b"\xff\xd8"
    i, qt, hd, ha, comps, rst, W, H = 2, {}, {}, {}, [], 0, 0, 0
    info = {}
    while i < len(data):
        assert data[i] == 0xFF, hex(data[i])
        while data[i] == 0xFF:
            i += 1
        m = data[i]
        i += 1
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        if m == 0xD9:
            break
        L = (data[i] << 8) | data[i + 1]
        seg = data[i + 2 : i + L]
        if m == 0xDB:
            o = 0
            while o < len(seg):
                pq, tq = seg[o] >> 4, seg[o] & 15
                t = np.zeros(64, np.int64)
                for k in range(64):
                    t[ZIGZAG[k]] = ((seg[o + 1 + 2 * k] << 8) | seg[o + 2 + 2 * k]) if pq else seg[o + 1 + k]
                qt[tq] = t
                o += 1 + (128 if pq else 64)
        elif m == 0xC4:
            o = 0
            while o < len(seg):
                tc, th = seg[o] >> 4, seg[o] & 15
                bits = list(seg[o + 1 : o + 17])
                n = sum(bits)
                (ha if tc else hd)[th] = _Huff(bits, list(seg[o + 17 : o + 17 + n]))
                o += 17 + n
        elif m in (0xC0, 0xC1):
            assert seg[0] == 8
            H, W, nc = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4], seg[5]
            comps = [dict(id=seg[6 + 3 * c], h=seg[7 + 3 * c] >> 4, v=seg[7 + 3 * c] & 15, tq=seg[8 + 3 * c]) for c in range(nc)]
            if nc == 1:
                comps[0]["h"] = comps[0]["v"] = 1
            hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
            mx, my = -(-W // (8 * hmax)), -(-H // (8 * vmax))
            for c in comps:
                c["coef"] = np.zeros((my * c["v"], mx * c["h"], 64), np.int16)
                c["rbw"], c["rbh"] = -(-(-(-W * c["h"] // hmax)) // 8), -(-(-(-H * c["v"] // vmax)) // 8)
            info = dict(hmax=hmax, vmax=vmax, mx=mx, my=my)
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise NotImplementedError("not a baseline Huffman JPEG")
        elif m == 0xDD:
            rst = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            ns = seg[0]
            sc = []
            for s in range(ns):
                c = next(c for c in comps if c["id"] == seg[1 + 2 * s])
                c["td"], c["ta"] = seg[2 + 2 * s] >> 4, seg[2 + 2 * s] & 15
                sc.append(c)
            i = _scan(data, i + L, sc, hd, ha, rst, info)
            continue
        i += L
    return dict(W=W, H=H, comps=comps, qt=qt, **info)


def _scan(data, pos, sc, hd, ha, rst, info):
    acc, cnt = 0, 0

    def bit():
        nonlocal acc, cnt, pos
        if cnt == 0:
            b = data[pos]
            if b == 0xFF:
                if data[pos + 1] == 0:
                    pos += 2
                else:
                    b = 0  # a marker: feed zeros
            else:
                pos += 1
            acc, cnt = b, 8
        cnt -= 1
        return (acc >> cnt) & 1

    def sym(t):
        code = 0
        for ln in range(1, 17):
            code = (code << 1) | bit()
            if (ln, code) in t.lut:
                return t.lut[(ln, code)]
        raise ValueError("bad Huffman code")

    def receive(s):
        v = 0
        for _ in range(s):
            v = (v << 1) | bit()
        return v if s == 0 or v >= (1 << (s - 1)) else v - (1 << s) + 1

    inter = len(sc) > 1
    ux, uy = (info["mx"], info["my"]) if inter else (sc[0]["rbw"], sc[0]["rbh"])
    pred = {id(c): 0 for c in sc}
    left, nxt = rst, 0
    for u in range(ux * uy):
        if rst and left == 0:
            cnt = 0
            while not (data[pos] == 0xFF and 0xD0 <= data[pos + 1] <= 0xD7):
                pos += 1
            assert data[pos + 1] == 0xD0 + nxt
            pos += 2
            nxt, left = (nxt + 1) & 7, rst
            pred = {id(c): 0 for c in sc}
        x, y = u % ux, u // ux
        for c in sc:
            for by in range(c["v"] if inter else 1):
                for bx in range(c["h"] if inter else 1):
                    blk = c["coef"][y * c["v"] + by, x * c["h"] + bx] if inter else c["coef"][y, x]
                    s = sym(hd[c["td"]])
                    pred[id(c)] += receive(s)
                    blk[0] = pred[id(c)]
                    k = 1
                    while k < 64:
                        rs = sym(ha[c["ta"]])
                        r, sz = rs >> 4, rs & 15
                        if sz == 0:
                            if r != 15:
                                break
                            k += 16
                            continue
                        k += r
                        blk[ZIGZAG[k]] = receive(sz)
                        k += 1
        if rst:
            left -= 1
    while not (data[pos] == 0xFF and data[pos + 1] not in (0, 0xFF) and not 0xD0 <= data[pos + 1] <= 0xD7):
        pos += 1
    return pos


def reconstruct(W, H, comps, qts, hmax, vmax, gray: bool) -> np.ndarray:
    """comps: [{h, v, coef [bh, bw, 64]}], qts: one [64] table per component -> [H, W] (gray) or [H, W, 3] uint8."""
    planes = [idct_islow(c["coef"], q) for c, q in zip(comps[: 1 if gray else len(comps)], qts)]
    y = planes[0][:H, :W]
    if gray:
        return np.ascontiguousarray(y)
    if len(comps) == 1:
        return np.repeat(y[:, :, None], 3, 2)
    out = []
    for c, p in zip(comps[1:], planes[1:]):
        cw, ch = -(-W * c["h"] // hmax), -(-H * c["v"] // vmax)
        out.append(upsample_fancy(p[:ch, :cw], hmax // c["h"], vmax // c["v"], W, H))
    return ycc_to_rgb(y, out[0], out[1])


def decode(data: bytes, gray: bool = False) -> np.ndarray:
    """The whole decode in Python / numpy (slow bit loop: small files)."""
    j = parse(data)
    return reconstruct(j["W"], j["H"], j["comps"], [j["qt"][c["tq"]] for c in j["comps"]], j["hmax"], j["vmax"], gray)


def orient(img: np.ndarray, orientation: int) -> np.ndarray:
    """EXIF orientation 1..8 applied to a decoded image, as cv2.imread does inside its decoder (pinned to PIL's ImageOps.exif_transpose
    in tests/test_jpeg_cpu.py): 2 mirror, 3 rotate 180, 4 flip, 5 transpose, 6 rotate 90 clockwise, 7 transverse, 8 rotate 90 counter-clockwise."""
    if orientation == 2:
        return img[:, ::-1]
    if orientation == 3:
        return img[::-1, ::-1]
    if orientation == 4:
        return img[::-1]
    t = np.swapaxes(img, 0, 1)
    if orientation == 5:
        return t
    if orientation == 6:
        return t[:, ::-1]
    if orientation == 7:
        return t[::-1, ::-1]
    if orientation == 8:
        return t[::-1]
    return img
