"""CPU restatement of the PNG scan-line reconstruction (test infrastructure only: nothing under image-matching-webui_amd/ imports it).

Follows the PNG specification (RFC 2083 / ISO 15948, section 6 "Filter Algorithms", section 9.2 the Paeth predictor) -- what libpng, the
library behind the reference's `cv2.imread` (imcui/hloc/utils/io.py:11-21), implements.  PINNED: `decode` equals PIL's decoder bit for bit
on the PNG files of the reference repository and on PIL-encoded files of every supported colour type (tests/test_png_cpu.py).  8-bit,
non-interlaced files only, like the device path (csrc/png.hip) it checks."""
from __future__ import annotations

import struct
import zlib

import numpy as np

SIG = b"\x89PNG\r\n\x1a\n"


def parse(data: bytes):
    """-> (W, H, colour type, bit depth, interlace, palette [n,3] or None, concatenated IDAT bytes)"""
    assert data[:8] == SIG
    i, idat, pal, hdr = 8, b"", None, None
    while i + 12 <= len(data):
        (ln,) = struct.unpack(">I", data[i : i + 4])
        typ, body = data[i + 4 : i + 8], data[i + 8 : i + 8 + ln]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            pal = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif typ == b"IDAT":
            idat += body
        elif typ == b"IEND":
            break
        i += 12 + ln
    W, H, depth, ct, _, _, lace = hdr
    return W, H, ct, depth, lace, pal, idat


def unfilter(raw: np.ndarray, W: int, H: int, bpp: int) -> np.ndarray:
    """raw: H x (1 + W * bpp) filtered bytes -> [H, W * bpp] reconstructed bytes.  Filters 0 (None), 1 (Sub), 2 (Up) are vectorised per byte
    lane; 3 (Average) and 4 (Paeth) walk the row (the recurrence is not a prefix sum)."""
    raw = raw.reshape(H, 1 + W * bpp)
    out = np.zeros((H, W * bpp), np.uint8)
    prev = np.zeros(W * bpp, np.int32)
    for r in range(H):
        ft, f = int(raw[r, 0]), raw[r, 1:].astype(np.int32)
        if ft == 0:
            cur = f
        elif ft == 1:
            cur = f.copy()
            for c in range(bpp):  # x[i] = f[i] + x[i - bpp]: a running sum per byte lane, modulo 256
                cur[c::bpp] = np.cumsum(f[c::bpp]) & 255
        elif ft == 2:
            cur = (f + prev) & 255
        elif ft in (3, 4):
            cur = np.zeros(W * bpp, np.int32)
            for i in range(W * bpp):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    c = prev[i - bpp] if i >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (f[i] + pred) & 255
        else:
            raise ValueError(f"filter type {ft}")
        out[r] = cur
        prev = cur
    return out


def decode(data: bytes) -> np.ndarray:
    """-> uint8 [H,W] (gray, gray + alpha) or [H,W,3] RGB (RGB, RGBA, palette), alpha dropped: `extract_features.read_image_u8`'s convention."""
    W, H, ct, depth, lace, pal, idat = parse(data)
    if depth != 8 or lace != 0:
        raise ValueError("8-bit non-interlaced files only")
    cin = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ct]
    px = unfilter(np.frombuffer(zlib.decompress(idat), np.uint8), W, H, cin).reshape(H, W, cin)
    if ct == 0:
        return px[:, :, 0].copy()
    if ct == 4:
        return px[:, :, 0].copy()
    if ct == 3:
        full = np.zeros((256, 3), np.uint8)
        full[: len(pal)] = pal
        return full[px[:, :, 0]]
    return px[:, :, :3].copy()


def libpng_rgb_to_gray(rgb: np.ndarray) -> np.ndarray:
    """Gray of a colour PNG as `cv2.imread(path, IMREAD_GRAYSCALE)` returns it (`read_image(path, grayscale=True)`,
    imcui/hloc/utils/io.py:11-21): OpenCV's PNG reader asks libpng for it, `png_set_rgb_to_gray(png_ptr, 1, 0.299, 0.587)`
    (modules/imgcodecs/src/grfmt_png.cpp).  libpng (pngrtran.c): `png_set_rgb_to_gray_fixed` scales the weights given in 1/100000 to
    15 bits by integer division -- 29900 * 32768 / 100000 = 9797, 58700 * 32768 / 100000 = 19234, blue = 32768 - 9797 - 19234 = 3737 --
    and `png_do_rgb_to_gray` (8-bit samples, no gamma tables: OpenCV sets no gamma) writes `(rc*r + gc*g + bc*b) >> 15`, the "historical
    approach which simply truncates"; pixels with r == g == b are copied (the formula gives the same, the weights sum to 2^15).
    Parity unpinned here (cv2 is not in the image); tests/test_png_cpu.py pins it to cv2.imread wherever cv2 is installed."""
    c = np.asarray(rgb).astype(np.int64)
    return ((9797 * c[..., 0] + 19234 * c[..., 1] + 3737 * c[..., 2]) >> 15).astype(np.uint8)
