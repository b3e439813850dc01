"""PNG decode, host side (no GPU): the library's chunk walk + zlib inflate (C++, through the C ABI) and the oracle's restatement of the
scan-line filters, both PINNED to PIL's decoder: bit-exact on PNG files of the reference repository (committed as fixtures; all 92 of them
when /root/reference is present) and on PIL-encoded files of every supported colour type."""
import ctypes as C
import io
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest
from PIL import Image

from imcui_hip import load_library
from oracle import png as op

HERE = os.path.dirname(os.path.abspath(__file__))


def c_inflate(data: bytes):
    lib = load_library()
    info = (C.c_int * 8)()
    assert lib.imcui_hip_png_info(data, len(data), info) == 0
    nb = lib.imcui_hip_png_raw_bytes(info)
    raw = np.zeros(nb + 64, np.uint8)
    raw[nb:] = 0xA5
    pal = np.zeros(768, np.uint8)
    assert lib.imcui_hip_png_inflate(data, len(data), raw.ctypes.data, nb, pal.ctypes.data) == 0
    assert (raw[nb:] == 0xA5).all(), "the inflater wrote past its buffer"
    return list(info), raw[:nb], pal


def pil_pixels(data: bytes) -> np.ndarray:
    """PIL's decode in `read_image_u8`'s convention: gray (+ alpha) files -> [H,W], everything else -> RGB [H,W,3], alpha dropped."""
    im = Image.open(io.BytesIO(data))
    im.load()
    if im.mode in ("L", "LA"):
        return np.array(im.convert("L"))
    return np.array(im.convert("RGB"))


def encode(arr: np.ndarray, mode: str, **kw) -> bytes:
    buf = io.BytesIO()
    Image.fromarray(arr, mode).save(buf, "PNG", **kw)
    return buf.getvalue()


def texture(seed, h, w, c):
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 120 + 60 * np.sin(xx / 9.0 + seed) + 50 * np.cos(yy / 7.0)
    img = np.stack([base + 30 * np.sin((xx + yy) / (5.0 + k)) for k in range(c)], -1) + g.normal(0, 6, (h, w, c))
    return np.clip(img, 0, 255).astype(np.uint8)


def synthetic_files():
    """(name, bytes) of PIL-encoded PNGs: every colour type the device path takes, odd sizes, a tall and a wide one, different filter heuristics."""
    out = []
    out.append(("rgb", encode(texture(1, 61, 83, 3), "RGB")))
    out.append(("rgb_l9", encode(texture(2, 40, 129, 3), "RGB", compress_level=9)))
    out.append(("rgba", encode(texture(3, 33, 47, 4), "RGBA")))
    out.append(("gray", encode(texture(4, 70, 51, 1)[:, :, 0], "L")))
    la = texture(5, 29, 31, 2)
    out.append(("gray_alpha", encode(la, "LA")))
    pal_img = Image.fromarray(texture(6, 45, 77, 3), "RGB").quantize(colors=200)
    buf = io.BytesIO()
    pal_img.save(buf, "PNG")
    out.append(("palette", buf.getvalue()))
    out.append(("one_pixel", encode(texture(7, 1, 1, 3), "RGB")))
    out.append(("one_row", encode(texture(8, 1, 300, 3), "RGB")))
    out.append(("tall", encode(texture(9, 1500, 12, 3), "RGB")))  # more rows than a workgroup has threads: two strips on the device
    return out


def reference_files():
    z = np.load(os.path.join(HERE, "golden", "png_reference_files.npz"))
    n = len([k for k in z.files if k.startswith("bytes")])
    return [(str(z[f"name{i}"]), z[f"bytes{i}"].tobytes(), z[f"pixels{i}"]) for i in range(n)]


def test_oracle_and_c_inflate_on_reference_files():
    for name, data, want in reference_files():
        assert np.array_equal(pil_pixels(data), want), name  # the fixture is what PIL decodes today
        info, raw, _ = c_inflate(data)
        W, H, ct, depth, lace, pal, idat = op.parse(data)
        assert info[:3] == [W, H, ct] and depth == 8 and lace == 0
        assert np.array_equal(raw, np.frombuffer(zlib.decompress(idat), np.uint8)), name
        assert np.array_equal(op.decode(data), want), name


@pytest.mark.parametrize("name,data", synthetic_files(), ids=[n for n, _ in synthetic_files()])
def test_oracle_and_c_inflate_on_every_colour_type(name, data):
    info, raw, pal = c_inflate(data)
    W, H, ct, depth, lace, ppal, idat = op.parse(data)
    assert info[0] == W and info[1] == H and info[2] == ct and info[4] == (1 if ct in (0, 4) else 3) and info[5] == {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ct]
    assert np.array_equal(raw, np.frombuffer(zlib.decompress(idat), np.uint8))
    if ct == 3:
        assert info[6] == len(ppal) and np.array_equal(pal[: 3 * len(ppal)].reshape(-1, 3), ppal)
    assert np.array_equal(op.decode(data), pil_pixels(data))


def test_every_png_of_the_reference_repository():
    """All 92 PNG files under /root/reference (the WxBS / EVD evaluation images): the C inflater against zlib, and the oracle against PIL on a
    sample of them (the pure-Python Paeth walk takes a second or two per file)."""
    if not os.path.isdir("/root/reference"):
        pytest.skip("no reference tree on this box (the committed fixtures cover it)")
    files = sorted(subprocess.check_output(["find", "/root/reference", "-iname", "*.png"]).decode().split(), key=os.path.getsize)
    assert len(files) >= 90
    for f in files:
        data = open(f, "rb").read()
        info, raw, _ = c_inflate(data)
        assert np.array_equal(raw, np.frombuffer(zlib.decompress(op.parse(data)[6]), np.uint8)), f
    for f in files[:3] + files[40:42]:
        data = open(f, "rb").read()
        assert np.array_equal(op.decode(data), pil_pixels(data)), f


def _with_ihdr(data: bytes, **kw) -> bytes:
    W, H, depth, ct, comp, filt, lace = struct.unpack(">IIBBBBB", data[16:29])
    v = dict(W=W, H=H, depth=depth, ct=ct, comp=comp, filt=filt, lace=lace)
    v.update(kw)
    body = struct.pack(">IIBBBBB", v["W"], v["H"], v["depth"], v["ct"], v["comp"], v["filt"], v["lace"])
    return data[:16] + body + struct.pack(">I", zlib.crc32(b"IHDR" + body)) + data[33:]


def test_unsupported_and_damaged_files_are_refused():
    lib = load_library()
    info = (C.c_int * 8)()
    good = encode(texture(11, 40, 40, 3), "RGB")
    assert lib.imcui_hip_png_info(good, len(good), info) == 0
    for bad, want in ((_with_ihdr(good, lace=1), -4), (_with_ihdr(good, depth=16), -4), (_with_ihdr(good, depth=4, ct=0), -4), (_with_ihdr(good, ct=5), -1),
                      (_with_ihdr(good, W=0), -1), (_with_ihdr(good, comp=1), -1), (b"\xff\xd8\xff\xe0" + good[4:], -1), (good[:30], -1)):  # fmt: skip
        assert lib.imcui_hip_png_info(bad, len(bad), info) == want
    i16 = io.BytesIO()
    Image.fromarray(texture(12, 20, 20, 1)[:, :, 0].astype(np.uint16) << 8).save(i16, "PNG")
    assert lib.imcui_hip_png_info(i16.getvalue(), len(i16.getvalue()), info) == -4  # a real 16-bit file: the caller keeps its host reader
    nb = 40 * (40 * 3 + 1)
    raw = np.zeros(nb, np.uint8)
    cut = good[: len(good) - 40]
    assert lib.imcui_hip_png_inflate(cut, len(cut), raw.ctypes.data, nb, None) == -1  # truncated stream
    assert lib.imcui_hip_png_inflate(good, len(good), raw.ctypes.data, nb - 1, None) == -1  # a destination of the wrong size
    taller = _with_ihdr(good, H=41)
    raw2 = np.zeros(41 * 121, np.uint8)
    assert lib.imcui_hip_png_inflate(taller, len(taller), raw2.ctypes.data, len(raw2), None) == -1  # fewer scan lines than the header promises


def test_mutated_files_never_overrun_or_crash():
    """1500 mutations (byte flips, truncation, insertions, deletions) of three valid files: each is decoded or refused with a status code, the guard
    bytes behind the destination stay untouched."""
    lib = load_library()
    rng = np.random.default_rng(0)
    seeds = [encode(texture(21, 30, 44, 3), "RGB"), encode(texture(22, 25, 25, 1)[:, :, 0], "L"), synthetic_files()[5][1]]
    decoded = refused = 0
    for it in range(1500):
        b = bytearray(seeds[it % 3])
        mode = it % 4
        if mode == 0:
            for _ in range(rng.integers(1, 5)):
                b[rng.integers(8, len(b))] = rng.integers(0, 256)
        elif mode == 1:
            b = b[: rng.integers(8, len(b))]
        elif mode == 2:
            i = rng.integers(8, len(b))
            b[i:i] = bytes(rng.integers(0, 256, rng.integers(1, 16)).astype(np.uint8))
        else:
            i = rng.integers(8, len(b) - 8)
            del b[i : i + rng.integers(1, 8)]
        data = bytes(b)
        info = (C.c_int * 8)()
        if lib.imcui_hip_png_info(data, len(data), info) != 0:
            refused += 1
            continue
        nb = lib.imcui_hip_png_raw_bytes(info)
        if nb > 50_000_000:
            refused += 1
            continue
        raw = np.zeros(nb + 64, np.uint8)
        raw[nb:] = 0x5A
        rc = lib.imcui_hip_png_inflate(data, len(data), raw.ctypes.data, nb, None)
        assert (raw[nb:] == 0x5A).all(), "the inflater wrote past its buffer"
        assert rc in (0, -1, -4)
        decoded += rc == 0
        refused += rc != 0
    assert decoded >= 5 and refused > 300, (decoded, refused)  # (zlib's adler32 catches nearly every mutation of the stream: the survivors hit ancillary chunks / CRC fields)


def test_batch_inflate_equals_single_file_calls():
    lib = load_library()
    files = [d for _, d in synthetic_files()[:6]] + [b"not a png at all"]
    n = len(files)
    infos, sizes = [], []
    for d in files[:-1]:
        info = (C.c_int * 8)()
        assert lib.imcui_hip_png_info(d, len(d), info) == 0
        infos.append(info)
        sizes.append(lib.imcui_hip_png_raw_bytes(info))
    sizes.append(16)
    bufs = [np.zeros(s, np.uint8) for s in sizes]
    pals = np.zeros(768 * n, np.uint8)
    status = (C.c_int * n)()
    rc = lib.imcui_hip_png_inflate_batch((C.c_char_p * n)(*files), (C.c_size_t * n)(*[len(d) for d in files]), n, (C.c_void_p * n)(*[b.ctypes.data for b in bufs]),
                                         (C.c_size_t * n)(*sizes), pals.ctypes.data, status, 4)  # fmt: skip
    assert rc == 0 and list(status)[:-1] == [0] * (n - 1) and status[n - 1] == -1
    for d, b in zip(files[:-1], bufs):
        assert np.array_equal(b, c_inflate(d)[1])


def _chunks(data: bytes):
    i, out = 8, []
    while i + 12 <= len(data):
        n = struct.unpack(">I", data[i : i + 4])[0]
        out.append((data[i + 4 : i + 8], data[i + 8 : i + 8 + n]))
        i += 12 + n
    return out


def _assemble(chunks) -> bytes:
    return b"\x89PNG\r\n\x1a\n" + b"".join(struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b)) for t, b in chunks)


def test_zlib_trailer_split_across_idat_chunks_is_a_valid_file():
    """the adler32 trailer may straddle an IDAT boundary (libpng writes fixed-size IDATs, so this happens in the wild); with
    the image already full, inflate is legally mid-trailer.  Every split position of the stream's last 6 bytes decodes; a stream that
    holds MORE pixel data than the header promises is still refused; so is a corrupted trailer."""
    lib = load_library()
    arr = texture(21, 33, 47, 3)
    good = encode(arr, "RGB")
    ch = _chunks(good)
    z = b"".join(b for t, b in ch if t == b"IDAT")
    head = [c for c in ch if c[0] not in (b"IDAT", b"IEND")]
    nb = 33 * (47 * 3 + 1)
    for cut in range(1, 7):
        for pieces in ([z[:-cut], z[-cut:]], [z[: len(z) // 2], z[len(z) // 2 : -cut], z[-cut:-1], z[-1:]], [z[:-cut], b"", z[-cut:]]):
            data = _assemble(head + [(b"IDAT", p) for p in pieces] + [(b"IEND", b"")])
            assert np.array_equal(pil_pixels(data), arr)
            raw = np.zeros(nb + 16, np.uint8)
            raw[nb:] = 0xA5
            assert lib.imcui_hip_png_inflate(data, len(data), raw.ctypes.data, nb, None) == 0, (cut, [len(p) for p in pieces])
            assert (raw[nb:] == 0xA5).all()
            assert np.array_equal(op.decode(data), arr)
    raw = np.zeros(nb, np.uint8)
    bad = _assemble(head + [(b"IDAT", z[:-4]), (b"IDAT", bytes([z[-4] ^ 1]) + z[-3:]), (b"IEND", b"")])
    assert lib.imcui_hip_png_inflate(bad, len(bad), raw.ctypes.data, nb, None) == -1  # adler32 mismatch
    shorter = _with_ihdr(good, H=32)  # the stream now holds one scan line more than the image
    raw = np.zeros(32 * (47 * 3 + 1) + 16, np.uint8)
    raw[-16:] = 0xA5
    assert lib.imcui_hip_png_inflate(shorter, len(shorter), raw.ctypes.data, len(raw) - 16, None) == -1
    assert (raw[-16:] == 0xA5).all()


def _exif(orientation: int, big_endian: bool) -> bytes:
    e = ">" if big_endian else "<"
    return (b"MM\x00*" if big_endian else b"II*\x00") + struct.pack(e + "I", 8) + struct.pack(e + "H", 1) + struct.pack(e + "HHIHH", 0x0112, 3, 1, orientation, 0) + struct.pack(e + "I", 0)


def test_exif_orientation_is_left_to_the_host_reader():
    """cv2.imread and PIL's exif_transpose rotate / mirror a PNG by its eXIf orientation; the device path does not, so it refuses such
    files (status -4: the caller keeps its host reader) and takes orientation 1 / a chunk without the tag."""
    lib = load_library()
    info = (C.c_int * 8)()
    good = encode(texture(22, 24, 36, 3), "RGB")
    ch = _chunks(good)
    for be in (False, True):
        for o in range(1, 9):
            data = _assemble(ch[:1] + [(b"eXIf", _exif(o, be))] + ch[1:])
            assert np.array_equal(pil_pixels(data), pil_pixels(good))  # a valid file either way (PIL decodes the stored pixels)
            assert lib.imcui_hip_png_info(data, len(data), info) == (0 if o == 1 else -4), (be, o)
    for body in (b"", b"II*\x00\x08\x00\x00\x00\x00\x00", b"garbage!", _exif(3, False)[:12]):
        data = _assemble(ch[:1] + [(b"eXIf", body)] + ch[1:])
        assert lib.imcui_hip_png_info(data, len(data), info) == 0  # no readable orientation = 1, like the readers


def test_gray_of_a_colour_png_is_libpngs(tmp_path):
    """`read_image(path, grayscale=True)` on a colour PNG = cv2.imread(IMREAD_GRAYSCALE) = libpng's truncating 15-bit formula, not cvtColor's
    (oracle/png.py: libpng_rgb_to_gray).  The host reader returns it (with cv2, or restated on top of PIL); pinned to cv2 where installed."""
    from imcui_hip.hloc.extract_features import read_image_u8
    from oracle.preprocess import rgb_to_gray_u8

    rgb = texture(23, 50, 70, 3)
    rgb[:5] = rgb[:5, :, :1]  # some r == g == b pixels: passed through
    want = op.libpng_rgb_to_gray(rgb)
    assert np.array_equal(want[:5], rgb[:5, :, 0])
    n_diff = int((want != rgb_to_gray_u8(rgb)).sum())
    assert 0 < n_diff and np.abs(want.astype(int) - rgb_to_gray_u8(rgb).astype(int)).max() == 1  # one level, on many pixels
    for name, data in (("c.png", encode(rgb, "RGB")), ("a.png", encode(np.dstack([rgb, rgb[..., :1]]), "RGBA"))):
        p = tmp_path / name
        p.write_bytes(data)
        assert np.array_equal(read_image_u8(p, True), want), name
        assert np.array_equal(read_image_u8(p, False), rgb), name
    pal = Image.fromarray(rgb).quantize(64)
    pal.save(tmp_path / "q.png")
    assert np.array_equal(read_image_u8(tmp_path / "q.png", True), op.libpng_rgb_to_gray(np.array(pal.convert("RGB"))))
    g = tmp_path / "g.png"
    g.write_bytes(encode(rgb[..., 0], "L"))
    assert np.array_equal(read_image_u8(g, True), rgb[..., 0])
    cv2 = pytest.importorskip("cv2")  # the pin: OpenCV's own reader on the same files
    for name in ("c.png", "a.png", "q.png", "g.png"):
        assert np.array_equal(cv2.imread(str(tmp_path / name), cv2.IMREAD_GRAYSCALE), read_image_u8(tmp_path / name, True)), name
