"""JPEG decode, host side (no GPU): the library's entropy decoder (C++, through the C ABI) and the oracle's restatement of libjpeg's
reconstruction (oracle/jpeg.py) against PIL's decoder = libjpeg-turbo, the library family behind the reference's `cv2.imread`
(imcui/hloc/utils/io.py:11-21) -- BIT FOR BIT, RGB and gray: this is what pins the oracle the GPU kernels are then held to."""
import ctypes as C
import glob
import io
import os

import numpy as np
import pytest
from PIL import Image

from imcui_hip import load_library
from oracle import jpeg as oj

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_reference_files.npz")


def c_entropy(data: bytes):
    lib = load_library()
    info = (C.c_int * 24)()
    assert lib.imcui_hip_jpeg_info(data, len(data), info) == 0
    coef = np.zeros(lib.imcui_hip_jpeg_coef_count(info), np.int16)
    qt = np.zeros(192, np.uint16)
    assert lib.imcui_hip_jpeg_entropy_decode(data, len(data), coef.ctypes.data, qt.ctypes.data) == 0
    return list(info), coef, qt


def oracle_from_coefficients(info, coef, qt, gray):
    W, H, nc, hmax, vmax, mx, my = info[:7]
    comps, off = [], 0
    for c in range(nc):
        h, v = info[9 + 4 * c], info[10 + 4 * c]
        n = mx * h * my * v * 64
        comps.append(dict(h=h, v=v, coef=coef[off : off + n].reshape(my * v, mx * h, 64)))
        off += n
    return oj.reconstruct(W, H, comps, [qt[64 * c : 64 * c + 64].astype(np.int64) for c in range(nc)], hmax, vmax, gray)


def pil_decode(data: bytes, gray: bool):
    im = Image.open(io.BytesIO(data))
    if gray:
        im.draft("L", im.size)  # libjpeg's JCS_GRAYSCALE output: what cv2.IMREAD_GRAYSCALE returns for a JPEG
        assert im.mode == "L"
        return np.array(im)
    return np.array(im.convert("RGB"))


def encode(img, **kw):
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", **kw)
    return buf.getvalue()


def smooth_image(seed, h, w, channels=3):
    g = np.random.default_rng(seed)
    y, x = np.mgrid[:h, :w]
    planes = [127 + 90 * np.sin(x / (7.0 + c) + g.uniform(0, 6)) * np.cos(y / (5.0 + 2 * c) + g.uniform(0, 6)) + g.normal(0, 12, (h, w)) for c in range(channels)]
    a = np.clip(np.stack(planes, -1), 0, 255).astype(np.uint8)
    return a[..., 0] if channels == 1 else a


CASES = [(37, 29, "4:2:0", 85, 0), (64, 48, "4:4:4", 92, 0), (50, 33, "4:2:2", 75, 0), (41, 57, "4:2:0", 60, 3), (16, 16, "4:2:0", 30, 1), (9, 7, "4:2:2", 98, 0),
         (8, 8, "4:4:4", 100, 0), (1, 1, "4:2:0", 90, 0), (130, 17, "4:2:0", 10, 5)]  # fmt: skip


@pytest.mark.parametrize("w,h,sub,q,rst", CASES)
def test_oracle_and_c_entropy_decoder_equal_pil(w, h, sub, q, rst):
    """Every supported sampling mode, odd sizes (partial MCUs, 1 x 1), restart intervals, quality 10 .. 100: the pure-Python bit loop
    and the C++ entropy decoder produce the same coefficients; the restated reconstruction equals PIL bit for bit."""
    kw = dict(quality=q, subsampling=sub)
    if rst:
        kw["restart_marker_blocks"] = rst
    data = encode(smooth_image(w * 100 + h, h, w), **kw)
    info, coef, qt = c_entropy(data)
    assert (info[0], info[1], info[2], info[7] > 0) == (w, h, 3, rst > 0)
    j = oj.parse(data)
    assert np.array_equal(np.concatenate([c["coef"].reshape(-1) for c in j["comps"]]), coef)
    for gray in (False, True):
        want = pil_decode(data, gray)
        assert np.array_equal(oj.decode(data, gray), want)
        assert np.array_equal(oracle_from_coefficients(info, coef, qt, gray), want)


def test_gray_files_and_optimised_tables():
    """One-component files (IMREAD_COLOR replicates them) and optimised Huffman tables (code lengths up to 16 bits)."""
    for data in (encode(smooth_image(5, 45, 70, 1), quality=80), encode(smooth_image(6, 64, 64), quality=95, optimize=True, subsampling="4:2:0")):
        info, coef, qt = c_entropy(data)
        for gray in (False, True):
            assert np.array_equal(oracle_from_coefficients(info, coef, qt, gray), pil_decode(data, gray))


def test_reference_repository_files_fixture():
    """Six JPEG files of the reference repository (bytes + PIL's decode committed by tests/golden/make_jpeg_fixtures.py)."""
    z = np.load(GOLD)
    for i in range(6):
        data = z[f"bytes{i}"].tobytes()
        info, coef, qt = c_entropy(data)
        assert info[8] == 1  # upright
        assert np.array_equal(oracle_from_coefficients(info, coef, qt, False), z[f"rgb{i}"]), str(z[f"name{i}"])
        assert np.array_equal(oracle_from_coefficients(info, coef, qt, True), z[f"gray{i}"]), str(z[f"name{i}"])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")
def test_every_jpeg_of_the_reference_repository():
    """All 155 JPEG files under /root/reference (tests/data, imcui/datasets: baseline 4:2:0), RGB and gray, against PIL."""
    files = sorted(sum((glob.glob(f"/root/reference/**/*.{e}", recursive=True) for e in ("jpg", "jpeg", "JPG")), []))
    assert len(files) >= 100
    for f in files:
        data = open(f, "rb").read()
        info, coef, qt = c_entropy(data)
        for gray in (False, True):
            assert np.array_equal(oracle_from_coefficients(info, coef, qt, gray), pil_decode(data, gray)), (f, gray)


def test_unsupported_and_damaged_files_are_refused():
    lib = load_library()
    info = (C.c_int * 24)()
    prog = encode(smooth_image(1, 40, 40), quality=80, progressive=True)
    assert lib.imcui_hip_jpeg_info(prog, len(prog), info) == 0  # progressive Huffman files are taken since round 5 (test below)
    arith = prog.replace(b"\xff\xc2", b"\xff\xca", 1)  # the same frame header announced as arithmetic-coded progressive
    assert lib.imcui_hip_jpeg_info(arith, len(arith), info) == -4  # IMCUI_HIP_ERR_UNSUPPORTED: the caller keeps its host decoder
    cmyk = io.BytesIO()
    Image.fromarray(smooth_image(2, 24, 24)).convert("CMYK").save(cmyk, "JPEG")
    assert lib.imcui_hip_jpeg_info(cmyk.getvalue(), len(cmyk.getvalue()), info) == -4
    png = io.BytesIO()
    Image.fromarray(smooth_image(3, 8, 8)).save(png, "PNG")
    assert lib.imcui_hip_jpeg_info(png.getvalue(), len(png.getvalue()), info) == -1
    good = encode(smooth_image(4, 64, 64), quality=80)
    cut = good[: len(good) // 2]
    coef = np.zeros(64 * 64 * 3, np.int16)
    qt = np.zeros(192, np.uint16)
    assert lib.imcui_hip_jpeg_info(cut, len(cut), info) == 0  # the header is intact ...
    assert lib.imcui_hip_jpeg_entropy_decode(cut, len(cut), coef.ctypes.data, qt.ctypes.data) in (0, -1)  # ... the scan is short: zeros are fed (libjpeg pads too) or refused, never a crash


def exif_jpeg(img, orientation, **kw):
    im = Image.fromarray(img)
    exif = im.getexif()
    exif[0x0112] = orientation
    buf = io.BytesIO()
    im.save(buf, "JPEG", exif=exif, **kw)
    return buf.getvalue()


def test_exif_orientation_is_reported_and_the_restated_transform_equals_pil():
    """The info record carries the EXIF orientation; `oracle.jpeg.orient` (what the device kernel is held to) equals PIL's
    `ImageOps.exif_transpose` -- the transform cv2.imread applies inside its decoder -- for all eight values."""
    from PIL import ImageOps

    lib = load_library()
    info = (C.c_int * 24)()
    for ori in range(1, 9):
        data = exif_jpeg(smooth_image(7 + ori, 20, 30), ori, quality=90)
        assert lib.imcui_hip_jpeg_info(data, len(data), info) == 0 and info[8] == ori
        want = np.array(ImageOps.exif_transpose(Image.open(io.BytesIO(data))).convert("RGB"))
        assert np.array_equal(np.ascontiguousarray(oj.orient(pil_decode(data, False), ori)), want), ori


def test_batch_entropy_decoder_equals_the_single_file_one():
    """`imcui_hip_jpeg_entropy_decode_batch` (the library's own host threads, per-component destinations: what `JpegDecoder` stages into
    one pinned buffer) writes the same coefficients and tables as the single-file entry point; a refused file does not stop the others."""
    lib = load_library()
    blobs = [encode(smooth_image(i, 48 + 8 * (i % 3), 64 + 16 * (i % 2)), quality=70 + i, subsampling=("4:2:0", "4:2:2", "4:4:4")[i % 3]) for i in range(12)]
    blobs += [encode(smooth_image(77, 40, 40, 1), quality=80), encode(smooth_image(1, 40, 40), quality=80, progressive=True).replace(b"\xff\xc2", b"\xff\xca", 1)]
    n = len(blobs)
    infos = []
    for b in blobs[:-1]:
        info = (C.c_int * 24)()
        assert lib.imcui_hip_jpeg_info(b, len(b), info) == 0
        infos.append(list(info))
    cnt = [(i[5] * i[9] * i[6] * i[10] * 64, i[5] * i[13] * i[6] * i[14] * 64 if i[2] == 3 else 0) for i in infos] + [(64, 64)]
    bufs = [[np.full(max(c[0], 1), 7, np.int16), np.full(max(c[1], 1), 7, np.int16), np.full(max(c[1], 1), 7, np.int16)] for c in cnt]
    planes = (C.c_void_p * (3 * n))()
    for i in range(n):
        planes[3 * i] = bufs[i][0].ctypes.data
        if cnt[i][1]:
            planes[3 * i + 1], planes[3 * i + 2] = bufs[i][1].ctypes.data, bufs[i][2].ctypes.data
    qt = np.zeros(n * 192, np.uint16)
    status = (C.c_int * n)()
    assert lib.imcui_hip_jpeg_entropy_decode_batch((C.c_char_p * n)(*blobs), (C.c_size_t * n)(*[len(b) for b in blobs]), n, planes, qt.ctypes.data, status, 4) == 0
    assert status[n - 1] == -4  # the file announced as arithmetic-coded: refused before anything is written
    for i, b in enumerate(blobs[:-1]):
        _, coef, q = c_entropy(b)
        got = np.concatenate([bufs[i][0]] + ([bufs[i][1], bufs[i][2]] if cnt[i][1] else []))
        assert status[i] == 0 and np.array_equal(got, coef) and np.array_equal(qt[192 * i : 192 * i + 192], q), i


def test_mutated_files_never_overrun_or_crash():
    """2000 mutations (byte flips, truncation, injected markers, insertions, deletions) of four valid files through the C entry points:
    every one is either decoded or refused with a status code; the guard words behind the coefficient buffer stay untouched."""
    lib = load_library()
    rng = np.random.default_rng(0)
    seeds = [encode(smooth_image(1, 40, 56), quality=80, subsampling="4:2:0"), encode(smooth_image(2, 33, 47), quality=60, subsampling="4:2:2", restart_marker_blocks=2),
             encode(smooth_image(3, 24, 24, 1), quality=90), encode(smooth_image(4, 64, 64), quality=95, optimize=True, subsampling="4:4:4")]  # fmt: skip
    decoded = refused = 0
    for it in range(2000):
        b = bytearray(seeds[it % 4])
        mode = it % 5
        if mode == 0:
            for _ in range(rng.integers(1, 6)):
                b[rng.integers(2, len(b))] = rng.integers(0, 256)
        elif mode == 1:
            b = b[: rng.integers(4, len(b))]
        elif mode == 2:
            i = rng.integers(2, len(b) - 4)
            b[i : i + 2] = bytes([0xFF, int(rng.integers(0xC0, 0xFF))])
        elif mode == 3:
            i = rng.integers(2, len(b))
            b[i:i] = bytes(rng.integers(0, 256, rng.integers(1, 20)).astype(np.uint8))
        else:
            i = rng.integers(2, len(b) - 10)
            del b[i : i + rng.integers(1, 10)]
        data = bytes(b)
        info = (C.c_int * 24)()
        if lib.imcui_hip_jpeg_info(data, len(data), info) != 0:
            refused += 1
            continue
        n = lib.imcui_hip_jpeg_coef_count(info)
        if n > 50_000_000:  # a mutated size field: the Python layer refuses such images (MAX_PIXELS)
            refused += 1
            continue
        coef = np.zeros(n + 64, np.int16)
        coef[n:] = 12345
        qt = np.zeros(192, np.uint16)
        rc = lib.imcui_hip_jpeg_entropy_decode(data, len(data), coef.ctypes.data, qt.ctypes.data)
        assert (coef[n:] == 12345).all(), "the decoder wrote past its buffer"
        assert rc in (0, -1, -4)
        decoded += rc == 0
        refused += rc != 0
    assert decoded > 200 and refused > 200


def _segments(data: bytes):
    """(marker, start, end) of every header segment up to and including SOS."""
    i, out = 2, []
    while i + 4 <= len(data):
        assert data[i] == 0xFF
        m, L = data[i + 1], (data[i + 2] << 8) | data[i + 3]
        out.append((m, i, i + 2 + L))
        if m == 0xDA:
            break
        i += 2 + L
    return out


def test_over_subscribed_huffman_tables_are_refused():
    """a DHT whose code-length counts do not form a prefix code must be refused (libjpeg's jpeg_make_d_derived_tbl
    test), not indexed past the 512-entry look-ahead table.  Every count byte of every table is pushed over its code space in turn."""
    lib = load_library()
    good = encode(smooth_image(7, 48, 48), quality=85, subsampling="4:2:0")
    info = (C.c_int * 24)()
    assert lib.imcui_hip_jpeg_info(good, len(good), info) == 0
    tried = 0
    for m, s, e in _segments(good):
        if m != 0xC4:
            continue
        o = s + 4
        while o < e:
            cnt = sum(good[o + 1 : o + 17])
            for l in range(1, 17):
                for v in (255, (1 << min(l, 8)) - 1 + 1 if l <= 7 else 255):
                    b = bytearray(good)
                    b[o + l] = v
                    data = bytes(b)
                    rc = lib.imcui_hip_jpeg_info(data, len(data), info)
                    over = sum(c << (16 - k) for k, c in enumerate(data[o + 1 : o + 17], start=1)) > (1 << 16)
                    if over:
                        assert rc == -1, (l, v, rc)
                        tried += 1
            o += 17 + cnt
    assert tried > 50
    # the same table arriving BETWEEN two scans (entropy_decode_impl's own segment walk): Y scan, then a bad DHT, then the Cb / Cr scans
    img = Image.fromarray(smooth_image(8, 32, 32))
    buf = io.BytesIO()
    img.save(buf, "JPEG", quality=80, subsampling="4:4:4")
    data = buf.getvalue()
    bad_dht = bytes([0xFF, 0xC4, 0x00, 0x14, 0x00, 3] + [0] * 15 + [1])  # three codes of length 1
    sos = [s for m, s, _ in _segments(data) if m == 0xDA][0]
    broken = data[:sos] + bad_dht + data[sos:]
    assert lib.imcui_hip_jpeg_info(broken, len(broken), info) == -1


def test_sos_truncation_and_fill_bytes():
    """a file cut right behind an SOS marker is refused without reading past its end, and FF FF .. fill bytes in
    front of a restart marker are skipped like libjpeg skips them (the coefficients stay those of the clean file)."""
    lib = load_library()
    good = encode(smooth_image(9, 40, 56), quality=80, subsampling="4:2:0", restart_marker_blocks=2)
    sos = [s for m, s, _ in _segments(good) if m == 0xDA][0]
    info = (C.c_int * 24)()
    assert lib.imcui_hip_jpeg_info(good, len(good), info) == 0
    n = lib.imcui_hip_jpeg_coef_count(info)
    for cut in range(sos + 2, sos + 14):
        data = good[:cut]
        coef = np.zeros(n, np.int16)
        qt = np.zeros(192, np.uint16)
        # (exact-size bytes object: an over-read would be caught by the allocator's red zone under ASan; here the status is what is checked)
        assert lib.imcui_hip_jpeg_entropy_decode(data, len(data), coef.ctypes.data, qt.ctypes.data) in (-1, 0)
    _, ref, _ = c_entropy(good)
    rst = good.index(b"\xff\xd0", sos)
    padded = good[:rst] + b"\xff\xff\xff" + good[rst:]
    _, got, _ = c_entropy(padded)
    assert np.array_equal(ref, got)
    assert np.array_equal(pil_decode(padded, True), pil_decode(good, True))


def test_rgb_stored_jpegs_are_left_to_the_host_reader():
    """three-component frames that libjpeg does NOT treat as YCbCr -- an Adobe APP14 marker with transform 0, or
    component ids 'R' 'G' 'B' without JFIF / Adobe markers -- are reported unsupported (the drivers then use the host decoder)."""
    lib = load_library()
    good = encode(smooth_image(11, 32, 32), quality=90, subsampling="4:4:4")
    info = (C.c_int * 24)()
    segs = _segments(good)
    app0 = [(s, e) for m, s, e in segs if m == 0xE0]
    assert app0, "PIL writes a JFIF header"
    s0, e0 = app0[0]
    no_jfif = good[:s0] + good[e0:]
    assert lib.imcui_hip_jpeg_info(no_jfif, len(no_jfif), info) == 0  # ids 1, 2, 3 without markers: YCbCr (libjpeg's guess as well)
    adobe = lambda t: bytes([0xFF, 0xEE, 0x00, 0x0E]) + b"Adobe" + bytes([0, 100, 0, 0, 0, 0, t])
    for t, want in ((0, -4), (1, 0), (2, -4)):
        data = good[:s0] + adobe(t) + good[e0:]
        assert lib.imcui_hip_jpeg_info(data, len(data), info) == want, t
    with_jfif_and_adobe0 = good[:e0] + adobe(0) + good[e0:]
    assert lib.imcui_hip_jpeg_info(with_jfif_and_adobe0, len(with_jfif_and_adobe0), info) == 0  # JFIF wins, as in libjpeg
    # component ids R, G, B (SOF0 and SOS selectors)
    b = bytearray(no_jfif)
    for m, s, e in _segments(no_jfif):
        if m == 0xC0:
            for c, ch in enumerate(b"RGB"):
                b[s + 4 + 6 + 3 * c] = ch
        if m == 0xDA:
            for c, ch in enumerate(b"RGB"):
                b[s + 5 + 2 * c] = ch
    data = bytes(b)
    assert lib.imcui_hip_jpeg_info(data, len(data), info) == -4
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)))[..., 0], np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[..., 0])  # (PIL agrees it is RGB: it opens)


@pytest.mark.parametrize("w,h,sub,q,rst,opt", [(64, 64, "4:2:0", 80, 0, False), (57, 43, "4:2:0", 90, 0, False), (123, 77, "4:2:2", 75, 0, True), (40, 56, "4:4:4", 95, 0, False),
                                               (200, 150, "4:2:0", 60, 3, False), (33, 47, "4:2:2", 85, 2, True), (16, 16, "4:4:4", 50, 0, False), (641, 479, "4:2:0", 90, 0, True)])
def test_progressive_files_equal_pil(w, h, sub, q, rst, opt):
    """Round 5 (missing 4): progressive Huffman files (SOF2) -- spectral selection, successive approximation, end-of-band runs,
    DC / AC refinement scans, restart intervals, optimised tables -- decode to the coefficients of the equivalent sequential file: the restated
    reconstruction of them equals PIL bit for bit, RGB and gray.  (PIL writes libjpeg's standard 10-scan script.)"""
    kw = dict(quality=q, subsampling=sub, progressive=True, optimize=opt)
    if rst:
        kw["restart_marker_blocks"] = rst
    data = encode(smooth_image(w + h, h, w), **kw)
    assert b"\xff\xc2" in data[:1200]
    info, coef, qt = c_entropy(data)
    for gray in (False, True):
        assert np.array_equal(oracle_from_coefficients(info, coef, qt, gray), pil_decode(data, gray)), gray
    gdata = encode(smooth_image(5, h, w, 1), quality=q, progressive=True)
    info, coef, qt = c_entropy(gdata)
    assert np.array_equal(oracle_from_coefficients(info, coef, qt, True), pil_decode(gdata, True))


def test_mutated_progressive_files_never_overrun_or_crash():
    lib = load_library()
    rng = np.random.default_rng(1)
    seeds = [encode(smooth_image(31, 40, 56), quality=80, subsampling="4:2:0", progressive=True), encode(smooth_image(32, 33, 47), quality=60, subsampling="4:2:2", progressive=True,
             restart_marker_blocks=2), encode(smooth_image(33, 24, 24, 1), quality=90, progressive=True)]  # fmt: skip
    decoded = refused = 0
    for it in range(1500):
        b = bytearray(seeds[it % 3])
        mode = it % 4
        if mode == 0:
            for _ in range(rng.integers(1, 6)):
                b[rng.integers(2, len(b))] = rng.integers(0, 256)
        elif mode == 1:
            b = b[: rng.integers(4, len(b))]
        elif mode == 2:
            i = rng.integers(2, len(b) - 4)
            b[i : i + 2] = bytes([0xFF, int(rng.integers(0xC0, 0xFF))])
        else:
            i = rng.integers(2, len(b) - 10)
            del b[i : i + rng.integers(1, 10)]
        data = bytes(b)
        info = (C.c_int * 24)()
        if lib.imcui_hip_jpeg_info(data, len(data), info) != 0:
            refused += 1
            continue
        n = lib.imcui_hip_jpeg_coef_count(info)
        if n > 50_000_000:
            refused += 1
            continue
        coef = np.zeros(n + 64, np.int16)
        coef[n:] = 12345
        qt = np.zeros(192, np.uint16)
        rc = lib.imcui_hip_jpeg_entropy_decode(data, len(data), coef.ctypes.data, qt.ctypes.data)
        assert (coef[n:] == 12345).all(), "the decoder wrote past its buffer"
        assert rc in (0, -1, -4)
        decoded += rc == 0
        refused += rc != 0
    assert decoded > 100 and refused > 200, (decoded, refused)
