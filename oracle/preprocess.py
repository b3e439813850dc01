"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's host preprocessing for images that need
no resize (imcui/hloc/extract_features.py:120-160): cv2.cvtColor(image, COLOR_RGB2GRAY) on uint8,
astype(float32), / 255.0.

**Parity unpinned**: cv2 is not installable here (no network) and the reference holds no fixture for this step.
The grey conversion follows OpenCV 4.x's published 8-bit fixed-point kernel (modules/imgproc/src/color_yuv.simd.hpp,
RGB2Gray<uchar>: RY15 = 9798, GY15 = 19235, BY15 = 3735, gray_shift = 15, round-half-up descale).
"""
import numpy as np


def rgb_to_gray_u8(rgb: np.ndarray) -> np.ndarray:
    """uint8 [...,3] (R,G,B) -> uint8 [...]"""
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
    return ((9798 * r + 19235 * g + 3735 * b + (1 << 14)) >> 15).astype(np.uint8)


def preprocess_gray(rgb: np.ndarray) -> np.ndarray:
    """uint8 [B,H,W,3] -> float32 [B,1,H,W] = gray.astype(float32) / 255.0 (float32 division, as numpy does it)"""
    gray = rgb_to_gray_u8(rgb).astype(np.float32)
    return (gray / np.float32(255.0))[:, None]


# ----------------------------------------------------------------------------- area resize (cv2.INTER_AREA)
def _area_table(ssize: int, dsize: int):
    """OpenCV's `computeResizeAreaTab` (modules/imgproc/src/resize.cpp): for every destination index the source
    indices it covers and their weights = covered length / cell width, as float32 (the table holds floats).
    Returns a list (per destination index) of (source indices int64[], weights float32[]) in table order."""
    scale = ssize / dsize
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        idx, wgt = [], []
        if sx1 - fsx1 > 1e-3:
            idx.append(sx1 - 1)
            wgt.append((sx1 - fsx1) / cell)
        for sx in range(sx1, sx2):
            idx.append(sx)
            wgt.append(1.0 / cell)
        if fsx2 - sx2 > 1e-3:
            idx.append(sx2)
            wgt.append(min(min(fsx2 - sx2, 1.0), cell) / cell)
        tab.append((np.asarray(idx, dtype=np.int64), np.asarray(wgt, dtype=np.float32)))
    return tab


def area_resize_f32(image: np.ndarray, size) -> np.ndarray:
    """`cv2.resize(image.astype(float32), (w, h), interpolation=cv2.INTER_AREA)` for a SHRINKING resize of a 2-D image
    (imcui/hloc/extract_features.py:26-32,124-133: `resize_image(..., "cv2_area")`; the reference switches to
    INTER_LINEAR when either side grows -- not restated here).  float32 arithmetic in OpenCV's order: a source row is
    first reduced horizontally (sum of value * alpha in table order), row results are then accumulated with the
    vertical weights (first row `beta * buf`, later rows `+= beta * buf`).  Integer scale factors take OpenCV's
    `resizeAreaFast` path: plain sum of the block (row-major) times float32(1 / area).

    **Parity unpinned** (cv2 is absent): follows the published algorithm of OpenCV 4.x `resize.cpp`; whether the
    library build contracts the multiply-adds into FMAs cannot be checked here."""
    w, h = int(size[0]), int(size[1])
    src = np.ascontiguousarray(image, dtype=np.float32)
    sh, sw = src.shape
    assert w <= sw and h <= sh, "area_resize_f32 restates the shrinking path only"
    if sw % w == 0 and sh % h == 0:
        fx, fy = sw // w, sh // h
        acc = np.zeros((h, w), dtype=np.float32)
        for ky in range(fy):
            for kx in range(fx):
                acc = acc + src[ky::fy, kx::fx][:h, :w]
        return acc * np.float32(1.0 / (fx * fy))
    xtab, ytab = _area_table(sw, w), _area_table(sh, h)
    kmax = max(len(i) for i, _ in xtab)
    hbuf = np.zeros((sh, w), dtype=np.float32)
    for k in range(kmax):  # k-th table entry of every destination column at once (same per-column order)
        cols = [dx for dx in range(w) if len(xtab[dx][0]) > k]
        si = np.asarray([xtab[dx][0][k] for dx in cols])
        al = np.asarray([xtab[dx][1][k] for dx in cols], dtype=np.float32)
        hbuf[:, cols] = hbuf[:, cols] + src[:, si] * al[None, :]
    out = np.zeros((h, w), dtype=np.float32)
    for dy in range(h):
        idx, beta = ytab[dy]
        acc = hbuf[idx[0]] * beta[0]
        for k in range(1, len(idx)):
            acc = acc + hbuf[idx[k]] * beta[k]
        out[dy] = acc
    return out


# ----------------------------------------------------------------------------- growing resize (cv2.INTER_LINEAR)
def _linear_table(ssize: int, dsize: int):
    """OpenCV `resize.cpp` (`resizeGeneric_` set-up for INTER_LINEAR, float images): per destination index the left source
    index and the float32 weight of the RIGHT tap; half-pixel centres, `fx = (float)((dx + 0.5) * scale - 0.5)` with a double
    `scale = 1 / (dsize / ssize)`, `sx = floor(fx)`, `fx -= sx`; a tap left of the image is pinned to pixel 0 with weight 0,
    a tap at / beyond the last pixel to the last pixel with weight 0 (horizontal rule).  `clip_rows=True` gives the vertical
    rule instead: the weight is kept and both row indices are clipped to the image."""
    inv = dsize / ssize
    scale = 1.0 / inv
    idx = np.zeros(dsize, dtype=np.int64)
    w1 = np.zeros(dsize, dtype=np.float32)
    for d in range(dsize):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        idx[d], w1[d] = s, f
    return idx, w1


def linear_resize_f32(image: np.ndarray, size) -> np.ndarray:
    """`cv2.resize(image.astype(float32), (w, h), interpolation=cv2.INTER_LINEAR)` for a 2-D image: what
    imcui/hloc/extract_features.py:26-32 (`resize_image`, "cv2_area") really runs when either side GROWS.  float32
    arithmetic in OpenCV's order: horizontal pass `S[sx] * (1 - fx) + S[sx + 1] * fx` per needed source row, then
    `row0 * (1 - fy) + row1 * fy` (multiply, multiply, add: no fused multiply-add).

    **Parity unpinned** (cv2 is absent): follows the published algorithm of OpenCV 4.x `resize.cpp` (HResizeLinear /
    VResizeLinear with float weights); whether the library build contracts the two products into an FMA cannot be checked."""
    w, h = int(size[0]), int(size[1])
    src = np.ascontiguousarray(image, dtype=np.float32)
    sh, sw = src.shape
    xi, xw = _linear_table(sw, w)
    yi, yw = _linear_table(sh, h)
    # horizontal rule: out-of-range taps are pinned with weight 0
    x0, x1, a1 = xi.copy(), xi + 1, xw.copy()
    left = xi < 0
    x0[left], x1[left], a1[left] = 0, 0, 0.0
    right = xi >= sw - 1
    x0[right], x1[right], a1[right] = sw - 1, sw - 1, 0.0
    a0 = (np.float32(1.0) - a1).astype(np.float32)
    hbuf = src[:, x0] * a0[None, :] + src[:, x1] * a1[None, :]
    # vertical rule: weights kept, row indices clipped
    y0, y1 = np.clip(yi, 0, sh - 1), np.clip(yi + 1, 0, sh - 1)
    b1 = yw
    b0 = (np.float32(1.0) - b1).astype(np.float32)
    return (hbuf[y0] * b0[:, None] + hbuf[y1] * b1[:, None]).astype(np.float32)


def resize_image_cv2_area(image_f32: np.ndarray, size) -> np.ndarray:
    """`resize_image(image, size, "cv2_area")` (extract_features.py:26-32): INTER_AREA, or INTER_LINEAR as soon as a side grows."""
    sh, sw = image_f32.shape
    if sw < size[0] or sh < size[1]:
        return linear_resize_f32(image_f32, size)
    return area_resize_f32(image_f32, size)


# ----------------------------------------------------------------------------- dfactor resize (torchvision F.resize, antialias=True)
def aa_table(in_size: int, out_size: int):
    """ATen `_compute_indices_weights_aa` for the bilinear (triangle) filter, align_corners=False, float32 throughout:
    per output index the first source index and the normalised float32 weights.  What `torchvision.transforms.functional.resize(
    image, size, antialias=True)` = `torch.nn.functional.interpolate(mode="bilinear", antialias=True)` evaluates for the float
    image of extract_features.py:142-148 / match_dense.py:182."""
    f = np.float32
    scale = f(in_size) / f(out_size)
    support = f(1.0) * scale if scale >= 1.0 else f(1.0)
    invscale = f(1.0) / scale if scale >= 1.0 else f(1.0)
    tab = []
    for i in range(out_size):
        center = scale * (f(i) + f(0.5))
        xmin = max(int(center - support + f(0.5)), 0)
        xsize = min(int(center + support + f(0.5)), in_size) - xmin
        ws, tot = [], f(0)
        for j in range(xsize):
            x = (f(j + xmin) - center + f(0.5)) * invscale
            wv = f(1.0) - abs(x) if abs(x) < 1.0 else f(0)
            ws.append(f(wv))
            tot = f(tot + f(wv))
        tab.append((xmin, np.asarray([f(wv / tot) for wv in ws], dtype=np.float32)))
    return tab


def _fma32(a, b, c):
    """float32 fused multiply-add: the product of two float32 is exact in float64, one rounding to float64 then to float32."""
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(np.float32)


def aa_resize_f32(image: np.ndarray, size_hw) -> np.ndarray:
    """`F.interpolate(image[None, None], size=(h, w), mode="bilinear", align_corners=False, antialias=True)[0, 0]` on CPU,
    restated: the separable kernel runs along the width first, then along the height; each output is `src[0] * w[0]`
    followed by fused multiply-adds in tap order (ATen `basic_loop_aa_*`).  **Pinned**: tests/test_oracle_preprocess.py
    requires bit-equality with torch itself (the arithmetic the reference runs through torchvision) on a range of sizes."""
    h, w = int(size_hw[0]), int(size_hw[1])
    src = np.ascontiguousarray(image, dtype=np.float32)
    sh, sw = src.shape
    if (h, w) == (sh, sw):
        return src.copy()  # torchvision returns the image unchanged when the size already matches

    def one_pass(a, tab):  # along the last axis
        out = np.zeros((a.shape[0], len(tab)), dtype=np.float32)
        for i, (xmin, ws) in enumerate(tab):
            t = a[:, xmin] * ws[0]
            for j in range(1, len(ws)):
                t = _fma32(a[:, xmin + j], ws[j], t)
            out[:, i] = t
        return out

    t = one_pass(src, aa_table(sw, w))
    return np.ascontiguousarray(one_pass(np.ascontiguousarray(t.T), aa_table(sh, h)).T)
