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
