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
