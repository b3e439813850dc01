"""Oracle of the device preprocessing step: known answers of the fixed-point grey formula."""
import numpy as np

from oracle.preprocess import preprocess_gray, rgb_to_gray_u8


def test_gray_known_answers():
    px = np.array([[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [12, 200, 77], [1, 1, 1]], np.uint8)
    # 9798 + 19235 + 3735 = 32768: white stays 255, grey levels are preserved; primaries = round(255 * 0.299 / 0.587 / 0.114)
    assert rgb_to_gray_u8(px).tolist() == [0, 255, 76, 150, 29, 130, 1]
    v = np.arange(256, dtype=np.uint8)
    assert np.array_equal(rgb_to_gray_u8(np.stack([v, v, v], -1)), v)


def test_preprocess_layout_and_range():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (2, 6, 8, 3), dtype=np.uint8)
    out = preprocess_gray(img)
    assert out.shape == (2, 1, 6, 8) and out.dtype == np.float32
    assert out.min() >= 0.0 and out.max() <= 1.0
    assert out[1, 0, 3, 5] == np.float32(rgb_to_gray_u8(img[1, 3, 5])) / np.float32(255.0)
