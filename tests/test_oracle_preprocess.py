"""The host restatements of the reference's preprocessing (oracle/preprocess.py) against what CAN be run here.

* the dfactor resize (`torchvision F.resize(..., antialias=True)`, extract_features.py:142-148, match_dense.py:182) is ATen's
  anti-aliased bilinear kernel: the restatement must equal `torch.nn.functional.interpolate(antialias=True)` BIT FOR BIT --
  this pins the tap tables and the accumulation order the HIP kernel copies;
* the cv2 paths (INTER_AREA / INTER_LINEAR) are unpinned (cv2 is absent): property tests only.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import preprocess as P


@pytest.mark.parametrize("H,W,h,w", [(487, 653, 480, 648), (480, 640, 480, 640), (300, 517, 296, 512), (96, 100, 48, 37), (33, 47, 32, 40),
                                      (512, 384, 512, 384), (517, 389, 512, 384), (1030, 771, 1024, 768), (31, 31, 64, 64)])
def test_aa_resize_equals_torch_bit_for_bit(H, W, h, w):
    g = torch.Generator().manual_seed(H * 7 + W)
    img = torch.rand(H, W, generator=g)
    ref = F.interpolate(img[None, None], size=(h, w), mode="bilinear", align_corners=False, antialias=True)[0, 0].numpy()
    got = P.aa_resize_f32(img.numpy(), (h, w))
    assert got.dtype == np.float32 and got.shape == ref.shape
    assert np.array_equal(got, ref), np.abs(got - ref).max()


def test_linear_resize_properties():
    """cv2.INTER_LINEAR restatement (unpinned): a constant image stays constant to round-off, an affine ramp is reproduced in
    the interior (bilinear interpolation is exact for affine functions), doubling maps pixel centres (dx + 0.5) / 2 - 0.5."""
    c = np.full((12, 17), 93.0, dtype=np.float32)
    assert np.abs(P.linear_resize_f32(c, (40, 29)) - 93.0).max() < 1e-4
    yy, xx = np.mgrid[0:20, 0:30].astype(np.float32)
    ramp = (2.0 * xx + 3.0 * yy + 1.0).astype(np.float32)
    out = P.linear_resize_f32(ramp, (60, 40))  # x2 in both directions
    X = (np.arange(60) + 0.5) / 2.0 - 0.5
    Y = (np.arange(40) + 0.5) / 2.0 - 0.5
    want = 2.0 * X[None, :] + 3.0 * Y[:, None] + 1.0
    inner = (slice(2, -2), slice(2, -2))
    assert np.abs(out[inner] - want[inner]).max() < 1e-3
    # borders clamp: the first output column of a x2 resize sits left of pixel 0's centre and equals pixel 0 horizontally
    assert np.allclose(out[:, 0], P.linear_resize_f32(ramp, (60, 40))[:, 0])
    assert np.isclose(out[0, 0], ramp[0, 0], atol=1e-4)


def test_resize_image_switches_to_linear_when_a_side_grows():
    """extract_features.py:29-31: INTER_AREA silently becomes INTER_LINEAR when w < size[0] or h < size[1] -- also when the
    other side shrinks."""
    g = np.random.default_rng(3)
    img = (g.random((240, 320)) * 255).astype(np.float32)
    grow = P.resize_image_cv2_area(img, (640, 480))
    assert grow.shape == (480, 640) and np.array_equal(grow, P.linear_resize_f32(img, (640, 480)))
    mixed = P.resize_image_cv2_area(img, (400, 200))  # wider, lower
    assert np.array_equal(mixed, P.linear_resize_f32(img, (400, 200)))
    shrink = P.resize_image_cv2_area(img, (160, 120))
    assert np.array_equal(shrink, P.area_resize_f32(img, (160, 120)))
