"""Synthetic image pairs at the model boundary.

float32 images in [0,1], shape [B,1,H,W], fed post-preprocess so cv2 is not
needed: `img0` = 6 octaves of bilinearly upsampled U(0,1) grids + random bright /
dark 3x3 blobs; `img1` = `img0` warped by a random homography (corner jitter
<= 48 px) + N(0, 0.01) noise, so a ground-truth H is known for inlier checks.
Pure torch; runs on any device.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _band_limited_noise(g: torch.Generator, h: int, w: int) -> torch.Tensor:
    img = torch.zeros(1, 1, h, w)
    amp = 1.0
    total = 0.0
    cells_w, cells_h = 4, 3
    for _ in range(6):
        grid = torch.rand(1, 1, cells_h, cells_w, generator=g)
        img += amp * F.interpolate(grid, size=(h, w), mode="bilinear", align_corners=False)
        total += amp
        amp *= 0.7
        cells_w *= 2
        cells_h *= 2
    return img / total


def _add_blobs(g: torch.Generator, img: torch.Tensor, n_blobs: int) -> torch.Tensor:
    _, _, h, w = img.shape
    ys = torch.randint(2, h - 2, (n_blobs,), generator=g)
    xs = torch.randint(2, w - 2, (n_blobs,), generator=g)
    sign = (torch.rand(n_blobs, generator=g) > 0.5).float() * 2 - 1
    out = img.clone()
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            wgt = 0.5 if (dy or dx) else 0.9
            out[0, 0, ys + dy, xs + dx] += sign * wgt * 0.6
    return out.clamp_(0, 1)


def random_homography(g: torch.Generator, h: int, w: int, jitter: float = 48.0) -> torch.Tensor:
    """3x3 H mapping img0 pixel coords -> img1 pixel coords (DLT on jittered corners)."""
    src = torch.tensor([[0.0, 0.0], [w - 1.0, 0.0], [w - 1.0, h - 1.0], [0.0, h - 1.0]], dtype=torch.float64)
    dst = src + (torch.rand(4, 2, generator=g, dtype=torch.float64) * 2 - 1) * jitter
    rows = []
    for (x, y), (u, v) in zip(src.tolist(), dst.tolist()):
        rows.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u])
        rows.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
    a = torch.tensor(rows, dtype=torch.float64)
    _, _, vh = torch.linalg.svd(a)
    hm = vh[-1].reshape(3, 3)
    return (hm / hm[2, 2]).to(torch.float32)


def warp_image(img: torch.Tensor, hmat: torch.Tensor) -> torch.Tensor:
    """img1(p1) = img0(H^-1 p1), bilinear, zero padding."""
    _, _, h, w = img.shape
    hinv = torch.linalg.inv(hmat.double())
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w, dtype=torch.float64), indexing="ij")
    pts = torch.stack([xs, ys, torch.ones_like(xs)], -1) @ hinv.T
    px = pts[..., 0] / pts[..., 2]
    py = pts[..., 1] / pts[..., 2]
    grid = torch.stack([px / (w - 1) * 2 - 1, py / (h - 1) * 2 - 1], -1).float()[None]
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


def make_pair(seed: int, h: int = 480, w: int = 640, n_blobs: int = 2000):
    """Returns (img0 [1,1,H,W], img1 [1,1,H,W], H [3,3]) on CPU."""
    g = torch.Generator().manual_seed(seed)
    img0 = _add_blobs(g, _band_limited_noise(g, h, w), n_blobs)
    hmat = random_homography(g, h, w)
    img1 = warp_image(img0, hmat)
    img1 = (img1 + 0.01 * torch.randn(img1.shape, generator=g)).clamp_(0, 1)
    return img0.contiguous(), img1.contiguous(), hmat


def make_pair_batch(seed: int, batch: int, h: int = 480, w: int = 640, n_blobs: int = 2000, distinct: int | None = None):
    """[B,1,H,W] x2 + [B,3,3].  `distinct` < batch tiles a few generated pairs (cheap host prep)."""
    distinct = batch if distinct is None else max(1, min(distinct, batch))
    pairs = [make_pair(seed * 100003 + i, h, w, n_blobs) for i in range(distinct)]
    idx = [i % distinct for i in range(batch)]
    img0 = torch.cat([pairs[i][0] for i in idx], 0)
    img1 = torch.cat([pairs[i][1] for i in idx], 0)
    hm = torch.stack([pairs[i][2] for i in idx], 0)
    return img0, img1, hm


def make_shifted_pair(seed: int, h: int = 480, w: int = 640, shift=(16, 8), n_blobs: int = 2000, noise: float = 0.0):
    """Two crops of one synthetic scene: img1(x, y) = img0(x + dx, y + dy).  Returns (img0, img1 [1,1,H,W], (dx, dy)).
    A match (p0, p1) of the pair satisfies p0 - p1 = (dx, dy).  Translations by multiples of 8 px keep the 1/8 grids of
    the dense matchers aligned, which is what lets random-weight networks produce many confident matches."""
    g = torch.Generator().manual_seed(seed)
    dx, dy = shift
    m = 8 * ((max(abs(dx), abs(dy)) + 7) // 8)
    base = _add_blobs(g, _band_limited_noise(g, h + 2 * m, w + 2 * m), n_blobs * (h + 2 * m) * (w + 2 * m) // (h * w))
    img0 = base[:, :, m : m + h, m : m + w]
    img1 = base[:, :, m + dy : m + dy + h, m + dx : m + dx + w]
    if noise > 0:
        img1 = (img1 + noise * torch.randn(img1.shape, generator=g)).clamp(0, 1)
    return img0.contiguous(), img1.contiguous(), (dx, dy)
