"""Deterministic homography estimation for the AUC parity leg.

TEST INFRASTRUCTURE ONLY.  The reference verifies matches on the host with cv2 MAGSAC
(`imcui/ui/utils.py:532-610`); cv2 is not installable here and RANSAC stays host-side by north_star, so
the parity leg uses this seeded numpy DLT-RANSAC on both the oracle's and the HIP path's matches:
identical matches must give identical homographies and therefore identical corner-error AUC.
"""
from __future__ import annotations

import numpy as np


def _normalise(pts: np.ndarray):
    c = pts.mean(0)
    s = np.sqrt(2.0) / max(np.sqrt(((pts - c) ** 2).sum(1)).mean(), 1e-12)
    t = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
    return (pts - c) * s, t


def dlt_homography(p0: np.ndarray, p1: np.ndarray):
    """Least-squares H (p1 ~ H p0) from >= 4 correspondences, Hartley-normalised DLT; None if degenerate."""
    if len(p0) < 4:
        return None
    a0, t0 = _normalise(p0.astype(np.float64))
    a1, t1 = _normalise(p1.astype(np.float64))
    rows = np.zeros((2 * len(a0), 9))
    x, y, u, v = a0[:, 0], a0[:, 1], a1[:, 0], a1[:, 1]
    rows[0::2] = np.stack([-x, -y, -np.ones_like(x), 0 * x, 0 * x, 0 * x, u * x, u * y, u], 1)
    rows[1::2] = np.stack([0 * x, 0 * x, 0 * x, -x, -y, -np.ones_like(x), v * x, v * y, v], 1)
    try:
        _, _, vh = np.linalg.svd(rows)
    except np.linalg.LinAlgError:
        return None
    h = np.linalg.inv(t1) @ vh[-1].reshape(3, 3) @ t0
    if abs(h[2, 2]) < 1e-12 or not np.isfinite(h).all():
        return None
    return h / h[2, 2]


def project(h: np.ndarray, pts: np.ndarray) -> np.ndarray:
    q = np.concatenate([pts, np.ones((len(pts), 1))], 1) @ h.T
    return q[:, :2] / np.where(np.abs(q[:, 2:3]) < 1e-12, 1e-12, q[:, 2:3])


def ransac_homography(p0: np.ndarray, p1: np.ndarray, thresh: float = 3.0, iters: int = 500, seed: int = 0):
    """Seeded 4-point RANSAC + refit on the inliers of the best hypothesis.  Returns (H or None, inlier mask).
    Deterministic for a given input: the sampler is a fixed-seed PCG stream, ties keep the first hypothesis."""
    n = len(p0)
    if n < 4:
        return None, np.zeros(n, bool)
    rng = np.random.Generator(np.random.PCG64(seed))
    best, best_cnt = None, -1
    for _ in range(iters):
        idx = rng.choice(n, 4, replace=False)
        h = dlt_homography(p0[idx], p1[idx])
        if h is None:
            continue
        inl = np.linalg.norm(project(h, p0) - p1, axis=1) < thresh
        if inl.sum() > best_cnt:
            best, best_cnt = inl, int(inl.sum())
    if best is None or best_cnt < 4:
        return None, np.zeros(n, bool)
    h = dlt_homography(p0[best], p1[best])
    if h is None:
        return None, best
    inl = np.linalg.norm(project(h, p0) - p1, axis=1) < thresh
    if inl.sum() >= 4:
        h2 = dlt_homography(p0[inl], p1[inl])
        if h2 is not None:
            h = h2
    return h, inl


def corner_error(h_est, h_gt: np.ndarray, width: int, height: int) -> float:
    """Mean distance of the four image corners mapped by the estimate and by the ground truth."""
    if h_est is None:
        return float("inf")
    c = np.array([[0, 0], [width - 1, 0], [width - 1, height - 1], [0, height - 1]], np.float64)
    return float(np.linalg.norm(project(h_est, c) - project(h_gt.astype(np.float64), c), axis=1).mean())


def error_auc(errors, thresholds=(3.0, 5.0, 10.0)):
    """Area under the recall-vs-error curve up to each threshold, normalised to [0, 1] (hloc convention)."""
    e = np.sort(np.asarray(list(errors), np.float64))
    recall = (np.arange(len(e)) + 1) / max(len(e), 1)
    e = np.concatenate([[0.0], e])
    recall = np.concatenate([[0.0], recall])
    out = []
    for t in thresholds:
        last = np.searchsorted(e, t)
        x = np.concatenate([e[:last], [t]])
        y = np.concatenate([recall[:last], [recall[last - 1]]])
        out.append(float(np.trapezoid(y, x) / t))
    return out
