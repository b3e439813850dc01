"""oracle/geometry.py (the CPU restatement of the batched RANSAC of csrc/geometry.hip) against ground-truth geometry: a known
homography / a known two-view geometry with outliers is recovered, the sampler is a pure function of its counters, the sequential
stopping rule behaves.  PARITY UNPINNED with respect to cv2's USAC_MAGSAC (imcui/ui/utils.py:424-456): cv2 is not installed."""
import numpy as np

from oracle import geometry as og


def homography_scene(seed, n=400, outliers=0.4, noise=0.3):
    g = np.random.default_rng(seed)
    H = np.array([[1.05, 0.08, 12.0], [-0.06, 0.97, -7.0], [1.2e-4, -0.8e-4, 1.0]])
    p0 = g.uniform([0, 0], [640, 480], (n, 2))
    q = np.concatenate([p0, np.ones((n, 1))], 1) @ H.T
    p1 = q[:, :2] / q[:, 2:] + g.normal(0, noise, (n, 2))
    bad = g.random(n) < outliers
    p1[bad] = g.uniform([0, 0], [640, 480], (int(bad.sum()), 2))
    return p0.astype(np.float32), p1.astype(np.float32), H, ~bad


def two_view_scene(seed, n=500, outliers=0.3, noise=0.3):
    g = np.random.default_rng(seed)
    X = np.concatenate([g.uniform(-2, 2, (n, 2)), g.uniform(4, 9, (n, 1))], 1)
    K = np.array([[520.0, 0, 320], [0, 520.0, 240], [0, 0, 1]])
    ang = 0.2
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.8, 0.05, 0.1])
    x0 = X @ K.T
    x1 = (X @ R.T + t) @ K.T
    p0, p1 = x0[:, :2] / x0[:, 2:], x1[:, :2] / x1[:, 2:]
    p0, p1 = p0 + g.normal(0, noise, p0.shape), p1 + g.normal(0, noise, p1.shape)
    bad = g.random(n) < outliers
    p1[bad] = g.uniform([0, 0], [640, 480], (int(bad.sum()), 2))
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
    return p0.astype(np.float32), p1.astype(np.float32), F / np.linalg.norm(F), ~bad


def test_sampler_is_a_pure_function_and_samples_are_distinct():
    a = [og.sample(7, 3, k, 8, 50) for k in range(200)]
    assert a == [og.sample(7, 3, k, 8, 50) for k in range(200)]
    assert all(len(set(s)) == 8 and min(s) >= 0 and max(s) < 50 for s in a)
    assert og.sample(7, 3, 0, 4, 4) != [] and sorted(og.sample(7, 3, 5, 4, 4)) == [0, 1, 2, 3]  # n = m: every index exactly once
    flat = np.array(a).ravel()
    assert np.bincount(flat, minlength=50).min() > 5  # roughly uniform


def test_homography_recovered_with_outliers():
    p0, p1, H, good = homography_scene(1)
    M, mask, info = og.ransac(p0, p1, 0, 3.0, 0.9999, 2000, seed=0)
    assert M is not None and abs(M[2, 2] - 1.0) < 1e-12
    c = np.array([[0, 0, 1], [639, 0, 1], [639, 479, 1], [0, 479, 1.0]])
    pe, pg = c @ M.T, c @ H.T
    assert np.abs(pe[:, :2] / pe[:, 2:] - pg[:, :2] / pg[:, 2:]).max() < 0.5
    assert (mask & good).sum() > 0.97 * good.sum() and (mask & ~good).sum() <= 3
    assert 0 < info["used"] < 2000  # the stopping rule ended the run early (60 % inliers, 4-point samples)


def test_fundamental_recovered_with_outliers():
    p0, p1, F, good = two_view_scene(2)
    M, mask, info = og.ransac(p0, p1, 1, 2.0, 0.999, 3000, seed=1)
    assert M is not None and abs(np.linalg.norm(M) - 1.0) < 1e-12 and abs(np.linalg.det(M)) < 1e-12
    # (entries of a fundamental matrix are not comparable one by one -- the estimate is judged by the geometry it implies)
    assert np.sqrt(og.errors2(F, p0[good], p1[good], 1)).mean() < 1.0  # the ground truth itself, for scale
    assert (mask & good).sum() > 0.95 * good.sum()
    # epipolar residuals of the true inliers are small
    assert np.sqrt(og.errors2(M, p0[good], p1[good], 1)).mean() < 1.0


def test_degenerate_inputs():
    p0, p1, _, _ = homography_scene(3, n=3)
    assert og.ransac(p0, p1, 0, 3.0, 0.99, 100)[0] is None  # fewer matches than a minimal sample
    line = np.stack([np.arange(20.0), 2 * np.arange(20.0)], 1).astype(np.float32)
    M, mask, _ = og.ransac(line, line + 1, 0, 3.0, 0.99, 50)  # collinear points: every sample is degenerate
    assert M is None and not mask.any()


def test_low_inlier_ratio_does_not_stop_the_run():
    """with 2500 matches, a tight threshold and 97 % outliers the first hypotheses have an inlier ratio w with
    w^8 < 2^-53, where log(1 - w^8) is exactly 0: the iteration bound must read as "no bound" (log1p), not as "done" -- the run goes
    on and still finds the geometry.  (The oracle used to raise ZeroDivisionError here, the device stopped at the first hypothesis.)"""
    p0, p1, F, good = two_view_scene(21, n=2500, outliers=0.90, noise=0.05)
    M, mask, info = og.ransac(p0, p1, 1, 0.25, 0.999, 400, seed=3)
    assert info["used"] > 1  # did not stop at the first valid hypothesis
    # the bound itself: 8 inliers of 2500 -> wm ~ 1e-20
    import math

    wm = (8 / 2500) ** 8
    assert math.log(1.0 - wm) == 0.0 and math.log1p(-wm) < 0.0
