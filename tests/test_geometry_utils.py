"""The AUC leg's own tool: the numpy DLT-RANSAC must recover a known homography and be deterministic."""
import numpy as np

from geometry_utils import corner_error, dlt_homography, error_auc, project, ransac_homography


def _h():
    return np.array([[1.02, 0.03, 5.0], [-0.02, 0.98, -3.0], [1e-5, -2e-5, 1.0]])


def test_dlt_exact_on_clean_points():
    rng = np.random.default_rng(0)
    p0 = rng.uniform(0, 600, (50, 2))
    h = dlt_homography(p0, project(_h(), p0))
    assert np.abs(h - _h()).max() < 1e-8


def test_ransac_recovers_h_with_outliers_and_is_deterministic():
    rng = np.random.default_rng(1)
    p0 = rng.uniform(0, 600, (300, 2))
    p1 = project(_h(), p0) + rng.normal(0, 0.3, (300, 2))
    p1[:120] = rng.uniform(0, 600, (120, 2))  # 40 % outliers
    h_a, inl_a = ransac_homography(p0, p1)
    h_b, inl_b = ransac_homography(p0, p1)
    assert np.array_equal(h_a, h_b) and np.array_equal(inl_a, inl_b)
    assert inl_a[120:].mean() > 0.95 and inl_a[:120].mean() < 0.1
    assert corner_error(h_a, _h(), 640, 480) < 0.5


def test_degenerate_inputs_and_auc():
    h, inl = ransac_homography(np.zeros((3, 2)), np.zeros((3, 2)))
    assert h is None and inl.sum() == 0 and corner_error(h, _h(), 640, 480) == float("inf")
    auc = error_auc([0.0, 0.0, 100.0, float("inf")])
    assert all(abs(a - 0.5) < 1e-9 for a in auc)
    assert abs(error_auc([1.5], thresholds=(3.0,))[0] - 0.75) < 1e-12  # (0,0) -> (1.5,1) -> (3,1)
