"""Batched RANSAC on the device (csrc/geometry.hip through the C ABI) against its CPU restatement oracle/geometry.py and against
ground-truth geometry (GPU box only).  PARITY UNPINNED with respect to cv2's USAC_MAGSAC (imcui/ui/utils.py:424-456)."""
import numpy as np
import pytest
import torch

from oracle import geometry as og
from test_geometry_cpu import homography_scene, two_view_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(scenes):
    n = max(len(s[0]) for s in scenes)
    p0, p1 = torch.zeros(len(scenes), n, 2), torch.zeros(len(scenes), n, 2)
    for b, s in enumerate(scenes):
        p0[b, : len(s[0])], p1[b, : len(s[0])] = torch.from_numpy(s[0]), torch.from_numpy(s[1])
    return p0.to(DEV), p1.to(DEV), torch.tensor([len(s[0]) for s in scenes], dtype=torch.int32, device=DEV)


@pytest.mark.parametrize("geometry_type,thr,conf,iters", [("Homography", 3.0, 0.9999, 2000), ("Fundamental", 2.0, 0.999, 3000), ("Homography", 8.0, 0.9999, 10000)])
def test_batched_ransac_equals_the_oracle(geometry_type, thr, conf, iters):
    """A ragged batch (400, 900, 37, 3 and 2048 matches; one pair below the minimal sample) in ONE call: per pair the winning hypothesis
    index and the number of hypotheses the stopping rule consumed equal the oracle's, the model agrees to 1e-7, the inlier masks are
    equal except for matches whose error sits within 1e-6 (relative) of the threshold."""
    from imcui_hip.geometry import ransac_batched

    geo = 0 if geometry_type == "Homography" else 1
    make = homography_scene if geo == 0 else two_view_scene
    scenes = [make(10, 400), make(11, 900, 0.5), make(12, 37, 0.2), make(13, 3), make(14, 2048, 0.6)]
    p0, p1, counts = _batch(scenes)
    out = ransac_batched(p0, p1, counts, geometry_type, thr, conf, iters, seed=5)
    torch.cuda.synchronize()
    out2 = ransac_batched(p0, p1, counts, geometry_type, thr, conf, iters, seed=5)
    assert all(torch.equal(out[k], out2[k]) for k in out)  # a pure function of (inputs, seed)
    for b, s in enumerate(scenes):
        M, mask, info = og.ransac(s[0], s[1], geo, thr, conf, iters, seed=5, pair_index=b)
        n = len(s[0])
        got_mask = out["mask"][b, :n].cpu().numpy()
        assert not out["mask"][b, n:].any()
        if M is None:
            assert not bool(out["ok"][b]) and not got_mask.any()
            continue
        assert bool(out["ok"][b])
        assert int(out["iterations"][b]) == info["used"], (b, int(out["iterations"][b]), info["used"])
        G = out["model"][b].cpu().numpy()
        assert np.abs(G - M).max() < 1e-7 * max(1.0, np.abs(M).max()), (b, np.abs(G - M).max())
        diff = np.nonzero(got_mask != mask)[0]
        if len(diff):
            e = np.sqrt(og.errors2(M, s[0][diff], s[1][diff], geo))
            assert (np.abs(e - thr) < 1e-6 * thr).all(), (b, e)
        assert abs(int(out["num_inliers"][b]) - info["inliers"]) <= len(diff)
        # ground truth: the true inliers are found
        good = s[3]
        assert (got_mask & good).sum() > 0.9 * good.sum()


def test_reference_call_surface_and_pipeline_hook():
    """`proc_ransac_matches` / `compute_geometry` / `filter_matches` with the signatures of imcui/ui/utils.py for the method "HIP_RANSAC",
    and `verify_matches_batched` on fixed-stride pipeline outputs."""
    from imcui_hip import geometry as hg

    p0, p1, H, good = homography_scene(21, 600, 0.3)
    M, mask = hg.proc_ransac_matches(p0, p1, "HIP_RANSAC", 3.0, 0.9999, 2000, "Homography")
    assert M.shape == (3, 3) and mask.dtype == bool and (mask & good).sum() > 0.95 * good.sum()
    assert hg.proc_ransac_matches(p0[:3], p1[:3], "HIP_RANSAC", 3.0, 0.99, 100, "Homography") == (None, None)
    with pytest.raises(NotImplementedError):
        hg.proc_ransac_matches(p0, p1, "CV2_USAC_MAGSAC")
    pred = {"mkeypoints0_orig": p0, "mkeypoints1_orig": p1, "mconf": np.linspace(0, 1, len(p0)).astype(np.float32), "image0_orig": np.zeros((480, 640, 3), np.uint8)}
    out = hg.filter_matches(dict(pred), ransac_reproj_threshold=3.0)
    assert out["H"].shape == (3, 3) and len(out["mmkeypoints0_orig"]) == len(out["mmconf"]) > 0.9 * good.sum()
    assert set(out["geom_info"]) >= {"Fundamental", "Homography"} and "mask_h" not in out["geom_info"]
    assert hg.filter_matches({"mkeypoints0_orig": p0[:2], "mkeypoints1_orig": p1[:2], "mconf": np.ones(2)})["H"] is None
    # fixed-stride pipeline outputs: pair 0 = a permuted correspondence table with unmatched rows, pair 1 = no matches at all
    K = 700
    k0, k1 = torch.zeros(2, K, 2), torch.zeros(2, K, 2)
    m0 = torch.full((2, K), -1, dtype=torch.int32)
    perm = torch.randperm(600, generator=torch.Generator().manual_seed(0))
    k0[0, :600], k1[0, perm] = torch.from_numpy(p0), torch.from_numpy(p1)
    m0[0, :600] = perm.int()
    m0[0, 5::7] = -1
    res = hg.verify_matches_batched({"keypoints0": k0.to(DEV), "keypoints1": k1.to(DEV), "matches0": m0.to(DEV),
                                     "num_keypoints0": torch.tensor([600, 0], dtype=torch.int32, device=DEV)}, "Homography", 3.0, 0.9999, 2000)
    nm = int(res["num_matches"][0])
    assert nm == int((m0[0] > -1).sum()) and int(res["num_matches"][1]) == 0 and not bool(res["ok"][1])
    rows = torch.nonzero(m0[0] > -1)[:, 0]
    assert torch.equal(res["mkeypoints0"][0, :nm].cpu(), k0[0, rows]) and torch.equal(res["mkeypoints1"][0, :nm].cpu(), k1[0, m0[0, rows].long()])
    Mh = res["model"][0].cpu().numpy()
    c = np.array([[0, 0, 1], [639, 0, 1], [639, 479, 1], [0, 479, 1.0]])
    pe, pg = c @ Mh.T, c @ H.T
    assert np.abs(pe[:, :2] / pe[:, 2:] - pg[:, :2] / pg[:, 2:]).max() < 1.0


def test_throughput_record():
    """Not an assertion on speed -- a record for DESIGN.md: 64 pairs x 2048 matches, the reference's default budget (threshold 8 px,
    confidence 0.9999, 10000 iterations), homography + fundamental matrix per pair as `compute_geometry` estimates them."""
    import time

    from imcui_hip.geometry import ransac_batched

    scenes = [homography_scene(100 + b, 2048, 0.5) for b in range(64)]
    p0, p1, counts = _batch(scenes)
    for _ in range(2):
        ransac_batched(p0, p1, counts, "Homography", 8.0, 0.9999, 10000)
        ransac_batched(p0, p1, counts, "Fundamental", 8.0, 0.9999, 10000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        h = ransac_batched(p0, p1, counts, "Homography", 8.0, 0.9999, 10000)
        f = ransac_batched(p0, p1, counts, "Fundamental", 8.0, 0.9999, 10000)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    assert bool(h["ok"].all()) and bool(f["ok"].all())
    print(f"[geometry] 64 pairs x 2048 matches, H + F, 10000 hypotheses each: {ms:.2f} ms per batch = {64 / ms * 1e3:.0f} pairs/s; "
          f"mean hypotheses consumed H {h['iterations'].float().mean().item():.0f} / F {f['iterations'].float().mean().item():.0f}")


def test_low_inlier_ratio_equals_the_oracle():
    """2500 matches, 90 % outliers, a tight threshold, Fundamental: the first hypotheses have w^8 below 2^-53,
    the regime where `log(1 - w^m)` is 0.  The device's replayed stopping rule must consume the same hypotheses as the oracle's loop."""
    from imcui_hip.geometry import ransac_batched

    scenes = [two_view_scene(21, n=2500, outliers=0.90, noise=0.05), two_view_scene(22, n=2100, outliers=0.93, noise=0.05)]
    p0, p1, counts = _batch(scenes)
    out = ransac_batched(p0, p1, counts, "Fundamental", 0.25, 0.999, 400, seed=3)
    torch.cuda.synchronize()
    for b, s in enumerate(scenes):
        M, mask, info = og.ransac(s[0], s[1], 1, 0.25, 0.999, 400, seed=3, pair_index=b)
        assert info["used"] > 1
        assert int(out["iterations"][b]) == info["used"], (b, int(out["iterations"][b]), info["used"])
        assert bool(out["ok"][b]) == (M is not None)
