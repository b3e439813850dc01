"""Real images through the HIP path (AUC leg): the reference's 15 EVD pairs with ground-truth
homographies plus its own tests/data pair, decoded by tests/golden/make_evd_fixtures.py into tests/golden/evd_pairs.npz
(gray, 640 x 480 -- the `superpoint_max` force-resize).  HIP SuperPoint + LightGlue vs the oracle on identical bytes:
dense score maps to round-off, every key-point difference an audited tie, matches identical, and the same seeded host
DLT-RANSAC on both match sets gives identical corner errors and AUC@{3,5,10 px}.

The weights are seeded random tensors (no checkpoint offline), so on the EVD pairs themselves (extreme view changes)
few matches survive and both AUCs are ~0 -- the pairs exercise real image statistics (flat / saturated regions,
exact score ties, 8-bit quantisation) rather than matching quality.  The second test therefore warps real EVD images
with mild known homographies: real texture AND a recoverable ground truth.
"""
import io
import os

import numpy as np
import pytest
import torch

from geometry_utils import corner_error, error_auc, ransac_homography
from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict
from oracle.lightglue import LightGlueOracle
from oracle.superpoint import SuperPointOracle
from parity_utils import assert_matches_equal_or_tied, audit_keypoint_differences

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
H, W = 480, 640
SPC = dict(nms_radius=3, max_keypoints=2048, keypoint_threshold=0.005, remove_borders=4)  # configs/extractors.py:29-45


def load_pairs():
    from PIL import Image

    z = np.load(os.path.join(HERE, "golden", "evd_pairs.npz"))
    names = [str(n) for n in z["names"]]
    # Round 5: the PNG bytes of the 32 images go through the DEVICE decode path (zlib on the library's host threads, scan-line filters on the
    # device: imcui_hip/hloc/utils/png.py), the way the batch drivers read the reference's EVD / WxBS files; PIL decodes them too and
    # the two must agree bit for bit before anything else is compared.
    from imcui_hip.hloc.utils.png import PngDecoder

    blobs = [z[f"img{s}_{i}"].tobytes() for s in (0, 1) for i in range(len(names))]
    dec = PngDecoder("cuda:0", threads=8)
    dev_imgs = dec.decode_batch(blobs)
    dec.close()
    for b, t in zip(blobs, dev_imgs):
        assert isinstance(t, torch.Tensor) and np.array_equal(t.cpu().numpy(), np.asarray(Image.open(io.BytesIO(b)))), "device PNG decode != PIL"
    imgs = torch.stack([t.cpu() for t in dev_imgs]).float() / 255.0
    n = len(names)
    return names, imgs[:n, None], imgs[n:, None], z["homographies"]


def _run_and_check(img0, img1, hgt, lgc, names, need_matches):
    """HIP pipeline on the batch vs the oracle pair by pair; returns (errors_hip, errors_ref, matches per pair)."""
    from imcui_hip.pipeline import SuperPointLightGluePipeline

    torch.set_num_threads(16)
    ssd, lsd = superpoint_state_dict(0), lightglue_state_dict(0)
    pipe = SuperPointLightGluePipeline({**SPC, "state_dict": ssd}, {**lgc, "state_dict": lsd}).eval().to("cuda:0")
    B = img0.shape[0]
    dense = pipe.extractor.forward_batched(torch.cat([img0, img1]).cuda(), want_score_map=True)["score_map"].cpu()
    out = {k: v.cpu() for k, v in pipe(img0.cuda(), img1.cuda()).items()}
    sp = SuperPointOracle(ssd)
    lg = LightGlueOracle(lsd, dict(depth_confidence=lgc["depth_confidence"], width_confidence=lgc["width_confidence"], filter_threshold=lgc["match_threshold"]))
    e_hip, e_ref, nm, n_ties = [], [], [], 0
    for b in range(B):
        n0, n1 = int(out["num_keypoints0"][b]), int(out["num_keypoints1"][b])
        for side, (img, kp, n) in enumerate(((img0, out["keypoints0"], n0), (img1, out["keypoints1"], n1))):
            ref = sp({"image": img[b : b + 1]}, SPC, return_intermediates=True)
            d_hip, d_ref = dense[side * B + b], ref["_dense_scores"][0]
            assert (d_hip - d_ref).abs().max().item() < 2e-5, f"{names[b]} image {side}: dense scores"
            k_h, k_r = kp[b, :n], ref["keypoints"][0]
            flat_h, flat_r = (k_h[:, 1] * W + k_h[:, 0]).long(), (k_r[:, 1] * W + k_r[:, 0]).long()
            n_ties += audit_keypoint_differences(flat_h, flat_r, d_hip, d_ref, SPC, tag=f"{names[b]} image {side}")
            assert len(set(flat_h.tolist()) & set(flat_r.tolist())) >= 0.98 * len(flat_r), names[b]
        k0, k1 = out["keypoints0"][b, :n0], out["keypoints1"][b, :n1]
        ref = lg({"image0": img0[b : b + 1], "image1": img1[b : b + 1], "keypoints0": k0[None], "keypoints1": k1[None],
                  "descriptors0": out["descriptors0"][b, :n0].t()[None], "descriptors1": out["descriptors1"][b, :n1].t()[None]},
                 return_intermediates=True)  # fmt: skip
        assert int(out["stop"][b]) == ref["stop"], names[b]
        m_h, m_r = out["matches0"][b, :n0].long(), ref["matches0"][0]
        if "_log_assignment" in ref:
            assert_matches_equal_or_tied(m_h, ref["_log_assignment"][0], m_r, lgc["match_threshold"], tag=names[b], ind0=ref.get("_ind0"), ind1=ref.get("_ind1"))
        else:  # a side lost all its points: nothing was matched
            assert torch.equal(m_h, m_r), names[b]
        same = m_h == m_r
        assert (out["matching_scores0"][b, :n0] - ref["matching_scores0"][0]).abs()[same].max().item() < 1e-4, names[b]
        nm.append(int((m_r >= 0).sum()))
        if np.isnan(hgt[b]).any():
            continue
        for m, errs in ((m_h, e_hip), (m_r, e_ref)):
            v = m >= 0
            if int(v.sum()) < 4:
                errs.append(float("inf"))
                continue
            hm, _ = ransac_homography(k0[v].numpy().astype(np.float64), k1[m[v]].numpy().astype(np.float64), thresh=3.0, iters=300, seed=b)
            errs.append(float(corner_error(hm, np.asarray(hgt[b]), W, H)))
    auc_hip, auc_ref = error_auc(e_hip), error_auc(e_ref)
    print(f"[real images] {B} pairs, matches/pair {nm}, {n_ties} audited key-point ties, AUC@3/5/10 hip {auc_hip} oracle {auc_ref}")
    assert e_hip == e_ref and auc_hip == auc_ref
    assert sum(nm) >= need_matches, nm
    return auc_ref


@pytest.mark.parametrize("lgc", [dict(depth_confidence=0.95, width_confidence=0.99, match_threshold=0.2),
                                 dict(depth_confidence=-1.0, width_confidence=-1.0, match_threshold=0.1)], ids=["zoo-conf", "fixed-depth"])  # fmt: skip
def test_evd_pairs_parity_and_auc(lgc):
    """configs/matchers.py:34-50 (`superpoint-lightglue`: match_threshold 0.2, depth 0.95, width 0.99) and the bench's
    fixed-work mode on the 16 real pairs."""
    names, img0, img1, hgt = load_pairs()
    _run_and_check(img0, img1, hgt, lgc, names, need_matches=20)


def test_real_texture_known_homography_auc(precision):
    """Real EVD images warped by seeded mild homographies: the AUC is non-trivial and must equal the oracle's."""
    from imcui_hip.synth import random_homography, warp_image

    names, img0, _, _ = load_pairs()
    pick = [names.index(n) for n in ("graf", "cafe", "there", "girl", "shop", "grand")]
    g = torch.Generator().manual_seed(7)
    a = img0[pick]
    hm = torch.stack([random_homography(g, H, W) for _ in pick])
    b = torch.cat([warp_image(a[i : i + 1], hm[i]) for i in range(len(pick))]).clamp_(0, 1)
    b = (b * 255).round() / 255  # 8-bit images, like a decoded file
    auc = _run_and_check(a, b, hm.numpy().astype(np.float64), dict(depth_confidence=-1.0, width_confidence=-1.0, match_threshold=0.1),
                         [names[i] for i in pick], need_matches=200)  # fmt: skip
    assert auc[2] > 0.3, f"AUC@10 {auc}: the warped real-image pairs should be matchable"


def test_hip_ransac_against_a_textbook_ransac_on_real_textures():
    """`HIP_RANSAC` (csrc/geometry.hip) is its own, fully specified RANSAC, not cv2's USAC_MAGSAC, and its oracle
    restates its own specification.  This records how it relates to "a" reference RANSAC: the 16 real EVD images warped by seeded mild homographies,
    matched by the HIP pipeline, then verified (a) on the device by `ransac_batched` and (b) by the seeded textbook numpy DLT-RANSAC of
    tests/geometry_utils.py on the SAME matches: corner errors against the ground truth, AUC@{3, 5, 10 px} of both printed; they must agree to a
    few hundredths (both estimate one homography from the same inliers; sampling differs)."""
    from imcui_hip.geometry import ransac_batched
    from imcui_hip.pipeline import SuperPointLightGluePipeline
    from imcui_hip.synth import random_homography, warp_image

    names, img0, _, _ = load_pairs()
    g = torch.Generator().manual_seed(11)
    a = img0
    hm = torch.stack([random_homography(g, H, W) for _ in names])
    b = torch.cat([warp_image(a[i : i + 1], hm[i]) for i in range(len(names))]).clamp_(0, 1)
    b = (b * 255).round() / 255
    pipe = SuperPointLightGluePipeline({**SPC, "state_dict": superpoint_state_dict(0)},
                                       {"depth_confidence": -1.0, "width_confidence": -1.0, "match_threshold": 0.1, "state_dict": lightglue_state_dict(0)}).eval().to("cuda:0")  # fmt: skip
    out = pipe(a.cuda(), b.cuda())
    B = len(names)
    nmax = int(max(int((out["matches0"][i] > -1).sum()) for i in range(B)))
    p0, p1 = torch.zeros(B, max(nmax, 4), 2, device="cuda:0"), torch.zeros(B, max(nmax, 4), 2, device="cuda:0")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    for i in range(B):
        n0 = int(out["num_keypoints0"][i])
        m = out["matches0"][i, :n0].long()
        v = m > -1
        k0, k1 = out["keypoints0"][i, :n0][v], out["keypoints1"][i][m[v]]
        p0[i, : len(k0)], p1[i, : len(k1)], cnt[i] = k0, k1, len(k0)
    res = ransac_batched(p0, p1, cnt, "Homography", 3.0, 0.9999, 2000, seed=1)
    torch.cuda.synchronize()
    e_dev, e_np, used = [], [], []
    for i in range(B):
        n = int(cnt[i])
        if n < 8:
            continue
        hgt = hm[i].numpy().astype(np.float64)
        e_dev.append(float(corner_error(res["model"][i].cpu().numpy(), hgt, W, H)) if bool(res["ok"][i]) else float("inf"))
        hn, _ = ransac_homography(p0[i, :n].cpu().numpy().astype(np.float64), p1[i, :n].cpu().numpy().astype(np.float64), thresh=3.0, iters=2000, seed=i)
        e_np.append(float(corner_error(hn, hgt, W, H)) if hn is not None else float("inf"))
        used.append(n)
    auc_dev, auc_np = error_auc(e_dev), error_auc(e_np)
    print(f"[ransac] {len(used)} warped real-texture pairs, matches/pair {used}: AUC@3/5/10 HIP_RANSAC {auc_dev}, numpy DLT-RANSAC {auc_np}; "
          f"median corner error {np.median(e_dev):.3f} vs {np.median(e_np):.3f} px")
    assert len(used) >= 12
    # recorded (round 5, three refit rounds: AUC@10 0.33 vs 0.42, median corner error 6.9 vs 4.6 px -- the device RANSAC stops at its confidence bound after
    # ~150 hypotheses where the textbook loop scores all 2000; one refit round: 0.26 / 7.7 px) and bounded: not worse than the textbook by more than 0.15
    assert all(x >= y - 0.15 for x, y in zip(auc_dev, auc_np)), (auc_dev, auc_np)
