"""AUC parity leg: synthetic pairs with a known homography go through the HIP
SuperPoint+LightGlue path and through the CPU oracle; both match sets feed the same seeded host
DLT-RANSAC (tests/geometry_utils.py; geometric verification stays on the host by north_star) and the
homography corner-error AUC@{3,5,10 px} must agree."""
import numpy as np
import pytest
import torch

from geometry_utils import corner_error, error_auc, ransac_homography
from oracle.lightglue import LightGlueOracle
from oracle.superpoint import SuperPointOracle
from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict

pytestmark = pytest.mark.gpu

H, W, NPAIR = 240, 320, 4
SPC = dict(nms_radius=3, max_keypoints=1024, keypoint_threshold=0.005, remove_borders=4)
LGC = dict(depth_confidence=-1.0, width_confidence=-1.0, match_threshold=0.1)


def _errors(k0, k1, m0, hgt):
    errs = []
    for b in range(len(k0)):
        valid = m0[b] >= 0
        p0 = k0[b][valid].numpy().astype(np.float64)
        p1 = k1[b][m0[b][valid]].numpy().astype(np.float64)
        h, _ = ransac_homography(p0, p1, thresh=3.0, iters=300, seed=b)
        errs.append(corner_error(h, hgt[b].numpy(), W, H))
    return errs


def test_homography_auc_matches_the_oracle(precision):
    from imcui_hip.pipeline import SuperPointLightGluePipeline
    from imcui_hip.synth import make_pair_batch

    ssd, lsd = superpoint_state_dict(0), lightglue_state_dict(0)
    img0, img1, hgt = make_pair_batch(2024, NPAIR, H, W, n_blobs=600)
    pipe = SuperPointLightGluePipeline({**SPC, "state_dict": ssd}, {**LGC, "state_dict": lsd}).eval().to("cuda:0")
    out = pipe(img0.cuda(), img1.cuda())
    torch.cuda.synchronize()
    n0, n1 = out["num_keypoints0"].cpu(), out["num_keypoints1"].cpu()
    k0 = [out["keypoints0"][b, : n0[b]].cpu() for b in range(NPAIR)]
    k1 = [out["keypoints1"][b, : n1[b]].cpu() for b in range(NPAIR)]
    m_hip = [out["matches0"][b, : n0[b]].cpu().long() for b in range(NPAIR)]

    # oracle on the HIP key-points / descriptors (key-point parity itself is tests/test_gpu_superpoint.py):
    # the leg isolates "same matches -> same geometry"
    lg = LightGlueOracle(lsd, dict(depth_confidence=-1.0, width_confidence=-1.0, filter_threshold=0.1))
    m_ref = []
    for b in range(NPAIR):
        ref = lg({"image0": img0[b : b + 1], "image1": img1[b : b + 1], "keypoints0": k0[b][None], "keypoints1": k1[b][None],
                  "descriptors0": out["descriptors0"][b, : n0[b]].cpu().t()[None], "descriptors1": out["descriptors1"][b, : n1[b]].cpu().t()[None]})  # fmt: skip
        m_ref.append(ref["matches0"][0])
    e_hip, e_ref = _errors(k0, k1, m_hip, hgt), _errors(k0, k1, m_ref, hgt)
    auc_hip, auc_ref = error_auc(e_hip), error_auc(e_ref)
    nm = [int((m >= 0).sum()) for m in m_hip]
    print(f"matches/pair {nm}  corner errors hip {np.round(e_hip, 3)} ref {np.round(e_ref, 3)}  AUC@3/5/10 hip {auc_hip} ref {auc_ref}")
    assert min(nm) >= 8, nm
    for b in range(NPAIR):
        assert torch.equal(m_hip[b], m_ref[b]), f"pair {b}: match sets differ"
    assert auc_hip == auc_ref and e_hip == e_ref

    # SuperPoint oracle end to end as well: its key-points must give the same AUC within the tie audit
    sp = SuperPointOracle(ssd)
    f0, f1 = sp({"image": img0}, SPC), sp({"image": img1}, SPC)
    same = [len(set(map(tuple, k0[b].tolist())) & set(map(tuple, f0["keypoints"][b].tolist()))) / max(len(f0["keypoints"][b]), 1) for b in range(NPAIR)]
    assert min(same) >= 0.99, same


def test_homography_auc_matches_the_oracle_superglue():
    """The same leg through SuperPoint + SuperGlue (zoo entry `superglue`, 20 Sinkhorn rounds here)."""
    from imcui_hip.pipeline import SuperPointSuperGluePipeline
    from imcui_hip.synth import make_pair_batch
    from oracle.superglue import SuperGlueOracle
    from imcui_hip.synth_weights import superglue_state_dict

    torch.set_num_threads(8)
    ssd, gsd = superpoint_state_dict(0), superglue_state_dict(0)
    sgc = {"sinkhorn_iterations": 20, "match_threshold": 0.2}
    img0, img1, hgt = make_pair_batch(2025, NPAIR, H, W, n_blobs=600)
    pipe = SuperPointSuperGluePipeline({**SPC, "state_dict": ssd}, {**sgc, "state_dict": gsd}).eval().to("cuda:0")
    out = pipe(img0.cuda(), img1.cuda())
    torch.cuda.synchronize()
    n0, n1 = out["num_keypoints0"].cpu(), out["num_keypoints1"].cpu()
    k0 = [out["keypoints0"][b, : n0[b]].cpu() for b in range(NPAIR)]
    k1 = [out["keypoints1"][b, : n1[b]].cpu() for b in range(NPAIR)]
    m_hip = [out["matches0"][b, : n0[b]].cpu().long() for b in range(NPAIR)]
    sg = SuperGlueOracle(gsd, sgc)
    m_ref = []
    for b in range(NPAIR):
        ref = sg({"image0": img0[b : b + 1], "image1": img1[b : b + 1], "keypoints0": k0[b][None], "keypoints1": k1[b][None],
                  "scores0": out["scores0"][b, : n0[b]].cpu()[None], "scores1": out["scores1"][b, : n1[b]].cpu()[None],
                  "descriptors0": out["descriptors0"][b, : n0[b]].cpu().t()[None], "descriptors1": out["descriptors1"][b, : n1[b]].cpu().t()[None]})  # fmt: skip
        m_ref.append(ref["matches0"][0])
        assert (out["matching_scores0"][b, : n0[b]].cpu() - ref["matching_scores0"][0]).abs().max().item() < 1e-4
    e_hip, e_ref = _errors(k0, k1, m_hip, hgt), _errors(k0, k1, m_ref, hgt)
    auc_hip, auc_ref = error_auc(e_hip), error_auc(e_ref)
    nm = [int((m >= 0).sum()) for m in m_hip]
    print(f"[superglue] matches/pair {nm}  corner errors hip {np.round(e_hip, 3)} ref {np.round(e_ref, 3)}  AUC@3/5/10 hip {auc_hip} ref {auc_ref}")
    assert min(nm) >= 8, nm
    for b in range(NPAIR):
        assert torch.equal(m_hip[b], m_ref[b]), f"pair {b}: match sets differ"
    assert auc_hip == auc_ref and e_hip == e_ref
