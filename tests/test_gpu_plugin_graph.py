"""The plugins' opt-in `hip_graph` conf (GPU box only): `SuperPoint._forward` / `LightGlue._forward` replayed from HIP graphs captured per image
shape / key-point capacity must return EXACTLY what the eager launches return -- for several images and pairs in turn (static buffers are
re-used: the outputs of an earlier call must not change when the next one runs), for pairs with different key-point counts inside one
capacity class, and for the reference's default adaptive conf (early stop and pruning live on the device: the graph carries them)."""
import pytest
import torch

from imcui_hip.synth import make_pair_batch
from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dc,wc", [(-1.0, -1.0), (0.95, 0.99)])
def test_plugins_replayed_from_graphs_equal_eager_launches(dc, wc):
    from imcui_hip.hloc.extractors.superpoint import SuperPoint
    from imcui_hip.hloc.matchers.lightglue import LightGlue

    spc = {"nms_radius": 3, "max_keypoints": 1024, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": superpoint_state_dict(0)}
    lgc = {"depth_confidence": dc, "width_confidence": wc, "match_threshold": 0.1, "state_dict": lightglue_state_dict(0)}
    ext_e, ext_g = SuperPoint(dict(spc)).eval().to(DEV), SuperPoint({**spc, "hip_graph": True}).eval().to(DEV)
    m_e, m_g = LightGlue(dict(lgc)).eval().to(DEV), LightGlue({**lgc, "hip_graph": True}).eval().to(DEV)
    img0, img1, _ = make_pair_batch(77, 4, 240, 320, distinct=4)
    img0, img1 = img0.to(DEV), img1.to(DEV)
    # an image with few key-points: another count inside the same capacity class of the matcher's graph
    img1[3] = torch.nn.functional.avg_pool2d(img1[3:4], 9, 1, 4)[0]
    kept = []
    with torch.no_grad():
        for i in range(4):
            feats = []
            for ext in (ext_e, ext_g):
                f0, f1 = ext({"image": img0[i : i + 1]}), ext({"image": img1[i : i + 1]})
                feats.append((f0, f1))
            (e0, e1), (g0, g1) = feats
            for a, b in ((e0, g0), (e1, g1)):
                for k in ("keypoints", "scores", "descriptors"):
                    assert torch.equal(a[k][0], b[k][0]), (i, k)
            kept.append((g0["keypoints"][0], g0["keypoints"][0].clone()))
            data = {"image0": img0[i : i + 1], "image1": img1[i : i + 1], "keypoints0": e0["keypoints"][0][None], "keypoints1": e1["keypoints"][0][None],
                    "scores0": e0["scores"][0][None], "scores1": e1["scores"][0][None], "descriptors0": e0["descriptors"][0][None], "descriptors1": e1["descriptors"][0][None]}  # fmt: skip
            pe, pg = m_e(data), m_g(data)
            for k in ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1"):
                assert torch.equal(pe[k], pg[k]), (i, k)
            assert pe["stop"] == pg["stop"], i
            assert len(pe["matches"][0]) == len(pg["matches"][0])
    for t, snap in kept:  # earlier outputs were cloned out of the static buffers: later replays did not touch them
        assert torch.equal(t, snap)
    assert len(m_g._graphs) >= 1 and all(v is not None for v in m_g._graphs.values()), "the capture fell back to eager launches"
    assert len(ext_g._graphs) == 1 and all(v is not None for v in ext_g._graphs.values())


def test_graphs_of_growing_capacity_each_own_their_workspace():
    """a second graph of the same plugin that needs a LARGER workspace (LightGlue capacity 1024 after 256, a bigger SuperPoint
    image after a small one) must capture -- not fall back to eager launches because the first graph pinned the shared capture stream's
    scratch -- and all graphs must keep replaying correctly afterwards, in any order."""
    import warnings

    from imcui_hip.hloc.extractors.superpoint import SuperPoint
    from imcui_hip.hloc.matchers.lightglue import LightGlue

    lsd, ssd = lightglue_state_dict(0), superpoint_state_dict(0)
    lgc = {"depth_confidence": -1.0, "width_confidence": -1.0, "match_threshold": 0.1, "state_dict": lsd}
    m_e, m_g = LightGlue(dict(lgc)).eval().to(DEV), LightGlue({**lgc, "hip_graph": True}).eval().to(DEV)
    g = torch.Generator().manual_seed(5)

    def pair(n):
        k0, k1 = torch.rand(1, n, 2, generator=g) * torch.tensor([640.0, 480.0]), torch.rand(1, n, 2, generator=g) * torch.tensor([640.0, 480.0])
        d0, d1 = torch.nn.functional.normalize(torch.randn(1, 256, n, generator=g), dim=1), torch.nn.functional.normalize(torch.randn(1, 256, n, generator=g), dim=1)
        img = torch.zeros(1, 1, 480, 640)
        return {k: v.to(DEV) for k, v in {"image0": img, "image1": img, "keypoints0": k0, "keypoints1": k1, "scores0": torch.rand(1, n, generator=g),
                                          "scores1": torch.rand(1, n, generator=g), "descriptors0": d0, "descriptors1": d1}.items()}  # fmt: skip

    pairs = [pair(n) for n in (200, 900, 1900, 250, 1000)]  # capacities 256, 1024, 1920, 256, 1024
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # a capture that falls back warns: that is a failure here
        with torch.no_grad():
            for rnd in range(2):
                for d in pairs:
                    pe, pg = m_e(d), m_g(d)
                    for k in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
                        assert torch.equal(pe[k], pg[k]), (rnd, d["keypoints0"].shape, k)
    assert len(m_g._graphs) == 3 and all(v is not None for v in m_g._graphs.values())
    spc = {"nms_radius": 3, "max_keypoints": 512, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": ssd}
    ext_e, ext_g = SuperPoint(dict(spc)).eval().to(DEV), SuperPoint({**spc, "hip_graph": True}).eval().to(DEV)
    imgs = [make_pair_batch(80 + i, 1, h, w)[0].to(DEV) for i, (h, w) in enumerate(((120, 160), (480, 640), (240, 320), (120, 160)))]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with torch.no_grad():
            for rnd in range(2):
                for im in imgs:
                    a, b = ext_e({"image": im}), ext_g({"image": im})
                    for k in ("keypoints", "scores", "descriptors"):
                        assert torch.equal(a[k][0], b[k][0]), (rnd, tuple(im.shape), k)
    assert len(ext_g._graphs) == 3 and all(v is not None for v in ext_g._graphs.values())
