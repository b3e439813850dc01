"""SuperPoint HIP path vs the CPU oracle on identical seeded inputs (GPU box only).

Bar (BASELINE.json north_star): key-point indices bit-exact, descriptors / scores within 1e-4
fp32.  Conv accumulation order differs between MFMA and the CPU reference, so decisions are
audited in two steps: (1) the dense score map agrees to round-off; (2) the selection logic
(NMS, threshold, border, top-k, order) is bit-exact when the oracle's selection is run on the
HIP score map; (3) every key-point present in only one of the two end-to-end sets is AUDITED: it must be caused by a
deciding margin (threshold, k-th score, or an order flip between two pixels at most r apart) below twice the measured
dense-map difference on the oracle's own score map (parity_utils.audit_keypoint_differences) -- otherwise the test fails.
"""
import pytest
import torch

from imcui_hip.synth import make_pair
from oracle.superpoint import SuperPointOracle
from imcui_hip.synth_weights import superpoint_state_dict
from parity_utils import audit_keypoint_differences, oracle_select_on

pytestmark = pytest.mark.gpu

SD = superpoint_state_dict(0)


def _model(conf):
    from imcui_hip.hloc.extractors.superpoint import SuperPoint

    return SuperPoint({**conf, "state_dict": SD}).eval().to("cuda:0")


@pytest.mark.parametrize(
    "h,w,conf",
    [
        (480, 640, dict(nms_radius=3, max_keypoints=2048, keypoint_threshold=0.005, remove_borders=4)),
        (240, 320, dict(nms_radius=4, max_keypoints=-1, keypoint_threshold=0.005, remove_borders=4)),
        (120, 200, dict(nms_radius=2, max_keypoints=300, keypoint_threshold=0.02, remove_borders=6, fix_sampling=True)),
    ],
)
def test_superpoint_vs_oracle(h, w, conf, precision):
    torch.set_num_threads(8)
    img0, img1, _ = make_pair(11, h, w, n_blobs=max(200, h * w // 150))
    images = torch.cat([img0, img1], 0)
    model = _model(conf)
    full = {**model.default_conf, **conf}
    out = model.forward_batched(images.cuda(), want_score_map=True)
    torch.cuda.synchronize()
    ora = SuperPointOracle(SD)
    for b in range(2):
        ref = ora({"image": images[b : b + 1]}, full, return_intermediates=True)
        dense_hip = out["score_map"][b].cpu()
        dense_ref = ref["_dense_scores"][0]
        assert (dense_hip - dense_ref).abs().max().item() < 2e-5, "dense detector scores"
        n = int(out["num_keypoints"][b])
        kp = out["keypoints"][b, :n].cpu()
        sc = out["scores"][b, :n].cpu()
        flat_hip = (kp[:, 1] * w + kp[:, 0]).long()
        # (2) selection logic, bit-exact on the HIP score map
        (flat_sel, sc_sel, tie), _ = oracle_select_on(dense_hip, full)
        if not tie:
            assert n == len(flat_sel)
            if full["max_keypoints"] >= 0 and full["max_keypoints"] == n:
                assert torch.equal(flat_hip, flat_sel), "top-k order / membership"
            else:
                assert torch.equal(flat_hip, flat_sel), "row-major candidate order"
            assert torch.equal(sc, sc_sel)
        # (3) end-to-end vs the pure oracle: same key-points up to round-off ties
        kp_ref = ref["keypoints"][0]
        flat_ref = (kp_ref[:, 1] * w + kp_ref[:, 0]).long()
        common = set(flat_hip.tolist()) & set(flat_ref.tolist())
        assert len(common) >= 0.99 * max(len(flat_ref), 1), (len(common), len(flat_ref))
        # every index present in one set only must be explained by a round-off tie on the oracle's own dense map
        n_ties = audit_keypoint_differences(flat_hip, flat_ref, dense_hip, dense_ref, full, tag=f"{h}x{w} image {b}")
        # descriptors / scores of the common key-points
        lut = {v: i for i, v in enumerate(flat_ref.tolist())}
        idx_h = [i for i, v in enumerate(flat_hip.tolist()) if v in lut]
        idx_r = [lut[flat_hip[i].item()] for i in idx_h]
        d_hip = out["descriptors"][b, :n].cpu()[idx_h]
        d_ref = ref["descriptors"][0].t()[idx_r]
        assert (d_hip - d_ref).abs().max().item() < 1e-4
        assert (sc[idx_h] - ref["scores"][0][idx_r]).abs().max().item() < 1e-4
        print(f"[audit] {h}x{w} image {b} precision {precision}: {len(flat_ref)} key-points, {n_ties} differ from the pure oracle, "
              f"all audited round-off ties (dense map max diff {(dense_hip - dense_ref).abs().max().item():.2e})")


def test_superpoint_plugin_contract():
    """Reference extractor contract: lists per image, (x,y) float32, descriptors [256, N]."""
    img0, _, _ = make_pair(5, 240, 320, n_blobs=400)
    model = _model(dict(nms_radius=3, max_keypoints=500))
    with torch.no_grad():
        pred = model({"image": img0.cuda()})
    assert set(pred) == {"keypoints", "scores", "descriptors"}
    k, s, d = pred["keypoints"][0], pred["scores"][0], pred["descriptors"][0]
    assert k.ndim == 2 and k.shape[1] == 2 and k.dtype == torch.float32
    assert d.shape == (256, k.shape[0]) and s.shape == (k.shape[0],)
    assert k.shape[0] <= 500
    assert (d.norm(dim=0) - 1).abs().max().item() < 1e-5
    # runtime conf is re-read on every call (imcui/ui/utils.py:961-962 mutates it)
    model.conf["max_keypoints"] = 100
    with torch.no_grad():
        pred2 = model({"image": img0.cuda()})
    assert pred2["keypoints"][0].shape[0] == 100
    assert torch.equal(pred2["keypoints"][0], k[:100])


def test_superpoint_flat_image_is_handled():
    """A constant image ties every score: the reference returns every interior pixel above thr."""
    model = _model(dict(nms_radius=3, max_keypoints=200))
    with torch.no_grad():
        pred = model({"image": torch.full((1, 1, 64, 96), 0.5).cuda()})
    assert pred["keypoints"][0].shape[0] <= 200


def test_superpoint_large_topk_and_rejects_oversize():
    """max_keypoints between 8193 and 16384 (the UI slider goes to 10000) runs on the 16384-key on-chip sorter and
    equals the selection logic on the HIP score map; above that the host rejects the call instead of returning nothing."""
    from imcui_hip import ImcuiHipError

    img0, _, _ = make_pair(21, 480, 640, n_blobs=6000)
    conf = dict(nms_radius=1, max_keypoints=10000, keypoint_threshold=0.0005, remove_borders=4)
    model = _model(conf)
    full = {**model.default_conf, **conf}
    out = model.forward_batched(img0.cuda(), want_score_map=True)
    torch.cuda.synchronize()
    assert int(out["status"][0]) == 0
    n = int(out["num_keypoints"][0])
    (flat_sel, sc_sel, tie), nms = oracle_select_on(out["score_map"][0].cpu(), full)
    assert (nms > conf["keypoint_threshold"]).sum() > 10000, "test image must yield more candidates than max_keypoints"
    assert n == 10000 == len(flat_sel)
    kp = out["keypoints"][0, :n].cpu()
    if not tie:
        assert torch.equal((kp[:, 1] * 640 + kp[:, 0]).long(), flat_sel)
        assert torch.equal(out["scores"][0, :n].cpu(), sc_sel)
    model.conf["max_keypoints"] = 20000
    with pytest.raises(ImcuiHipError, match="top-k sorter"):
        model.forward_batched(img0.cuda())


def test_superpoint_all_keypoints_path_reports_overflow_on_the_device():
    """max_keypoints = -1 never synchronises: the selection status is a device tensor.  An output capacity below the
    number of key-points sets bit 1 and returns the first `kcap` (row-major) ones; the ragged plugin path reads the
    status together with the counts and retries with room for every pixel."""
    conf = dict(nms_radius=3, max_keypoints=-1, remove_borders=4, keypoint_threshold=0.005)
    model = _model(conf)
    img0, _, _ = make_pair(5, 240, 320, n_blobs=400)
    full = model.forward_batched(img0.cuda())
    n = int(full["num_keypoints"][0])
    assert int(full["status"][0]) == 0 and n > 64
    small = model._impl.forward(model.packed, img0.cuda(), model.conf, kcap=64)
    assert int(small["status"][0]) & 2 and int(small["num_keypoints"][0]) == 64
    assert torch.equal(small["keypoints"][0], full["keypoints"][0, :64])
    # the plugin's retry: force the first attempt to overflow by shrinking the bound it sizes the outputs with
    lib = model._impl.forward.__globals__["load_library"]()
    orig = lib.imcui_hip_superpoint_max_keypoints_bound
    try:
        lib.imcui_hip_superpoint_max_keypoints_bound = lambda *a: 32
        with torch.no_grad():
            pred = model({"image": img0.cuda()})
    finally:
        lib.imcui_hip_superpoint_max_keypoints_bound = orig
    assert pred["keypoints"][0].shape[0] == n
    assert torch.equal(pred["keypoints"][0], full["keypoints"][0, :n])
