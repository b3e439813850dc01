"""LightGlue HIP path vs the CPU oracle on identical seeded inputs (GPU box only).

Bar: matches0 / matches1 / stop / prune bit-exact, matching scores within 1e-4.  Ragged batches:
several pairs of different sizes go through ONE HIP call and are compared with per-pair oracle runs.
"""
import pytest
import torch

from oracle.lightglue import LightGlueOracle
from imcui_hip.synth_weights import lightglue_state_dict
from parity_utils import assert_matches_equal_or_tied, synthetic_matching_problem

pytestmark = pytest.mark.gpu

LSD = lightglue_state_dict(0)
SIZES = [(700, 650, 150), (512, 512, 100), (130, 257, 30), (1024, 900, 300)]
# Weight sets of the full-size tests.  "damped" = the structured set above (residual updates x 0.03: control flow is
# exercised, but the layers are near-identity).  "strong" = same shaped heads with full-strength layers: per-layer
# relative updates 0.4-1.4, LayerNorm gamma / beta spread 0.1, final_proj gain lowered so similarities stay ~100
# (above that fp32 round-off alone exceeds 1e-4 on the scores).  "random" = plain random weights, nothing shaped.
WEIGHTS = {
    "damped": LSD,
    "strong": lightglue_state_dict(0, damp=0.1, ln_noise=0.1, final_gain=10.0),
    "random": lightglue_state_dict(1, structured=False),
}
IMG = torch.zeros(1, 1, 480, 640)


def _model(dc, wc, th=0.1, sd=LSD, pruning_device="cpu"):
    from imcui_hip.hloc.matchers.lightglue import LightGlue

    return LightGlue({"depth_confidence": dc, "width_confidence": wc, "match_threshold": th, "state_dict": sd,
                      "pruning_device": pruning_device}).eval().to("cuda:0")  # fmt: skip


def _oracle_pair(ora, a, c, e, f):
    return ora({"image0": IMG, "image1": IMG, "keypoints0": a[None], "keypoints1": c[None],
                "descriptors0": e.t()[None], "descriptors1": f.t()[None]}, return_intermediates=True)  # fmt: skip


def _check_layers(dump, b, ref, tag):
    """Token states after every executed layer vs the oracle's (`_layers`), 1e-4 of the layer's magnitude."""
    worst = 0.0
    for li, (r0, r1) in enumerate(ref["_layers"]):
        for s, r in enumerate((r0[0], r1[0])):
            got = dump[li, 2 * b + s, : r.shape[0]].cpu()
            err = (got - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
            worst = max(worst, err)
            assert err < 1e-4, f"{tag}: layer {li} image {s}: relative error {err:.3e}"
    return worst


def _batch(problems):
    B = len(problems)
    ncap = max(max(p[0].shape[0], p[1].shape[0]) for p in problems)
    k0 = torch.zeros(B, ncap, 2)
    k1 = torch.zeros(B, ncap, 2)
    d0 = torch.zeros(B, ncap, 256)
    d1 = torch.zeros(B, ncap, 256)
    n0 = torch.zeros(B, dtype=torch.int32)
    n1 = torch.zeros(B, dtype=torch.int32)
    for b, (a, c, e, f) in enumerate(problems):
        k0[b, : len(a)], k1[b, : len(c)], d0[b, : len(a)], d1[b, : len(c)] = a, c, e, f
        n0[b], n1[b] = len(a), len(c)
    return k0, k1, d0, d1, n0, n1


@pytest.mark.parametrize("dc,wc", [(-1, -1), (0.95, 0.99), (0.95, -1), (-1, 0.99)])
def test_lightglue_ragged_batch_vs_oracle(dc, wc, precision):
    torch.set_num_threads(8)
    problems = [synthetic_matching_problem(20 + i, n, m, o) for i, (n, m, o) in enumerate(SIZES)]
    k0, k1, d0, d1, n0, n1 = _batch(problems)
    model = _model(dc, wc)
    out = model.forward_batched(k0.cuda(), k1.cuda(), d0.cuda(), d1.cuda(), n0.cuda(), n1.cuda(), (640, 480), (640, 480))
    torch.cuda.synchronize()
    out = {k: v.cpu() for k, v in out.items()}
    ora = LightGlueOracle(LSD, dict(depth_confidence=dc, width_confidence=wc, filter_threshold=0.1))
    img = torch.zeros(1, 1, 480, 640)
    for b, (a, c, e, f) in enumerate(problems):
        ref = ora({"image0": img, "image1": img, "keypoints0": a[None], "keypoints1": c[None],
                   "descriptors0": e.t()[None], "descriptors1": f.t()[None]})  # fmt: skip
        na, nc = len(a), len(c)
        tag = f"pair {b} (n={na},{nc}) dc={dc} wc={wc}"
        assert int(out["stop"][b]) == ref["stop"], tag
        assert (ref["matches0"] > -1).sum() > 10, tag
        assert torch.equal(out["prune0"][b, :na].long(), ref["prune0"][0].long()), tag
        assert torch.equal(out["prune1"][b, :nc].long(), ref["prune1"][0].long()), tag
        assert torch.equal(out["matches0"][b, :na].long(), ref["matches0"][0]), tag
        assert torch.equal(out["matches1"][b, :nc].long(), ref["matches1"][0]), tag
        assert (out["matching_scores0"][b, :na] - ref["matching_scores0"][0]).abs().max().item() < 1e-4, tag
        assert (out["matching_scores1"][b, :nc] - ref["matching_scores1"][0]).abs().max().item() < 1e-4, tag
        assert (out["matches0"][b, na:] == -1).all() and (out["matching_scores0"][b, na:] == 0).all()


@pytest.mark.parametrize("weights", ["damped", "strong", "random"])
@pytest.mark.parametrize("dc,wc", [(-1, -1), (0.95, 0.99), (0.95, -1)])
def test_lightglue_full_size_vs_oracle(dc, wc, weights, precision):
    """The sizes the bench runs (BASELINE configs[2]: N = M = 2048 -> R = 2048, 16 query blocks per (sequence, head),
    128 x 256 GEMM tiles) against the oracle: per-layer token states within 1e-4, matches exact (a differing row must
    be an audited tie of the oracle's own log-assignment), scores within 1e-4, stop / prune exact."""
    torch.set_num_threads(16)
    sd = WEIGHTS[weights]
    problems = [synthetic_matching_problem(40, 2048, 2048, 300), synthetic_matching_problem(41, 2048, 1900, 250)]
    k0, k1, d0, d1, n0, n1 = _batch(problems)
    model = _model(dc, wc, sd=sd)
    out = model.forward_batched(k0.cuda(), k1.cuda(), d0.cuda(), d1.cuda(), n0.cuda(), n1.cuda(), (640, 480), (640, 480), layer_dump=True)
    torch.cuda.synchronize()
    dump = out.pop("_layers")
    out = {k: v.cpu() for k, v in out.items()}
    ora = LightGlueOracle(sd, dict(depth_confidence=dc, width_confidence=wc, filter_threshold=0.1))
    for b, (a, c, e, f) in enumerate(problems):
        ref = _oracle_pair(ora, a, c, e, f)
        na, nc = len(a), len(c)
        tag = f"{weights} pair {b} dc={dc} wc={wc} precision={precision}"
        assert int(out["stop"][b]) == ref["stop"], tag
        worst = _check_layers(dump, b, ref, tag)
        assert torch.equal(out["prune0"][b, :na].long(), ref["prune0"][0].long()), tag
        assert torch.equal(out["prune1"][b, :nc].long(), ref["prune1"][0].long()), tag
        # two fp32 evaluations of a similarity of magnitude |sim| differ by ~1e-6 |sim| in the log-assignment: the
        # shaped weight sets keep |sim| ~ 100 (-> 1e-4); plain random weights reach |sim| ~ 2000 and get that much more
        tol = 1e-4 * max(1.0, ref["_sim"].abs().max().item() / 100.0)
        ties = assert_matches_equal_or_tied(out["matches0"][b, :na], ref["_log_assignment"][0], ref["matches0"][0], 0.1, tol=tol, tag=tag,
                                            ind0=ref.get("_ind0"), ind1=ref.get("_ind1"))  # fmt: skip
        if ties == 0:
            assert torch.equal(out["matches1"][b, :nc].long(), ref["matches1"][0]), tag
        same = out["matches0"][b, :na].long() == ref["matches0"][0]
        d0s = (out["matching_scores0"][b, :na] - ref["matching_scores0"][0]).abs()
        assert d0s[same].max().item() < tol, (tag, d0s[same].max().item(), tol)
        # measured-class bounds next to the magnitude-scaled tolerance: round 3 measured layer errors
        # <= 1.8e-6 and score errors <= 1.2e-5 (damped), 2.7e-5 (strong), 1.6e-4 (plain random, |sim| ~ 2000) in both modes
        assert worst < 1e-5, (tag, worst)
        assert d0s[same].max().item() < {"damped": 5e-5, "strong": 8e-5, "random": 5e-4}[weights], (tag, d0s[same].max().item())
        print(f"[parity] {tag}: layers {len(ref['_layers'])}, worst layer error {worst:.2e}, matches {(ref['matches0'] > -1).sum().item()}, ties {ties}, score error {d0s[same].max().item():.2e}")
    if weights != "random":
        assert (out["matches0"] > -1).sum() > 20


def test_lightglue_gpu_pruning_thresholds_vs_oracle():
    """pruning_device = "cuda" / "flash" (upstream pruning_keypoint_thresholds 1024 / 1536): a side is pruned only
    while it holds more points than that -- what the reference does when it runs on a GPU."""
    torch.set_num_threads(16)
    problems = [synthetic_matching_problem(50, 1800, 1300, 200), synthetic_matching_problem(51, 900, 1700, 100)]
    k0, k1, d0, d1, n0, n1 = _batch(problems)
    for dev_name, pth in (("cuda", 1024), ("flash", 1536)):
        model = _model(0.95, 0.99, pruning_device=dev_name)
        out = model.forward_batched(k0.cuda(), k1.cuda(), d0.cuda(), d1.cuda(), n0.cuda(), n1.cuda(), (640, 480), (640, 480))
        torch.cuda.synchronize()
        out = {k: v.cpu() for k, v in out.items()}
        ora = LightGlueOracle(LSD, dict(depth_confidence=0.95, width_confidence=0.99, filter_threshold=0.1, pruning_threshold=pth))
        pruned_any = False
        for b, (a, c, e, f) in enumerate(problems):
            ref = _oracle_pair(ora, a, c, e, f)
            na, nc = len(a), len(c)
            tag = f"{dev_name} pair {b}"
            assert int(out["stop"][b]) == ref["stop"], tag
            assert torch.equal(out["prune0"][b, :na].long(), ref["prune0"][0].long()), tag
            assert torch.equal(out["prune1"][b, :nc].long(), ref["prune1"][0].long()), tag
            assert torch.equal(out["matches0"][b, :na].long(), ref["matches0"][0]), tag
            assert (out["matching_scores0"][b, :na] - ref["matching_scores0"][0]).abs().max().item() < 1e-4, tag
            pruned_any |= bool((ref["prune0"] != ref["prune0"].max()).any() or (ref["prune1"] != ref["prune1"].max()).any())
        assert pruned_any or dev_name == "flash"


@pytest.mark.parametrize("add_scale_ori", [False, True], ids=["disk-aliked-128d", "sift-128d-scale-ori"])
def test_lightglue_variants_vs_oracle(add_scale_ori, precision):
    """The zoo's other LightGlue entries (configs/matchers.py:51-83,140-150): 128-d descriptors through `input_proj`
    (disk- / aliked- / raco-lightglue) and, for sift-lightglue, key-point scale + orientation in the positional encoding."""
    import torch.nn.functional as F

    torch.set_num_threads(8)
    sd = lightglue_state_dict(0, input_dim=128, add_scale_ori=add_scale_ori)
    g = torch.Generator().manual_seed(11)
    problems, extras = [], []
    for n, m, n_out in [(700, 650, 150), (130, 257, 30), (1024, 900, 300)]:
        a, c, e, f = synthetic_matching_problem(60 + n, n, m, n_out)
        proj = torch.randn(256, 128, generator=g) / 16.0  # 128-d descriptors with the same correspondences
        problems.append((a, c, F.normalize(e @ proj, dim=1), F.normalize(f @ proj, dim=1)))
        extras.append((torch.rand(n, generator=g) * 4 + 1, torch.rand(n, generator=g) * 6.28, torch.rand(m, generator=g) * 4 + 1, torch.rand(m, generator=g) * 6.28))
    B = len(problems)
    ncap = max(max(p[0].shape[0], p[1].shape[0]) for p in problems)
    k0, k1 = torch.zeros(B, ncap, 2), torch.zeros(B, ncap, 2)
    d0, d1 = torch.zeros(B, ncap, 128), torch.zeros(B, ncap, 128)
    so = [torch.zeros(B, ncap) for _ in range(4)]
    n0, n1 = torch.zeros(B, dtype=torch.int32), torch.zeros(B, dtype=torch.int32)
    for b, ((a, c, e, f), ex) in enumerate(zip(problems, extras)):
        k0[b, : len(a)], k1[b, : len(c)], d0[b, : len(a)], d1[b, : len(c)] = a, c, e, f
        n0[b], n1[b] = len(a), len(c)
        for t, v in zip(so, ex):
            t[b, : len(v)] = v
    model = _model(0.95, 0.99, sd=sd)
    assert model.input_dim == 128 and model.add_scale_ori == add_scale_ori
    out = model.forward_batched(k0.cuda(), k1.cuda(), d0.cuda(), d1.cuda(), n0.cuda(), n1.cuda(), (640, 480), (640, 480),
                                scales_oris=tuple(t.cuda() for t in so) if add_scale_ori else None)  # fmt: skip
    torch.cuda.synchronize()
    out = {k: v.cpu() for k, v in out.items()}
    ora = LightGlueOracle(sd, dict(depth_confidence=0.95, width_confidence=0.99, filter_threshold=0.1))
    total = 0
    for b, ((a, c, e, f), ex) in enumerate(zip(problems, extras)):
        data = {"image0": IMG, "image1": IMG, "keypoints0": a[None], "keypoints1": c[None], "descriptors0": e.t()[None], "descriptors1": f.t()[None]}
        if add_scale_ori:
            data.update(scales0=ex[0][None], oris0=ex[1][None], scales1=ex[2][None], oris1=ex[3][None])
        ref = ora(data, return_intermediates=True)
        na, nc = len(a), len(c)
        tag = f"variant pair {b}"
        assert int(out["stop"][b]) == ref["stop"], tag
        assert torch.equal(out["prune0"][b, :na].long(), ref["prune0"][0].long()), tag
        assert_matches_equal_or_tied(out["matches0"][b, :na], ref["_log_assignment"][0], ref["matches0"][0], 0.1, tag=tag, ind0=ref.get("_ind0"), ind1=ref.get("_ind1"))
        same = out["matches0"][b, :na].long() == ref["matches0"][0]
        assert (out["matching_scores0"][b, :na] - ref["matching_scores0"][0]).abs()[same].max().item() < 1e-4, tag
        total += int((ref["matches0"] > -1).sum())
    assert total > 100
    # wrong descriptor size / missing scales fail loudly
    from imcui_hip import ImcuiHipError

    with pytest.raises(ImcuiHipError):
        model.forward_batched(k0.cuda(), k1.cuda(), torch.zeros(B, ncap, 256).cuda(), torch.zeros(B, ncap, 256).cuda(), n0.cuda(), n1.cuda(), (640, 480), (640, 480))


def test_lightglue_plugin_contract_and_empty():
    """Flat hloc dict in (descriptors [B,256,N]) -> reference keys out; empty side -> all -1."""
    a, c, e, f = synthetic_matching_problem(3, 300, 280, 60)
    model = _model(0.95, 0.99, th=0.2)
    img = torch.zeros(1, 1, 480, 640)
    data = {"image0": img, "image1": img, "keypoints0": a[None].cuda(), "keypoints1": c[None].cuda(),
            "scores0": torch.ones(1, 300).cuda(), "scores1": torch.ones(1, 280).cuda(),
            "descriptors0": e.t()[None].contiguous().cuda(), "descriptors1": f.t()[None].contiguous().cuda()}  # fmt: skip
    with torch.no_grad():
        pred = model(data)
    for key in ("matches0", "matches1", "matching_scores0", "matching_scores1", "stop", "matches", "scores", "prune0", "prune1"):
        assert key in pred
    assert pred["matches0"].dtype == torch.int64 and pred["matches0"].shape == (1, 300)
    assert isinstance(pred["stop"], int)
    ref = LightGlueOracle(LSD, dict(depth_confidence=0.95, width_confidence=0.99, filter_threshold=0.2))(
        {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in data.items()}
    )
    assert torch.equal(pred["matches0"].cpu(), ref["matches0"])
    assert pred["stop"] == ref["stop"]
    assert torch.equal(pred["matches"][0].cpu(), ref["matches"][0])
    # empty second image
    data["keypoints1"] = torch.zeros(1, 0, 2).cuda()
    data["descriptors1"] = torch.zeros(1, 256, 0).cuda()
    data["scores1"] = torch.zeros(1, 0).cuda()
    with torch.no_grad():
        pred = model(data)
    assert (pred["matches0"] == -1).all() and pred["matches1"].shape == (1, 0) and pred["stop"] == 1


def test_superpoint_lightglue_end_to_end(precision):
    """Images in -> match table out through the batched pipeline, vs the oracle chain."""
    from imcui_hip.pipeline import SuperPointLightGluePipeline
    from imcui_hip.synth import make_pair_batch
    from oracle.superpoint import SuperPointOracle
    from imcui_hip.synth_weights import superpoint_state_dict

    torch.set_num_threads(8)
    ssd = superpoint_state_dict(0)
    spc = dict(nms_radius=3, max_keypoints=512, keypoint_threshold=0.005, remove_borders=4)
    pipe = SuperPointLightGluePipeline({**spc, "state_dict": ssd}, {"depth_confidence": 0.95, "width_confidence": 0.99, "match_threshold": 0.1, "state_dict": LSD}).eval().to("cuda:0")
    img0, img1, _ = make_pair_batch(2, 2, 240, 320, n_blobs=600)
    out = pipe(img0.cuda(), img1.cuda())
    torch.cuda.synchronize()
    sp = SuperPointOracle(ssd)
    lg = LightGlueOracle(LSD, dict(depth_confidence=0.95, width_confidence=0.99, filter_threshold=0.1))
    n_equal_sets = 0
    for b in range(2):
        n0, n1 = int(out["num_keypoints0"][b]), int(out["num_keypoints1"][b])
        # (1) the matcher stage in isolation: oracle LightGlue on the HIP key-points / descriptors -> exact matches
        # (or an audited tie of the oracle's own log-assignment), scores within 1e-4
        hk0, hk1 = out["keypoints0"][b, :n0].cpu(), out["keypoints1"][b, :n1].cpu()
        hd0, hd1 = out["descriptors0"][b, :n0].cpu().t(), out["descriptors1"][b, :n1].cpu().t()
        ref = lg({"image0": img0[b : b + 1], "image1": img1[b : b + 1], "keypoints0": hk0[None], "keypoints1": hk1[None],
                  "descriptors0": hd0[None], "descriptors1": hd1[None]}, return_intermediates=True)  # fmt: skip
        assert int(out["stop"][b]) == ref["stop"]
        m_h, m_r = out["matches0"][b, :n0].cpu().long(), ref["matches0"][0]
        assert_matches_equal_or_tied(m_h, ref["_log_assignment"][0], m_r, 0.1, tag=f"end-to-end pair {b}", ind0=ref.get("_ind0"), ind1=ref.get("_ind1"))
        assert (out["matching_scores0"][b, :n0].cpu() - ref["matching_scores0"][0]).abs()[m_h == m_r].max().item() < 1e-4
        # (2) the whole chain against the pure oracle chain: the extractor's round-off (descriptors 1e-5) is
        # amplified by the matcher, so matches are required to agree on 99 % of the key-points common to both
        f0, f1 = sp({"image": img0[b : b + 1]}, spc), sp({"image": img1[b : b + 1]}, spc)
        if torch.equal(hk0, f0["keypoints"][0]) and torch.equal(hk1, f1["keypoints"][0]):
            n_equal_sets += 1
            pure = lg({"image0": img0[b : b + 1], "image1": img1[b : b + 1], "keypoints0": f0["keypoints"][0][None], "keypoints1": f1["keypoints"][0][None],
                       "descriptors0": f0["descriptors"][0][None], "descriptors1": f1["descriptors"][0][None]})  # fmt: skip
            assert (m_h != pure["matches0"][0]).sum().item() <= max(1, n0 // 100)
    print(f"[audit] end-to-end: {n_equal_sets}/2 pairs had bit-identical key-point sets")


def test_pipeline_full_batch_replicas_are_identical():
    """BASELINE configs[2] at the bench batch (16 pairs, 640x480, 2048 key-points): size-independent
    property -- the batch holds 4 distinct pairs replicated 4x, and every replica must produce the SAME
    key-points, matches and scores (bit for bit): the result of a pair may not depend on its slot in the
    batch, on which CU / XCD ran it, or on timing (this caught a load-ordering race in the GEMM)."""
    from imcui_hip.pipeline import SuperPointLightGluePipeline
    from imcui_hip.synth import make_pair_batch
    from imcui_hip.synth_weights import superpoint_state_dict

    B = 16
    pipe = SuperPointLightGluePipeline(
        {"nms_radius": 3, "max_keypoints": 2048, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": superpoint_state_dict(0)},
        {"depth_confidence": -1.0, "width_confidence": -1.0, "match_threshold": 0.1, "state_dict": LSD},
    ).eval().to("cuda:0")
    img0, img1, _ = make_pair_batch(1234, B, 480, 640, distinct=4)
    out = pipe(img0.to("cuda:0"), img1.to("cuda:0"))
    torch.cuda.synchronize()
    assert int(out["num_keypoints0"].min()) > 1000  # the synthetic pairs are feature rich
    assert int((out["matches0"] >= 0).sum()) > 0
    for key in ("keypoints0", "keypoints1", "num_keypoints0", "matches0", "matches1", "matching_scores0", "stop"):
        v = out[key]
        assert not torch.isnan(v.float()).any(), key
        for i in range(4, B):
            assert torch.equal(v[i], v[i % 4]), f"{key}: replica {i} differs from pair {i % 4}"
    # sortedness / consistency of the match table at full size
    m0, m1 = out["matches0"], out["matches1"]
    for b in range(B):
        idx = torch.nonzero(m0[b] >= 0).flatten()
        assert torch.equal(m1[b][m0[b][idx]], idx), "matches0 / matches1 are not mutual"


def test_graph_replay_equals_eager_launches():
    """The batched step has no host synchronisation (counts, early stop and pruning live on the device), so it can be
    captured in a HIP graph; a replay on new inputs must reproduce the eager result bit for bit."""
    from imcui_hip.pipeline import GraphedPipeline, SuperPointLightGluePipeline
    from imcui_hip.synth import make_pair_batch
    from imcui_hip.synth_weights import superpoint_state_dict

    pipe = SuperPointLightGluePipeline(
        {"nms_radius": 3, "max_keypoints": 512, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": superpoint_state_dict(0)},
        {"depth_confidence": 0.95, "width_confidence": 0.99, "match_threshold": 0.1, "state_dict": LSD},
    ).eval().to("cuda:0")
    a0, a1, _ = make_pair_batch(5, 2, 240, 320, n_blobs=600)
    b0, b1, _ = make_pair_batch(6, 2, 240, 320, n_blobs=600)
    a0, a1, b0, b1 = a0.cuda(), a1.cuda(), b0.cuda(), b1.cuda()
    eager_a = {k: v.clone() for k, v in pipe(a0, a1).items()}
    eager_b = {k: v.clone() for k, v in pipe(b0, b1).items()}
    g = GraphedPipeline(pipe, a0, a1)
    for inp, ref in (((b0, b1), eager_b), ((a0, a1), eager_a)):
        out = g(*inp)
        torch.cuda.synchronize()
        for k in ("num_keypoints0", "keypoints1", "descriptors0", "matches0", "matches1", "matching_scores0", "stop", "prune0"):
            assert torch.equal(out[k], ref[k]), k
    assert (eager_a["matches0"] > -1).sum() > 20
    with pytest.raises(ValueError):
        g(a0[:1], a1[:1])


def test_two_streams_do_not_share_scratch():
    """Scratch workspaces are per (device, stream): the same plugin objects driven from two HIP streams at once (two
    Gradio worker threads) must give the results of serial calls -- with a shared workspace the second launch would
    overwrite the first one's intermediates while its kernels are still in flight."""
    from imcui_hip.pipeline import SuperPointLightGluePipeline
    from imcui_hip.synth import make_pair_batch
    from imcui_hip.synth_weights import superpoint_state_dict

    pipe = SuperPointLightGluePipeline(
        {"nms_radius": 3, "max_keypoints": 1024, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": superpoint_state_dict(0)},
        {"depth_confidence": -1.0, "width_confidence": -1.0, "match_threshold": 0.1, "state_dict": LSD},
    ).eval().to("cuda:0")
    a0, a1, _ = make_pair_batch(31, 4, 480, 640)
    b0, b1, _ = make_pair_batch(32, 4, 480, 640)
    a0, a1, b0, b1 = a0.cuda(), a1.cuda(), b0.cuda(), b1.cuda()
    keys = ("num_keypoints0", "keypoints0", "matches0", "matching_scores0", "matches1")
    ref_a = {k: v.clone() for k, v in pipe(a0, a1).items() if k in keys}
    ref_b = {k: v.clone() for k, v in pipe(b0, b1).items() if k in keys}
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):  # several rounds: the overlap is a race, give it chances
        with torch.cuda.stream(s1):
            out_a = pipe(a0, a1)
        with torch.cuda.stream(s2):
            out_b = pipe(b0, b1)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(out_a[k], ref_a[k]), f"stream 1: {k}"
            assert torch.equal(out_b[k], ref_b[k]), f"stream 2: {k}"


def test_all_keypoints_conf_is_graph_capturable():
    """max_keypoints = -1 (the plugin default) used to synchronise to read the selection status, which made HIP-graph
    capture fail; the status is a device tensor now."""
    from imcui_hip.pipeline import GraphedPipeline, SuperPointLightGluePipeline
    from imcui_hip.synth import make_pair_batch
    from imcui_hip.synth_weights import superpoint_state_dict

    pipe = SuperPointLightGluePipeline(
        {"nms_radius": 4, "max_keypoints": -1, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": superpoint_state_dict(0)},
        {"depth_confidence": 0.95, "width_confidence": 0.99, "match_threshold": 0.1, "state_dict": LSD},
    ).eval().to("cuda:0")
    a0, a1, _ = make_pair_batch(7, 1, 240, 320, n_blobs=500)
    a0, a1 = a0.cuda(), a1.cuda()
    eager = {k: v.clone() for k, v in pipe(a0, a1).items()}
    g = GraphedPipeline(pipe, a0, a1)
    out = g(a0, a1)
    torch.cuda.synchronize()
    n0, n1 = int(eager["num_keypoints0"][0]), int(eager["num_keypoints1"][0])
    assert torch.equal(out["num_keypoints0"], eager["num_keypoints0"]) and torch.equal(out["num_keypoints1"], eager["num_keypoints1"])
    assert torch.equal(out["keypoints1"][0, :n1], eager["keypoints1"][0, :n1])  # rows past the count are unwritten capacity
    for k in ("matches0", "matching_scores0"):
        assert torch.equal(out[k][0, :n0], eager[k][0, :n0]), k
    assert torch.equal(out["stop"], eager["stop"])
    assert n0 > 100


def test_pair_alone_equals_pair_in_a_batch_bitwise():
    """one pair per call runs its attention as a KEY-SPLIT launch (a chunk of 512 keys per workgroup + a combine
    kernel; csrc/attention.hip) so that a lone pair fills the chip; a batch keeps one workgroup per query block.  Both geometries evaluate
    the same chunked soft-max with the same fold, so every token state of every layer, every match and every score of a pair must be
    BITWISE the same alone (split), alone with the split switched off, and inside a batch of six (unsplit) -- with ragged counts whose
    last chunks are partial or missing."""
    from imcui_hip import backend

    dev = torch.device("cuda:0")
    backend.set_precision(dev, 1)
    sd = WEIGHTS["strong"]
    problems = [synthetic_matching_problem(60, 2048, 1900, 250), synthetic_matching_problem(61, 600, 2048, 100), synthetic_matching_problem(62, 1500, 1025, 200),
                synthetic_matching_problem(63, 2048, 2048, 300), synthetic_matching_problem(64, 513, 700, 50), synthetic_matching_problem(65, 1300, 1800, 150)]  # fmt: skip
    model = _model(-1, -1, sd=sd)

    def run(ps, cap):
        k0, k1, d0, d1 = torch.zeros(len(ps), cap, 2), torch.zeros(len(ps), cap, 2), torch.zeros(len(ps), cap, 256), torch.zeros(len(ps), cap, 256)
        n0, n1 = torch.zeros(len(ps), dtype=torch.int32), torch.zeros(len(ps), dtype=torch.int32)
        for b, (a, c, e, f) in enumerate(ps):
            k0[b, : len(a)], k1[b, : len(c)], d0[b, : len(a)], d1[b, : len(c)] = a, c, e, f
            n0[b], n1[b] = len(a), len(c)
        out = model.forward_batched(k0.cuda(), k1.cuda(), d0.cuda(), d1.cuda(), n0.cuda(), n1.cuda(), (640, 480), (640, 480), layer_dump=True)
        torch.cuda.synchronize()
        return {k: v.cpu() for k, v in out.items()}

    batch = run(problems, 2048)
    assert backend.get_option(dev, "attn_split") == 1
    for b, pr in enumerate(problems):
        alone = run([pr], 2048)
        with backend.option(dev, attn_split=0):
            alone_unsplit = run([pr], 2048)
        na, nc = len(pr[0]), len(pr[1])
        for other, tag in ((alone_unsplit, "split vs unsplit, alone"), (batch, "alone (split) vs in the batch")):
            ob = 0 if other is alone_unsplit else b
            for li in range(9):
                for s, n in ((0, na), (1, nc)):
                    assert torch.equal(alone["_layers"][li, s, :n], other["_layers"][li, 2 * ob + s, :n]), (tag, b, li, s)
            assert torch.equal(alone["matches0"][0, :na], other["matches0"][ob, :na]) and torch.equal(alone["matches1"][0, :nc], other["matches1"][ob, :nc]), (tag, b)
            if other is alone_unsplit:
                assert torch.equal(alone["matching_scores0"][0, :na], other["matching_scores0"][ob, :na]), (tag, b)
            else:
                # the ASSIGNMENT's soft-max statistics are reduced in column chunks whose number follows the batch size (simred_chunks: enough
                # workgroups to fill the chip), so the log-sum-exp of a row is associated differently alone and in a batch: last-bit differences
                # of the scores (the matches above are equal); everything upstream -- every token state -- is bitwise equal
                assert (alone["matching_scores0"][0, :na] - other["matching_scores0"][ob, :na]).abs().max().item() < 2e-6, (tag, b)
