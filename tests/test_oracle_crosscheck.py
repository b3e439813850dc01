"""Oracle vs the independent `transformers` ports (architecture cross-check, CPU).

The reference's SuperPoint / LightGlue arithmetic is absent from /root/reference
(empty submodules) and it ships no golden vectors, so the restatement in oracle/ is
checked against the HF ports with identical seeded weights.  Known
HF-vs-upstream deltas handled here: HF SuperPoint applies border removal against an
8x too large image (only the low border is enforced) and returns relative coords.
"""
import pytest
import torch

from imcui_hip.synth import make_pair
from oracle.lightglue import LightGlueOracle
from oracle.superpoint import SuperPointOracle
from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict

transformers = pytest.importorskip("transformers")

SP_MAP = {
    "conv1a": "encoder.conv_blocks.0.conv_a",
    "conv1b": "encoder.conv_blocks.0.conv_b",
    "conv2a": "encoder.conv_blocks.1.conv_a",
    "conv2b": "encoder.conv_blocks.1.conv_b",
    "conv3a": "encoder.conv_blocks.2.conv_a",
    "conv3b": "encoder.conv_blocks.2.conv_b",
    "conv4a": "encoder.conv_blocks.3.conv_a",
    "conv4b": "encoder.conv_blocks.3.conv_b",
    "convPa": "keypoint_decoder.conv_score_a",
    "convPb": "keypoint_decoder.conv_score_b",
    "convDa": "descriptor_decoder.conv_descriptor_a",
    "convDb": "descriptor_decoder.conv_descriptor_b",
}


def _hf_superpoint(sd, **kw):
    from transformers import SuperPointConfig, SuperPointForKeypointDetection

    hf = SuperPointForKeypointDetection(SuperPointConfig(**kw)).eval()
    m = {}
    for k, v in SP_MAP.items():
        m[v + ".weight"] = sd[k + ".weight"]
        m[v + ".bias"] = sd[k + ".bias"]
    hf.load_state_dict(m, strict=True)
    return hf


@pytest.mark.parametrize("nms_radius,max_kpts", [(3, -1), (4, 300)])
def test_superpoint_oracle_vs_hf(nms_radius, max_kpts):
    torch.set_num_threads(4)
    h, w = 240, 320
    img0, _, _ = make_pair(3, h, w, n_blobs=500)
    sd = superpoint_state_dict(0)
    conf = dict(nms_radius=nms_radius, max_keypoints=max_kpts, keypoint_threshold=0.005, remove_borders=4)
    out = SuperPointOracle(sd)({"image": img0}, conf)
    hf = _hf_superpoint(
        sd, keypoint_threshold=0.005, max_keypoints=max_kpts, nms_radius=nms_radius, border_removal_distance=4
    )
    with torch.no_grad():
        ref = hf(img0.repeat(1, 3, 1, 1))
    mask = ref.mask[0].bool()
    k_hf = (ref.keypoints[0][mask] * torch.tensor([float(w), float(h)])).round()
    s_hf, d_hf = ref.scores[0][mask], ref.descriptors[0][mask]
    k_or, s_or, d_or = out["keypoints"][0], out["scores"][0], out["descriptors"][0].T
    assert len(k_or) > 100
    # HF keeps the high-border band the upstream code removes: compare on the interior
    inner = (k_hf[:, 0] < w - 4) & (k_hf[:, 1] < h - 4)
    if max_kpts < 0:
        k_hf, s_hf, d_hf = k_hf[inner], s_hf[inner], d_hf[inner]
        assert torch.equal(k_hf, k_or)
        assert torch.equal(s_hf, s_or)
        assert (d_hf - d_or).abs().max().item() < 1e-6
    else:
        # top-k is taken before/after a different border filter: compare the common points
        key = lambda k: (k[:, 1] * w + k[:, 0]).long()  # noqa: E731
        common = set(key(k_hf).tolist()) & set(key(k_or).tolist())
        assert len(common) > 0.8 * len(k_or)
        lut = {v: i for i, v in enumerate(key(k_hf).tolist())}
        for i, v in enumerate(key(k_or).tolist()):
            if v in lut:
                assert s_or[i].item() == s_hf[lut[v]].item()
                assert (d_or[i] - d_hf[lut[v]]).abs().max().item() < 1e-6


def _hf_lightglue(lsd, dc, wc, th):
    from transformers import LightGlueConfig, LightGlueForKeypointMatching, SuperPointConfig

    input_dim = lsd["input_proj.weight"].shape[1] if "input_proj.weight" in lsd else 256
    cfg = LightGlueConfig(
        keypoint_detector_config=SuperPointConfig(descriptor_decoder_dim=input_dim), depth_confidence=dc, width_confidence=wc, filter_threshold=th
    )
    hf = LightGlueForKeypointMatching(cfg).eval()
    m = {k: v for k, v in hf.state_dict().items() if k.startswith("keypoint_detector")}
    m["positional_encoder.projector.weight"] = lsd["posenc.Wr.weight"]
    if input_dim != 256:  # the port projects 128-d descriptors (disk / aliked variants) like upstream's `input_proj`
        m["input_projection.weight"], m["input_projection.bias"] = lsd["input_proj.weight"], lsd["input_proj.bias"]
    for i in range(9):
        p, q = f"transformers.{i}.", f"transformer_layers.{i}."
        W = lsd[p + "self_attn.Wqkv.weight"].view(4, 64, 3, 256)
        B = lsd[p + "self_attn.Wqkv.bias"].view(4, 64, 3)
        for t, nm in enumerate(["q_proj", "k_proj", "v_proj"]):
            m[q + f"self_attention.{nm}.weight"] = W[:, :, t, :].reshape(256, 256)
            m[q + f"self_attention.{nm}.bias"] = B[:, :, t].reshape(256)
        m[q + "self_attention.o_proj.weight"] = lsd[p + "self_attn.out_proj.weight"]
        m[q + "self_attention.o_proj.bias"] = lsd[p + "self_attn.out_proj.bias"]
        for blk, src in [("self_mlp", "self_attn"), ("cross_mlp", "cross_attn")]:
            for a, b in [("fc1", "ffn.0"), ("layer_norm", "ffn.1"), ("fc2", "ffn.3")]:
                m[q + f"{blk}.{a}.weight"] = lsd[p + f"{src}.{b}.weight"]
                m[q + f"{blk}.{a}.bias"] = lsd[p + f"{src}.{b}.bias"]
        for nm in ["q_proj", "k_proj"]:
            m[q + f"cross_attention.{nm}.weight"] = lsd[p + "cross_attn.to_qk.weight"]
            m[q + f"cross_attention.{nm}.bias"] = lsd[p + "cross_attn.to_qk.bias"]
        for a, b in [("v_proj", "to_v"), ("o_proj", "to_out")]:
            m[q + f"cross_attention.{a}.weight"] = lsd[p + f"cross_attn.{b}.weight"]
            m[q + f"cross_attention.{a}.bias"] = lsd[p + f"cross_attn.{b}.bias"]
        a, b = f"log_assignment.{i}.", f"match_assignment_layers.{i}."
        for x, y in [("final_projection", "final_proj"), ("matchability", "matchability")]:
            m[b + x + ".weight"] = lsd[a + y + ".weight"]
            m[b + x + ".bias"] = lsd[a + y + ".bias"]
        if i < 8:
            m[f"token_confidence.{i}.token.weight"] = lsd[f"token_confidence.{i}.token.0.weight"]
            m[f"token_confidence.{i}.token.bias"] = lsd[f"token_confidence.{i}.token.0.bias"]
    hf.load_state_dict(m, strict=True)
    return hf


def synthetic_matching_problem(seed, n, m, n_out, noise=0.05, dim=256):
    """Keypoints/descriptors with known correspondences (distinctive random descriptors)."""
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(seed)
    k0 = torch.rand(n, 2, generator=g) * torch.tensor([632.0, 472.0]) + 4
    perm = torch.randperm(n, generator=g)[:m]
    k1 = k0[perm] + torch.randn(m, 2, generator=g)
    d0 = F.normalize(torch.randn(n, dim, generator=g), dim=1)
    d1 = F.normalize(d0[perm] + noise * torch.randn(m, dim, generator=g), dim=1)
    d1[:n_out] = F.normalize(torch.randn(n_out, dim, generator=g), dim=1)
    img = torch.zeros(1, 1, 480, 640)
    return {
        "image0": img,
        "image1": img,
        "keypoints0": k0[None],
        "keypoints1": k1[None],
        "descriptors0": d0.T[None].contiguous(),
        "descriptors1": d1.T[None].contiguous(),
    }


# Only these two modes run in the HF port: it crashes when exactly one of early-stop / pruning is on.
@pytest.mark.parametrize("input_dim", [256, 128])
@pytest.mark.parametrize("dc,wc", [(-1.0, -1.0), (0.95, 0.99)])
def test_lightglue_oracle_vs_hf(dc, wc, input_dim):
    torch.set_num_threads(4)
    lsd = lightglue_state_dict(0, input_dim=input_dim)
    data = synthetic_matching_problem(7, 400, 350, 100, dim=input_dim)
    out = LightGlueOracle(lsd, dict(depth_confidence=dc, width_confidence=wc, filter_threshold=0.1))(data)
    hf = _hf_lightglue(lsd, dc, wc, 0.1)
    n0, n1 = data["keypoints0"].shape[1], data["keypoints1"].shape[1]
    N = max(n0, n1)
    kp = torch.zeros(1, 2, N, 2)
    de = torch.zeros(1, 2, N, input_dim)
    mask = torch.zeros(1, 2, N, dtype=torch.int)
    kp[0, 0, :n0], kp[0, 1, :n1] = data["keypoints0"][0], data["keypoints1"][0]
    de[0, 0, :n0], de[0, 1, :n1] = data["descriptors0"][0].T, data["descriptors1"][0].T
    mask[0, 0, :n0] = 1
    mask[0, 1, :n1] = 1
    with torch.no_grad():
        matches, mscores, prune, _, _ = hf._match_image_pair(kp, de, 480, 640, mask=mask)
    matches, mscores, prune = matches.reshape(1, 2, -1), mscores.reshape(1, 2, -1), prune.reshape(1, 2, -1)
    assert (out["matches0"] > -1).sum() > 50
    assert torch.equal(matches[0, 0, :n0].long(), out["matches0"][0])
    assert torch.equal(matches[0, 1, :n1].long(), out["matches1"][0])
    assert (mscores[0, 0, :n0] - out["matching_scores0"][0]).abs().max().item() < 2e-5
    assert torch.equal(prune[0, 0, :n0].long(), out["prune0"][0].long())


def _hf_superglue(sd, iters):
    """HF keeps the heads contiguous (channel = head * 64 + d); upstream interleaves them (d * 4 + head)."""
    from transformers import SuperGlueConfig, SuperGlueForKeypointMatching

    hf = SuperGlueForKeypointMatching(SuperGlueConfig(sinkhorn_iterations=iters, matching_threshold=0.0)).eval()
    m = {k: v for k, v in hf.state_dict().items() if k.startswith("keypoint_detector")}
    perm = torch.tensor([d * 4 + h for h in range(4) for d in range(64)])  # HF channel -> upstream channel
    for i in range(5):
        src = f"kenc.encoder.{3 * i}"
        dst = f"keypoint_encoder.encoder.{i}" + (".linear" if i < 4 else "")
        m[dst + ".weight"], m[dst + ".bias"] = sd[src + ".weight"][:, :, 0], sd[src + ".bias"]
        if i < 4:
            for f in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
                m[f"keypoint_encoder.encoder.{i}.batch_norm.{f}"] = sd[f"kenc.encoder.{3 * i + 1}.{f}"]
    for i in range(18):
        p, q = f"gnn.layers.{i}.", f"gnn.layers.{i}."
        for j, nm in enumerate(["query", "key", "value"]):
            m[q + f"attention.self.{nm}.weight"] = sd[p + f"attn.proj.{j}.weight"][:, :, 0][perm]
            m[q + f"attention.self.{nm}.bias"] = sd[p + f"attn.proj.{j}.bias"][perm]
        m[q + "attention.output.dense.weight"] = sd[p + "attn.merge.weight"][:, :, 0][:, perm]
        m[q + "attention.output.dense.bias"] = sd[p + "attn.merge.bias"]
        m[q + "mlp.0.linear.weight"], m[q + "mlp.0.linear.bias"] = sd[p + "mlp.0.weight"][:, :, 0], sd[p + "mlp.0.bias"]
        for f in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            m[q + f"mlp.0.batch_norm.{f}"] = sd[p + f"mlp.1.{f}"]
        m[q + "mlp.1.weight"], m[q + "mlp.1.bias"] = sd[p + "mlp.3.weight"][:, :, 0], sd[p + "mlp.3.bias"]
    m["final_projection.final_proj.weight"] = sd["final_proj.weight"][:, :, 0]
    m["final_projection.final_proj.bias"] = sd["final_proj.bias"]
    m["bin_score"] = sd["bin_score"]
    hf.load_state_dict(m, strict=True)
    return hf


@pytest.mark.parametrize("iters", [5, 50])
def test_superglue_oracle_vs_hf(iters):
    """Same seeded weights through the restatement and through the independent HF port (equal key-point
    counts: the HF port stacks the two images)."""
    from oracle.superglue import SuperGlueOracle
    from imcui_hip.synth_weights import superglue_state_dict

    torch.set_num_threads(4)
    sd = superglue_state_dict(0)
    data = synthetic_matching_problem(11, 300, 300, 80)
    g = torch.Generator().manual_seed(5)
    data["scores0"], data["scores1"] = torch.rand(1, 300, generator=g), torch.rand(1, 300, generator=g)
    out = SuperGlueOracle(sd, {"sinkhorn_iterations": iters, "match_threshold": 0.0})(data)
    hf = _hf_superglue(sd, iters)
    kp = torch.stack([data["keypoints0"], data["keypoints1"]], 1)
    de = torch.stack([data["descriptors0"].transpose(1, 2), data["descriptors1"].transpose(1, 2)], 1)
    sc = torch.stack([data["scores0"], data["scores1"]], 1)
    with torch.no_grad():
        matches, mscores, _, _ = hf._match_image_pair(kp, de, sc, 480, 640)
    assert (out["matches0"][0] > -1).sum() > 150
    assert torch.equal(matches[0, 0].long(), out["matches0"][0].long())
    assert torch.equal(matches[0, 1].long(), out["matches1"][0].long())
    assert (mscores[0, 0] - out["matching_scores0"][0]).abs().max().item() < 2e-5
    assert (mscores[0, 1] - out["matching_scores1"][0]).abs().max().item() < 2e-5


def test_eloftr_oracle_vs_hf_port():
    """oracle/eloftr.py against `transformers.EfficientLoFTRForKeypointMatching` with the same seeded weights: backbone
    features, transformed coarse features, coarse matches (the port's per-row form) and the unfolded fine windows.  The
    port's fine MATCHING is not compared: it applies the two soft-maxes over the (key-point, window-0) axes of its 4-D
    tensors, which equals the upstream per-match soft-max only for a single match, and it pairs windows by list position."""
    import torch.nn.functional as F
    from transformers import EfficientLoFTRConfig, EfficientLoFTRForKeypointMatching

    from imcui_hip.synth import make_shifted_pair
    from imcui_hip.synth_weights import eloftr_state_dict
    from oracle.eloftr import ELoFTROracle

    torch.set_num_threads(4)
    sd = eloftr_state_dict(3)
    hf = EfficientLoFTRForKeypointMatching(EfficientLoFTRConfig()).eval()
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    i0, i1, _ = make_shifted_pair(4, 160, 224, (16, 8), 400)
    out = ELoFTROracle(sd).net(i0, i1, True)
    assert len(out["confidence"]) > 100  # the shaped weights produce a dense match list on an aligned shift
    x = torch.stack([i0, i1], 1).expand(-1, -1, 3, -1, -1).contiguous()
    with torch.no_grad():
        bo = hf.efficientloftr(x)
        fc = bo.feature_maps[0]
        assert torch.equal(bo.feature_maps[1], torch.cat(out["_x1"], 0)) and torch.equal(bo.feature_maps[2], torch.cat(out["_x2"], 0))
        scale = fc.abs().max().item()
        assert (fc[:, 0] - out["_feat_c0"]).abs().max().item() < 2e-5 * scale
        assert (fc[:, 1] - out["_feat_c1"]).abs().max().item() < 2e-5 * scale
        _, sc, mi = hf._coarse_matching(fc, 8.0)
        rows = (sc[0, 1] > 0).nonzero()[:, 0]
        assert torch.equal(rows, out["_i_ids"]) and torch.equal(mi[0, 1][rows], out["_j_ids"])
        assert (sc[0, 1][rows] - out["confidence"]).abs().max().item() < 1e-4
        f0, f1 = hf.refinement_layer(fc / 16.0, bo.feature_maps[1:])
    ff0, ff1 = out["_fine"]
    u0 = F.unfold(ff0, 8, stride=8).view(1, 64, 64, -1).permute(0, 3, 2, 1)
    u1 = F.unfold(ff1, 10, stride=8, padding=1).view(1, 64, 100, -1).permute(0, 3, 2, 1)
    fs = f0.abs().max().item()
    assert (u0 - f0).abs().max().item() < 2e-5 * fs and (u1 - f1).abs().max().item() < 2e-5 * fs
    # geometry: the matches agree with the known translation (fine stage included)
    e = (out["keypoints0"] - out["keypoints1"] - torch.tensor([16.0, 8.0])).norm(dim=1)
    assert (e < 2).float().mean().item() > 0.9
    # images of different sizes: a narrower second image (a crop of the same scene) still matches cell to cell
    out2 = ELoFTROracle(sd).net(i0, i1[..., :128, :192].contiguous(), True)
    e2 = (out2["keypoints0"] - out2["keypoints1"] - torch.tensor([16.0, 8.0])).norm(dim=1)
    assert len(e2) > 50 and (e2 < 2).float().mean().item() > 0.8


def test_eloftr_reparameterisation_and_packing():
    """Host logic of the HIP path: the folded RepVGG blocks (one 3x3 convolution each) reproduce the three-branch
    backbone of the oracle, and the packer accepts the state dict (layer table shapes checked inside)."""
    import torch.nn.functional as F

    from imcui_hip import backend
    from imcui_hip.synth_weights import eloftr_state_dict
    from oracle.eloftr import ELoFTROracle

    sd = eloftr_state_dict(1)
    packed = backend.pack_eloftr(sd)
    assert packed.dtype == torch.float32 and packed.numel() == backend.load_library().imcui_hip_eloftr_packed_floats()
    orc = ELoFTROracle(sd)
    x = torch.rand(2, 1, 64, 96, generator=torch.Generator().manual_seed(0))
    _, _, x3 = orc.backbone(x)
    y = x
    for s, nb, st in ((0, 1, 2), (1, 2, 1), (2, 4, 2), (3, 14, 2)):
        for b in range(nb):
            w, bias = backend._repvgg_reparam(orc.sd, f"efficientloftr.backbone.stages.{s}.blocks.{b}")
            y = F.relu(F.conv2d(y, w, bias, st if b == 0 else 1, 1))
    assert (y - x3).abs().max().item() < 2e-5 * x3.abs().max().item()
    from imcui_hip.hloc.matchers.eloftr import to_port_names

    # upstream-named checkpoints are mapped (tests/test_host_cpu.py round-trips the map); anything else is refused
    assert "efficientloftr.backbone.stages.0.blocks.0.conv1.conv.weight" in to_port_names({"matcher.backbone.layer0.rbr_dense.conv.weight": torch.zeros(1)})
    with pytest.raises(KeyError):
        to_port_names({"net.conv.weight": torch.zeros(1)})
