"""HIP path vs the INDEPENDENT `transformers` ports (run on the host CPU of the GPU box), with the same seeded weights:
a second reference beside oracle/ -- the ports were written by other people from the same upstream code.
LightGlue: the port's two runnable modes (tests/test_oracle_crosscheck.py).  SuperGlue: equal key-point counts
(the port stacks the two images), threshold 0 (the port zeroes sub-threshold scores, upstream does not)."""
import pytest
import torch

from test_oracle_crosscheck import _hf_lightglue, _hf_superglue, synthetic_matching_problem

pytestmark = pytest.mark.gpu
transformers = pytest.importorskip("transformers")


@pytest.mark.parametrize("dc,wc", [(-1.0, -1.0), (0.95, 0.99)])
def test_lightglue_hip_vs_hf_port(dc, wc):
    from imcui_hip.hloc.matchers.lightglue import LightGlue
    from imcui_hip.synth_weights import lightglue_state_dict

    torch.set_num_threads(8)
    lsd = lightglue_state_dict(0)
    data = synthetic_matching_problem(7, 400, 350, 100)
    model = LightGlue({"depth_confidence": dc, "width_confidence": wc, "match_threshold": 0.1, "state_dict": lsd}).eval().to("cuda:0")
    gpu = {k: v.cuda() for k, v in data.items()}
    gpu["scores0"], gpu["scores1"] = torch.ones(1, 400).cuda(), torch.ones(1, 350).cuda()
    with torch.no_grad():
        pred = model(gpu)
    hf = _hf_lightglue(lsd, dc, wc, 0.1)
    n0, n1 = 400, 350
    kp, de, mask = torch.zeros(1, 2, n0, 2), torch.zeros(1, 2, n0, 256), torch.zeros(1, 2, n0, dtype=torch.int)
    kp[0, 0, :n0], kp[0, 1, :n1] = data["keypoints0"][0], data["keypoints1"][0]
    de[0, 0, :n0], de[0, 1, :n1] = data["descriptors0"][0].T, data["descriptors1"][0].T
    mask[0, 0, :n0] = 1
    mask[0, 1, :n1] = 1
    with torch.no_grad():
        matches, mscores, prune, _, _ = hf._match_image_pair(kp, de, 480, 640, mask=mask)
    matches, mscores, prune = matches.reshape(1, 2, -1), mscores.reshape(1, 2, -1), prune.reshape(1, 2, -1)
    assert (pred["matches0"] > -1).sum() > 50
    assert torch.equal(matches[0, 0, :n0].long(), pred["matches0"][0].cpu())
    assert torch.equal(matches[0, 1, :n1].long(), pred["matches1"][0].cpu())
    assert (mscores[0, 0, :n0] - pred["matching_scores0"][0].cpu()).abs().max().item() < 1e-4
    assert torch.equal(prune[0, 0, :n0].long(), pred["prune0"][0].cpu())


@pytest.mark.parametrize("iters", [5, 50])
def test_superglue_hip_vs_hf_port(iters):
    from imcui_hip.hloc.matchers.superglue import SuperGlue
    from imcui_hip.synth_weights import superglue_state_dict

    torch.set_num_threads(8)
    sd = superglue_state_dict(0)
    data = synthetic_matching_problem(11, 300, 300, 80)
    g = torch.Generator().manual_seed(5)
    data["scores0"], data["scores1"] = torch.rand(1, 300, generator=g), torch.rand(1, 300, generator=g)
    model = SuperGlue({"sinkhorn_iterations": iters, "match_threshold": 0.0, "state_dict": sd}).eval().to("cuda:0")
    with torch.no_grad():
        pred = model({k: v.cuda() for k, v in data.items()})
    hf = _hf_superglue(sd, iters)
    kp = torch.stack([data["keypoints0"], data["keypoints1"]], 1)
    de = torch.stack([data["descriptors0"].transpose(1, 2), data["descriptors1"].transpose(1, 2)], 1)
    sc = torch.stack([data["scores0"], data["scores1"]], 1)
    with torch.no_grad():
        matches, mscores, _, _ = hf._match_image_pair(kp, de, sc, 480, 640)
    assert (pred["matches0"] > -1).sum() > 150
    assert torch.equal(matches[0, 0].long(), pred["matches0"][0].cpu())
    assert torch.equal(matches[0, 1].long(), pred["matches1"][0].cpu())
    assert (mscores[0, 0] - pred["matching_scores0"][0].cpu()).abs().max().item() < 1e-4
    assert (mscores[0, 1] - pred["matching_scores1"][0].cpu()).abs().max().item() < 1e-4
