"""Properties of the SuperGlue oracle's optimal-transport stage (oracle/superglue.py), independent of any other
implementation: the log-domain Sinkhorn with dust-bins must produce the marginals the algorithm prescribes."""
import torch

from oracle.superglue import SuperGlueOracle, log_optimal_transport, normalize_keypoints
from imcui_hip.synth_weights import superglue_state_dict


def test_transport_marginals():
    g = torch.Generator().manual_seed(0)
    m, n = 37, 52
    scores = torch.randn(2, m, n, generator=g) * 3
    alpha = torch.tensor(1.5)
    Z = log_optimal_transport(scores, alpha, iters=100)
    assert Z.shape == (2, m + 1, n + 1)
    P = Z.exp()  # probabilities multiplied by m + n
    # the last update normalises the columns exactly: every key-point column carries mass 1, the dust-bin column m
    assert torch.allclose(P[:, :, :-1].sum(1), torch.ones(2, n), atol=1e-4)
    assert torch.allclose(P[:, :, -1].sum(1), torch.full((2,), float(m)), rtol=1e-4)
    # after 100 rounds the rows have converged too: mass 1 per key-point row, n for the dust-bin row
    assert torch.allclose(P[:, :-1, :].sum(2), torch.ones(2, m), atol=2e-3)
    assert torch.allclose(P[:, -1, :].sum(1), torch.full((2,), float(n)), rtol=2e-3)
    # zero rounds: the couplings are returned shifted by log(m + n) only
    Z0 = log_optimal_transport(scores, alpha, iters=0)
    assert torch.allclose(Z0[:, :m, :n], scores + torch.log(torch.tensor(float(m + n))), atol=1e-5)
    assert torch.allclose(Z0[:, -1, :], alpha + torch.log(torch.tensor(float(m + n))).expand(2, n + 1), atol=1e-5)


def test_keypoint_normalisation_and_empty_input():
    k = torch.tensor([[[0.0, 0.0], [320.0, 240.0], [640.0, 480.0]]])
    kn = normalize_keypoints(k, (1, 1, 480, 640))
    assert torch.allclose(kn[0, 1], torch.zeros(2)) and torch.allclose(kn[0, 2], torch.tensor([320.0, 240.0]) / (640 * 0.7))
    sg = SuperGlueOracle(superglue_state_dict(1), {"sinkhorn_iterations": 3})
    img = torch.zeros(1, 1, 48, 64)
    out = sg({"image0": img, "image1": img, "keypoints0": torch.zeros(1, 4, 2), "keypoints1": torch.zeros(1, 0, 2), "scores0": torch.zeros(1, 4),
              "scores1": torch.zeros(1, 0), "descriptors0": torch.zeros(1, 256, 4), "descriptors1": torch.zeros(1, 256, 0)})  # fmt: skip
    assert out["matches0"].dtype == torch.int32 and (out["matches0"] == -1).all() and out["matching_scores1"].shape == (1, 0)


def test_matches_are_a_partial_assignment_and_swap_symmetric():
    """A column is matched by at most one row, matches1 is the inverse map, and swapping the two images swaps the
    outputs (the network treats both sides with the same weights)."""
    import torch.nn.functional as F

    torch.set_num_threads(4)
    g = torch.Generator().manual_seed(3)
    n0, n1 = 90, 110
    d0 = F.normalize(torch.randn(1, 256, n0, generator=g), dim=1)
    d1 = F.normalize(torch.cat([d0[:, :, :60] + 0.05 * torch.randn(1, 256, 60, generator=g), torch.randn(1, 256, n1 - 60, generator=g)], 2), dim=1)
    k0 = torch.rand(1, n0, 2, generator=g) * torch.tensor([640.0, 480.0])
    k1 = torch.rand(1, n1, 2, generator=g) * torch.tensor([640.0, 480.0])
    s0, s1 = torch.rand(1, n0, generator=g), torch.rand(1, n1, generator=g)
    img = torch.zeros(1, 1, 480, 640)
    sg = SuperGlueOracle(superglue_state_dict(0), {"sinkhorn_iterations": 20})
    a = sg({"image0": img, "image1": img, "keypoints0": k0, "keypoints1": k1, "scores0": s0, "scores1": s1, "descriptors0": d0, "descriptors1": d1})
    b = sg({"image0": img, "image1": img, "keypoints0": k1, "keypoints1": k0, "scores0": s1, "scores1": s0, "descriptors0": d1, "descriptors1": d0})
    m0, m1 = a["matches0"][0], a["matches1"][0]
    v = m0[m0 > -1]
    assert len(v) >= 40 and len(torch.unique(v)) == len(v)
    assert torch.equal(m1[v], torch.where(m0 > -1)[0])
    assert torch.equal(b["matches0"][0], m1) and torch.equal(b["matches1"][0], m0)
    # scores only agree up to the convergence of the 20 rounds: a round normalises rows first and columns last
    assert (b["matching_scores0"][0] - a["matching_scores1"][0]).abs().max().item() < 0.1
