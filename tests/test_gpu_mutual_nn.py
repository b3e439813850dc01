"""HIP mutual-NN vs the reference-generated golden vectors and the oracle (GPU box only)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _conf(z):
    conf = {}
    for k, v in zip(z["conf_keys"], z["conf_vals"]):
        conf[str(k)] = bool(v) if str(k) == "do_mutual_check" else float(v)
    return conf


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "nn_*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_nn_plugin_vs_reference_golden(path, precision):
    from imcui_hip.hloc.matchers.nearest_neighbor import NearestNeighbor

    z = np.load(path)
    model = NearestNeighbor(_conf(z)).eval().to("cuda:0")
    d0 = torch.from_numpy(z["descriptors0"]).cuda()
    d1 = torch.from_numpy(z["descriptors1"]).cuda()
    with torch.no_grad():
        out = model({"descriptors0": d0, "descriptors1": d1})
    m0 = out["matches0"].cpu().numpy()
    assert m0.shape == z["matches0"].shape
    s0 = out["matching_scores0"].cpu().numpy().astype(np.float32)
    same = m0 == z["matches0"]
    _audit_nn_rows(z["descriptors0"], z["descriptors1"], _conf(z), np.argwhere(~same))
    if z["matching_scores0"].size:
        assert np.abs(s0 - z["matching_scores0"])[same].max() < 1e-5


def _audit_nn_rows(d0, d1, conf, bad):
    """A row whose match differs from the reference's must be a near-tie of one of the reference's own decisions
    (imcui/hloc/matchers/nearest_neighbor.py:6-24), evaluated here in float64: nearest vs second-nearest (arg-max and
    mutual check), the ratio test, or the distance threshold, decided by a margin below 1e-5.  At most two rows."""
    assert len(bad) <= 2, f"{len(bad)} mismatching rows"
    if len(bad) == 0:
        return
    sim = np.einsum("bdn,bdm->bnm", d0.astype(np.float64), d1.astype(np.float64))
    for b, i in bad:
        row = np.sort(sim[b, i])[::-1]
        j = int(np.argmax(sim[b, i]))
        col = np.sort(sim[b, :, j])[::-1]
        margins = [row[0] - row[1] if len(row) > 1 else np.inf, col[0] - col[1] if len(col) > 1 else np.inf]
        for vals in (row, col):  # the thresholds apply to both directions (the mutual check runs find_nn on sim^T)
            dist = 2 * (1 - vals[:2])
            if conf.get("ratio_threshold") and len(vals) > 1:
                margins.append(abs(dist[0] - conf["ratio_threshold"] ** 2 * dist[1]))
            if conf.get("distance_threshold"):
                margins.append(abs(dist[0] - conf["distance_threshold"] ** 2))
        assert min(margins) < 1e-5, f"row ({b},{i}): deciding margins {margins} -- not a tie"


def test_nn_large_vs_oracle(precision):
    from imcui_hip import backend
    from oracle.mutual_nn import mutual_nn

    g = torch.Generator().manual_seed(3)
    d0 = torch.nn.functional.normalize(torch.randn(2, 256, 2048, generator=g), dim=1)
    d1 = torch.nn.functional.normalize(d0[:, :, torch.randperm(2048, generator=g)[:1900]] + 0.3 * torch.randn(2, 256, 1900, generator=g), dim=1)
    ref = mutual_nn({"descriptors0": d0, "descriptors1": d1}, {"ratio_threshold": 0.95})
    m0, s0 = backend.mutual_nn(d0.permute(0, 2, 1).cuda(), d1.permute(0, 2, 1).cuda(), 0.95, None, True)
    same = (m0.cpu().long() == ref["matches0"]).numpy()
    _audit_nn_rows(d0.numpy(), d1.numpy(), {"ratio_threshold": 0.95}, np.argwhere(~same))
    assert (s0.cpu() - ref["matching_scores0"]).abs().numpy()[same].max() < 1e-5


def test_nn_exact_ties_resolve_to_the_lowest_index(precision):
    """Duplicated descriptors give bit-equal similarities; `find_nn`'s arg-max on the CPU reference returns the FIRST maximum, and so must
    every stage of the fused reduction (32-column segments, 128-column tiles, 64-row halves, the fold kernels): duplicates are placed so
    that the tied candidates fall into different segments, different tiles and different row groups.  No audit: the matches are equal."""
    from imcui_hip import backend
    from oracle.mutual_nn import mutual_nn

    g = torch.Generator().manual_seed(11)
    N, M, D = 700, 900, 128
    d0 = torch.nn.functional.normalize(torch.randn(1, D, N, generator=g), dim=1)
    d1 = torch.nn.functional.normalize(torch.randn(1, D, M, generator=g), dim=1)
    # columns (image-1 descriptors) duplicated across segments / tiles: j and j + 37, j + 129, j + 514 are copies of the true partner of row j
    for j in range(0, 60):
        d1[0, :, j] = torch.nn.functional.normalize(d0[0, :, j] + 0.05 * torch.randn(D, generator=g), dim=0)
        for off in (37, 129, 514):
            d1[0, :, j + 60 + off] = d1[0, :, j]
    # rows (image-0 descriptors) duplicated across row groups / halves / tiles: the column's best row is then tied
    for i in range(100, 140):
        for off in (1, 8, 64, 200, 400):
            d0[0, :, i + 40 * (1 + (off % 7)) + off] = d0[0, :, i]
    for conf in ({"ratio_threshold": None, "distance_threshold": None, "do_mutual_check": True}, {"ratio_threshold": 0.99, "distance_threshold": None, "do_mutual_check": False}):
        ref = mutual_nn({"descriptors0": d0, "descriptors1": d1}, conf)
        m0, s0 = backend.mutual_nn(d0.permute(0, 2, 1).cuda(), d1.permute(0, 2, 1).cuda(), conf["ratio_threshold"], conf["distance_threshold"], conf["do_mutual_check"])
        assert torch.equal(m0.cpu().long(), ref["matches0"].long()), (conf, (m0.cpu().long() != ref["matches0"].long()).nonzero()[:8])
        assert (s0.cpu() - ref["matching_scores0"]).abs().max().item() < 1e-5
