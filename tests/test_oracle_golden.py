"""Oracle vs golden vectors produced by the reference's own code (CPU, no GPU).

tests/golden/nn_*.npz come from imcui/hloc/matchers/nearest_neighbor.py run in the
build container (tests/golden/make_golden.py).  Bit-exact indices and scores.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle.mutual_nn import mutual_nn


def _load(path):
    z = np.load(path)
    conf = {}
    for k, v in zip(z["conf_keys"], z["conf_vals"]):
        conf[str(k)] = bool(v) if str(k) == "do_mutual_check" else float(v)
    return z, conf


def golden_files():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return sorted(glob.glob(os.path.join(here, "nn_*.npz")))


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_mutual_nn_oracle_matches_reference(path):
    torch.set_num_threads(1)
    z, conf = _load(path)
    out = mutual_nn(
        {"descriptors0": torch.from_numpy(z["descriptors0"]), "descriptors1": torch.from_numpy(z["descriptors1"])}, conf
    )
    assert np.array_equal(out["matches0"].numpy(), z["matches0"])
    assert np.array_equal(out["matching_scores0"].numpy().astype(np.float32), z["matching_scores0"])


def test_golden_present():
    assert len(golden_files()) >= 6


def ds_golden_files():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return sorted(glob.glob(os.path.join(here, "ds_*.npz")))


@pytest.mark.parametrize("path", ds_golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_dual_softmax_oracle_matches_reference(path):
    """tests/golden/ds_*.npz come from imcui/hloc/matchers/dual_softmax.py (make_golden.py): indices exact;
    scores bit-exact too (same torch ops on the same host kernels), stored as the reference's float64."""
    from oracle.dual_softmax import DualSoftMaxOracle

    torch.set_num_threads(1)
    z = np.load(path)
    conf = {str(k): float(v) for k, v in zip(z["conf_keys"], z["conf_vals"])}
    out = DualSoftMaxOracle(conf)({"descriptors0": torch.from_numpy(z["descriptors0"]), "descriptors1": torch.from_numpy(z["descriptors1"])})
    empty = z["descriptors0"].shape[-1] == 0 or z["descriptors1"].shape[-1] == 0
    assert out["matches0"].dtype == torch.int64 and out["matching_scores0"].dtype == (torch.int64 if empty else torch.float64)
    assert np.array_equal(out["matches0"].numpy(), z["matches0"])
    assert np.abs(out["matching_scores0"].numpy().astype(np.float64) - z["matching_scores0"]).max(initial=0.0) <= 1e-6


def test_ds_golden_present():
    assert len(ds_golden_files()) >= 6


def coarse_golden_files():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return sorted(glob.glob(os.path.join(here, "coarse_*.npz")))


@pytest.mark.parametrize("family", ["loftr", "eloftr"])
@pytest.mark.parametrize("path", coarse_golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_coarse_matching_rule_matches_reference_dual_softmax(path, family):
    """tests/golden/coarse_*.npz: the REFERENCE's own `dual_softmax_matcher` (imcui/hloc/matchers/dual_softmax.py:8-41) run on LoFTR-style
    coarse features with `normalize=False, inv_temperature = 1 / (C * temperature)` (make_coarse_golden.py).  The dual-softmax
    confidence, the mutual-maximum test and the threshold of the LoFTR / EfficientLoFTR oracles' coarse matching (border removal
    switched off: it is not part of the reference function) must give the same matches and confidences."""
    torch.set_num_threads(1)
    z = np.load(path)
    f0, f1 = torch.from_numpy(z["feat0"]), torch.from_numpy(z["feat1"])
    (h0, w0), (h1, w1) = z["grid0"].tolist(), z["grid1"].tolist()
    thr = float(z["threshold"])
    if family == "loftr":
        from imcui_hip.synth_weights import loftr_state_dict
        from oracle.loftr import LoFTROracle

        o = LoFTROracle(loftr_state_dict(0))
        o.border_rm, o.temperature = 0, float(z["temperature"])
        cm = o.coarse_matching(f0, f1, (h0, w0), (h1, w1), (h0 * 8, w0 * 8), thr)
    else:
        from imcui_hip.synth_weights import eloftr_state_dict
        from oracle.eloftr import ELoFTROracle

        o = ELoFTROracle(eloftr_state_dict(0))
        o.border_rm, o.temperature = 0, float(z["temperature"])
        C = f0.shape[-1]
        cm = o.coarse_matching(f0.transpose(1, 2).reshape(1, C, h0, w0), f1.transpose(1, 2).reshape(1, C, h1, w1), (h0 * 8, w0 * 8), thr)
    m0 = torch.full((f0.shape[1],), -1, dtype=torch.int64)
    s0 = torch.zeros(f0.shape[1], dtype=torch.float64)
    m0[cm["i_ids"]] = cm["j_ids"]
    s0[cm["i_ids"]] = cm["mconf"].double()
    assert (z["matches0"][0] > -1).sum() >= 10
    assert np.array_equal(m0.numpy(), z["matches0"][0])
    assert np.abs(s0.numpy() - z["scores0"][0]).max() <= 1e-6


def test_coarse_golden_present():
    assert len(coarse_golden_files()) >= 3


def test_reciprocal_matcher_from_every_pixel_equals_reference_mutual_nn():
    """MASt3R's `fast_reciprocal_NNs` started from EVERY position (subsample 1) returns exactly the mutual nearest neighbours: a chain
    a -> nn(a) -> nn(nn(a)) closes at once on a mutual pair and only on one.  The reference's own mutual-NN matcher
    (imcui/hloc/matchers/nearest_neighbor.py, golden vectors nn_mutual.npz) therefore pins the oracle's restatement of the matcher."""
    from oracle.dust3r import fast_reciprocal_nns

    torch.set_num_threads(1)
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(here, "nn_mutual.npz"))
    d0, d1 = torch.from_numpy(z["descriptors0"]), torch.from_numpy(z["descriptors1"])  # [B, D, N], [B, D, M]
    for b in range(d0.shape[0]):
        a = d0[b].t().reshape(15, 20, -1)  # 300 descriptors as a 15 x 20 map
        c = d1[b].t().reshape(14, 20, -1)  # 280 as 14 x 20
        xy1, xy2 = fast_reciprocal_nns(a, c, subsample=1)
        got = {(int(y) * 20 + int(x), int(v) * 20 + int(u)) for (x, y), (u, v) in zip(xy1.tolist(), xy2.tolist())}
        want = {(i, int(j)) for i, j in enumerate(z["matches0"][b]) if j > -1}
        assert len(want) > 100 and got == want
