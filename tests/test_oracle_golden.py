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
