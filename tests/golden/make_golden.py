"""Generate golden vectors by running the REFERENCE's own code (build container only).

/root/reference is importable only where it is mounted; this script is committed
together with its outputs (tests/golden/nn_*.npz) so the GPU box never needs it.
Only `imcui.hloc.matchers.nearest_neighbor` and `imcui.hloc.matchers.dual_softmax` can run (SURVEY.md
section 8c): the SuperPoint / LightGlue / LoFTR arithmetic lives in absent submodules.

Run from a scratch CWD (importing imcui.hloc truncates ./log.txt):
    cd /tmp && python /root/repo/tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

from imcui.hloc.matchers.dual_softmax import DualSoftMax  # noqa: E402
from imcui.hloc.matchers.nearest_neighbor import NearestNeighbor  # noqa: E402


def unit(g, b, d, n):
    x = torch.randn(b, d, n, generator=g)
    return x / x.norm(dim=1, keepdim=True)


CASES = {
    # name: (B, D, N, M, conf)
    "nn_mutual": (2, 128, 300, 280, {"do_mutual_check": True}),
    "nn_ratio": (1, 128, 257, 333, {"ratio_threshold": 0.9, "do_mutual_check": True}),
    "nn_dist": (1, 256, 200, 190, {"distance_threshold": 1.2, "do_mutual_check": True}),
    "nn_ratio_dist_nomutual": (1, 64, 129, 65, {"ratio_threshold": 0.95, "distance_threshold": 1.3, "do_mutual_check": False}),
    "nn_single": (1, 128, 50, 1, {"ratio_threshold": 0.8, "do_mutual_check": True}),
    "nn_empty": (1, 128, 40, 0, {"do_mutual_check": True}),
}

if __name__ == "__main__":
    torch.set_num_threads(1)
    for i, (name, (b, d, n, m, conf)) in enumerate(CASES.items()):
        g = torch.Generator().manual_seed(100 + i)
        d0 = unit(g, b, d, n)
        # correlated second set so that mutual matches exist
        perm = torch.randperm(max(n, 1), generator=g)[:m] if m <= n else torch.arange(m) % max(n, 1)
        d1 = d0[:, :, perm] + 0.25 * torch.randn(b, d, m, generator=g) if m > 0 else torch.zeros(b, d, 0)
        if m > 0:
            d1 = d1 / d1.norm(dim=1, keepdim=True)
        model = NearestNeighbor(conf).eval()
        with torch.no_grad():
            out = model({"descriptors0": d0, "descriptors1": d1})
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            descriptors0=d0.numpy(),
            descriptors1=d1.numpy(),
            matches0=out["matches0"].numpy(),
            matching_scores0=out["matching_scores0"].numpy().astype(np.float32),
            conf_keys=np.array(list(conf.keys())),
            conf_vals=np.array([float(v) for v in conf.values()]),
        )
        print(name, "matches", int((out["matches0"] > -1).sum()), "of", n)

    # ---- dual-softmax matcher (imcui/hloc/matchers/dual_softmax.py; batch 1 as hloc drives it)
    DS_CASES = {
        # name: (D, N, M, noise, conf)
        "ds_default": (256, 300, 280, 0.25, {}),
        "ds_disk128": (128, 257, 333, 0.35, {"match_threshold": 0.1}),
        "ds_temp": (64, 129, 65, 0.2, {"match_threshold": 0.3, "inv_temperature": 10}),
        "ds_unnormalised_inputs": (128, 200, 190, 0.3, {"match_threshold": 0.05}),
        "ds_single": (128, 50, 1, 0.1, {}),
        "ds_empty": (128, 40, 0, 0.0, {}),
    }
    for i, (name, (d, n, m, noise, conf)) in enumerate(DS_CASES.items()):
        g = torch.Generator().manual_seed(500 + i)
        d0 = unit(g, 1, d, n)
        perm = torch.randperm(max(n, 1), generator=g)[:m] if m <= n else torch.arange(m) % max(n, 1)
        d1 = d0[:, :, perm] + noise * torch.randn(1, d, m, generator=g) if m > 0 else torch.zeros(1, d, 0)
        if name == "ds_unnormalised_inputs":  # the matcher normalises itself
            d0 = d0 * (0.5 + torch.rand(1, 1, n, generator=g))
            d1 = d1 * (0.5 + torch.rand(1, 1, m, generator=g))
        model = DualSoftMax(conf).eval()
        with torch.no_grad():
            out = model({"descriptors0": d0, "descriptors1": d1})
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            descriptors0=d0.numpy(),
            descriptors1=d1.numpy(),
            matches0=out["matches0"].numpy().astype(np.int64),
            matching_scores0=out["matching_scores0"].numpy().astype(np.float64),
            conf_keys=np.array(list(conf.keys())),
            conf_vals=np.array([float(v) for v in conf.values()]),
        )
        print(name, "matches", int((out["matches0"] > -1).sum()), "of", n)
