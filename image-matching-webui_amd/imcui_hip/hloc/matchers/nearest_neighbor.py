"""Mutual nearest-neighbour matcher plugin on the MI355X HIP backend.

Drop-in for imcui/hloc/matchers/nearest_neighbor.py (module name `nearest_neighbor`, same
`default_conf` :31-35 and `required_inputs` :36); `_forward` (:38-66) runs in libimcui_hip
(imcui_hip_mutual_nn: MFMA similarity + fused best-two / mutual check).
"""
from __future__ import annotations

import torch

from ... import backend
from ..utils.base_model import BaseModel


class NearestNeighbor(BaseModel):
    default_conf = {
        "ratio_threshold": None,
        "distance_threshold": None,
        "do_mutual_check": True,
    }
    required_inputs = ["descriptors0", "descriptors1"]

    def _init(self, conf):
        pass

    def _forward(self, data):
        d0, d1 = data["descriptors0"], data["descriptors1"]  # [B, D, N], [B, D, M]
        if d0.size(-1) == 0 or d1.size(-1) == 0:
            matches0 = torch.full(d0.shape[:2], -1, device=d0.device)
            return {"matches0": matches0, "matching_scores0": torch.zeros_like(matches0)}
        m0, s0 = backend.mutual_nn_dn(  # (the [B, D, N] tensors as they come: transposed on the device inside the C call)
            d0,
            d1,
            self.conf["ratio_threshold"],
            self.conf["distance_threshold"],
            self.conf["do_mutual_check"],
        )
        return {"matches0": m0.long(), "matching_scores0": s0}
