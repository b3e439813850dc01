"""Dual-softmax matcher plugin on the MI355X HIP backend.

Drop-in for imcui/hloc/matchers/dual_softmax.py (module name `dual_softmax`, the matcher behind the zoo's
`disk+dualsoftmax` / `superpoint+dualsoftmax`): same `default_conf` (:45-48) and `required_inputs` (:50);
`_forward` (:55-75) runs in libimcui_hip (imcui_hip_dual_softmax: MFMA similarity + row / column soft-max
statistics + mutual-maximum test).  Output dtypes follow the reference: `matches0` int64, `matching_scores0`
float64 (it builds them from numpy defaults, :33-39); its empty-input answer (:56-65) is reproduced as is.
"""
from __future__ import annotations

import torch

from ... import backend
from ..utils.base_model import BaseModel


class DualSoftMax(BaseModel):
    default_conf = {
        "match_threshold": 0.2,
        "inv_temperature": 20,
    }
    required_inputs = ["descriptors0", "descriptors1"]  # B x DIM x M

    def _init(self, conf):
        pass

    def _forward(self, data):
        d0, d1 = data["descriptors0"], data["descriptors1"]
        if d0.size(-1) == 0 or d1.size(-1) == 0:
            matches0 = torch.full(d0.shape[:2], -1, device=d0.device)
            return {"matches0": matches0, "matching_scores0": torch.zeros_like(matches0)}
        m0, s0 = backend.dual_softmax(d0, d1, self.conf["match_threshold"], self.conf["inv_temperature"])
        return {"matches0": m0.long(), "matching_scores0": s0.double()}
