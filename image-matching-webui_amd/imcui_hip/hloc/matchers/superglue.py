"""SuperGlue matcher plugin on the MI355X HIP backend.

Drop-in for imcui/hloc/matchers/superglue.py: module name `superglue`, same `default_conf` (:14-19) and
`required_inputs` (:20-29); the checkpoint is `superglue/<model_name>` of the model repo (:32-35).  The
arithmetic (:42-43 `self.net(data)` -> upstream SuperGlue.forward) runs in libimcui_hip
(imcui_hip_superglue_forward) for all B pairs of the call at once.
"""
from __future__ import annotations

import torch

from ... import backend
from ..utils.base_model import BaseModel
from ..utils.weights import resolve_state_dict


class SuperGlue(BaseModel):
    default_conf = {
        "weights": "outdoor",
        "model_name": "superglue_outdoor.pth",
        "sinkhorn_iterations": 100,
        "match_threshold": 0.2,
    }
    # The reference hands its conf to upstream once (`self.net = SG(conf)`, imcui/hloc/matchers/superglue.py:37; upstream's
    # `self.config = {**self.default_config, **config}` is a copy), so sinkhorn_iterations and match_threshold are those of `_init`: the
    # UI's later `matcher.conf["match_threshold"] = ...` on a cached model (imcui/ui/utils.py:921-922) changes nothing.  False (default)
    # = exactly that; True = re-read both on every call (what the slider intends).  Not a reference key: read with conf.get("runtime_match_threshold", False), default_conf stays the reference's.
    required_inputs = [
        "image0",
        "keypoints0",
        "scores0",
        "descriptors0",
        "image1",
        "keypoints1",
        "scores1",
        "descriptors1",
    ]

    def _init(self, conf):
        self._frozen = (int(conf["sinkhorn_iterations"]), float(conf["match_threshold"]))  # upstream's config copy (see default_conf)
        sd = resolve_state_dict(conf, "superglue")
        conf.pop("state_dict", None)
        self.conf.pop("state_dict", None)
        self.register_buffer("packed", backend.pack_superglue(sd), persistent=False)
        self._impl = backend.SuperGlueHIP()

    def forward_batched(self, kpts0, kpts1, scores0, scores1, desc0, desc1, n0, n1, size0, size1) -> dict:
        """Row-per-point descriptors [B,N,256]; n0/n1 [B] int32 valid counts; sizes (W, H).
        Fixed-stride int32 outputs, no host synchronisation."""
        c = self.conf
        iters, thr = (int(c["sinkhorn_iterations"]), float(c["match_threshold"])) if c.get("runtime_match_threshold", False) else self._frozen
        return self._impl.forward(self.packed, kpts0, kpts1, scores0, scores1, desc0, desc1, n0, n1, size0, size1, iters, thr)

    def _forward(self, data):
        kpts0, kpts1 = data["keypoints0"], data["keypoints1"]
        if kpts0.shape[1] == 0 or kpts1.shape[1] == 0:  # no keypoints (upstream early return, int32 matches)
            shape0, shape1 = kpts0.shape[:-1], kpts1.shape[:-1]
            return {
                "matches0": kpts0.new_full(shape0, -1, dtype=torch.int),
                "matches1": kpts1.new_full(shape1, -1, dtype=torch.int),
                "matching_scores0": kpts0.new_zeros(shape0),
                "matching_scores1": kpts1.new_zeros(shape1),
            }
        desc0 = data["descriptors0"].permute(0, 2, 1)
        desc1 = data["descriptors1"].permute(0, 2, 1)
        B, m = kpts0.shape[0], kpts0.shape[1]
        n = kpts1.shape[1]
        dev = kpts0.device
        size0 = tuple(data["image0"].shape[-2:][::-1])
        size1 = tuple(data["image1"].shape[-2:][::-1])
        n0 = torch.full((B,), m, dtype=torch.int32, device=dev)
        n1 = torch.full((B,), n, dtype=torch.int32, device=dev)
        out = self.forward_batched(kpts0, kpts1, data["scores0"], data["scores1"], desc0, desc1, n0, n1, size0, size1)
        return {
            "matches0": out["matches0"].long(),  # use -1 for invalid match
            "matches1": out["matches1"].long(),
            "matching_scores0": out["matching_scores0"],
            "matching_scores1": out["matching_scores1"],
        }
