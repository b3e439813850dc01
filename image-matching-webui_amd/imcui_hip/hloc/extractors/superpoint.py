"""SuperPoint extractor plugin on the MI355X HIP backend.

Drop-in for imcui/hloc/extractors/superpoint.py: same module name (`superpoint`), one BaseModel
subclass, same `default_conf` (:34-41), `required_inputs` (:42), `detection_noise` (:43), and the
runtime conf is re-read on every call like `self.net(data, self.conf)` (:57).  The arithmetic
(:56-57 -> upstream SuperPoint.forward) runs in libimcui_hip (imcui_hip_superpoint_forward).
"""
from __future__ import annotations

import warnings

import torch

from ... import backend
from ..utils.base_model import BaseModel
from ..utils.weights import resolve_state_dict


class SuperPoint(BaseModel):
    default_conf = {
        "nms_radius": 4,
        "model_name": "superpoint_v1.pth",
        "keypoint_threshold": 0.005,
        "max_keypoints": -1,
        "remove_borders": 4,
        "fix_sampling": False,
        # opt-in (not a reference key): replay the extractor from a captured HIP graph per (image shape, conf) -- the reference calls
        # `_forward` with ONE image at a time (extract_features.py:166-170), ~70 launches of a few microseconds; outputs are unchanged
        "hip_graph": False,
    }
    required_inputs = ["image"]
    detection_noise = 2.0

    def _init(self, conf):
        sd = resolve_state_dict(conf, "superglue")
        conf.pop("state_dict", None)  # keep self.conf small / printable
        self.conf.pop("state_dict", None)
        # registered buffer: counted by the UI model cache and moved by `.to(device)`
        self.register_buffer("packed", backend.pack_superpoint(sd), persistent=False)
        self._impl = backend.SuperPointHIP()

    def forward_batched(self, image: torch.Tensor, want_score_map: bool = False) -> dict:
        """Fixed-stride outputs, no host synchronisation (graph-capturable for every conf): keypoints [B,K,2],
        scores [B,K], descriptors [B,K,256] (row per key-point), num_keypoints [B] int32, status [1] int32."""
        return self._impl.forward(self.packed, self._gray(image), self.conf, want_score_map)

    @staticmethod
    def _gray(image: torch.Tensor) -> torch.Tensor:
        if image.shape[1] == 3:  # RGB -> gray (upstream weights), the hloc path always feeds gray
            scale = image.new_tensor([0.299, 0.587, 0.114]).view(1, 3, 1, 1)
            image = (image * scale).sum(1, keepdim=True)
        return image

    def forward_checked(self, image: torch.Tensor):
        """`forward_batched` + the ONE device->host copy that brings the per-image counts and the selection status word; a
        capacity overflow (status bit 1: `max_keypoints = -1` sizes the outputs by the NMS bound, which only exactly tied
        scores of flat images can exceed) is retried with room for every pixel, any other non-zero status raises.
        -> (outputs, counts).  Used by `_forward` and by the batched extraction driver."""
        out = self._forward_graphed(image) if self.conf.get("hip_graph", False) else self.forward_batched(image)
        *counts, status = torch.cat([out["num_keypoints"], out["status"]]).tolist()
        if status & 2:
            out = self._impl.forward(self.packed, self._gray(image), self.conf, kcap=image.shape[-2] * image.shape[-1])
            *counts, status = torch.cat([out["num_keypoints"], out["status"]]).tolist()
        if status:
            raise backend.ImcuiHipError(f"SuperPoint key-point selection failed (status {status})")
        return out, counts

    def _forward_graphed(self, image: torch.Tensor) -> dict:
        """`forward_batched` replayed from a HIP graph captured for this (shape, conf); outputs are CLONED out of the graph's static buffers (the
        next call -- the pair's second image -- overwrites them).  Falls back to eager launches if the capture fails."""
        from ...pipeline import GraphedCall

        c = self.conf
        key = (tuple(image.shape), str(image.device), c["nms_radius"], c["max_keypoints"], c["keypoint_threshold"], c["remove_borders"], c.get("fix_sampling", False))
        cache = self.__dict__.setdefault("_graphs", {})
        if key not in cache:
            try:
                cache[key] = GraphedCall(lambda img: self.forward_batched(img), image)
            except Exception as e:  # noqa: BLE001 -- capture is an optimisation: keep working without it, but say so (once per key)
                warnings.warn(f"SuperPoint hip_graph: capture failed for {key[:2]}, running eager launches instead ({type(e).__name__}: {e})", RuntimeWarning, stacklevel=2)
                cache[key] = None
        g = cache[key]
        if g is None:
            return self.forward_batched(image)
        return {k: v.clone() for k, v in g(image).items()}

    def _forward(self, data):
        # ragged lists are the reference contract
        out, counts = self.forward_checked(data["image"])
        kpts = [out["keypoints"][b, :n] for b, n in enumerate(counts)]
        scores = [out["scores"][b, :n] for b, n in enumerate(counts)]
        # reference layout is [256, N]: a transposed view of the row-per-keypoint buffer
        desc = [out["descriptors"][b, :n].t() for b, n in enumerate(counts)]
        return {"keypoints": kpts, "scores": scores, "descriptors": desc}
