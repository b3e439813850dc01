"""EfficientLoFTR dense matcher plugin on the MI355X HIP backend.

Drop-in for imcui/hloc/matchers/eloftr.py: module name `eloftr`, same `default_conf` (:25-34) and `required_inputs`
(:35); `_forward` keeps the wrapper's semantics -- image0 <-> image1 are swapped before the net ("we refine kpts in
image0", :69-78), the top-k matches by confidence are kept with `argsort(descending)[:k]` (:90-97), key names are swapped
back (:100) and `mconf` becomes `scores`.  The model itself (the 'full' EfficientLoFTR, fp32, after `reparameter()`,
:56-61) runs in libimcui_hip (imcui_hip_eloftr_forward).

Weights: conf["state_dict"] / conf["weights_path"] with the parameter names of the maintained port
(`transformers.EfficientLoFTRForKeypointMatching.state_dict()`, e.g. the `zju-community/efficientloftr` checkpoint).
The upstream `eloftr_outdoor.ckpt` the reference downloads stores the same tensors under the upstream module names (its
sources are an un-vendored submodule, absent from the reference tree, so the name mapping cannot be verified here): such
a checkpoint is refused with a clear message rather than mapped by guesswork -- convert it once with the port.

Image pairs whose two images differ in size are accepted (the batch path of `match_dense.py` produces them).

`model_type: "opt"` and `precision: "mp" / "fp16"` (eloftr.py:39-47) are speed variants of the same network for CUDA
hosts; the HIP path always computes the 'full' network with fp32-grade arithmetic and rejects other settings loudly.
"""
from __future__ import annotations

import torch

from ... import backend
from ..utils.base_model import BaseModel
from ..utils.weights import resolve_state_dict

def check_port_names(sd: dict) -> dict:
    """The HIP packer reads the port's parameter names.  An upstream-named checkpoint (`matcher.backbone...`,
    `loftr_coarse...`) is refused with a pointer to the conversion instead of being mapped by guesswork: the upstream
    module tree cannot be confirmed offline and a silent mis-assignment would produce plausible but wrong matches."""
    if not any(k.startswith("efficientloftr.") for k in sd) or not any(k.startswith("refinement_layer.") for k in sd):
        raise KeyError(
            "EfficientLoFTR weights must use the parameter names of transformers.EfficientLoFTRForKeypointMatching "
            f"(got keys like {next(iter(sd))!r}); load the upstream checkpoint into the port once and save its state_dict()"
        )
    return sd


class ELoFTR(BaseModel):
    default_conf = {
        "model_name": "eloftr_outdoor.ckpt",
        "match_threshold": 0.2,
        "max_keypoints": -1,
        "model_type": "full",
        "precision": "fp32",
    }
    required_inputs = ["image0", "image1"]

    def _init(self, conf):
        if conf.get("model_type", "full") != "full" or conf.get("precision", "fp32") != "fp32":
            raise NotImplementedError("the HIP EfficientLoFTR computes the 'full' model in fp32-grade arithmetic (model_type / precision are CUDA speed knobs)")
        sd = resolve_state_dict(conf, "eloftr")
        if "state_dict" in sd and isinstance(sd["state_dict"], dict):
            sd = sd["state_dict"]
        sd = check_port_names(sd)
        self.conf.pop("state_dict", None)
        self.register_buffer("packed", backend.pack_eloftr(sd), persistent=False)
        self._impl = backend.ELoFTRHIP()

    def forward_batched(self, image0: torch.Tensor, image1: torch.Tensor, debug_windows: bool = False) -> dict:
        """Upstream forward(image0, image1) on a batch: fixed-capacity outputs, no host sync."""
        return self._impl.forward(self.packed, image0, image1, self.conf["match_threshold"], debug_windows)

    def forward_pairs(self, image0: torch.Tensor, image1: torch.Tensor) -> list:
        """`_forward` on B pairs at once (the batched dense driver): the per-pair dictionaries the wrapper would return for
        `{"image0": image0[b:b+1], "image1": image1[b:b+1]}` -- images exchanged before the net, per-pair top-k by confidence,
        key names exchanged back.  One device-to-host read (the match count)."""
        out = self.forward_batched(image1, image0)
        n = int(out["num_matches"][0])
        bidx = out["batch_indexes"][:n]
        kp0, kp1, conf = out["keypoints0"][:n], out["keypoints1"][:n], out["confidence"][:n]
        top_k = self.conf["max_keypoints"]
        res = []
        for b in range(image0.shape[0]):
            sel = (bidx == b).nonzero()[:, 0]
            k0, k1, sc = kp0[sel], kp1[sel], conf[sel]
            if top_k is not None and len(sc) > top_k:
                keep = torch.argsort(sc, descending=True)[:top_k]
                k0, k1, sc = k0[keep], k1[keep], sc[keep]
            res.append({"keypoints0": k1, "keypoints1": k0, "scores": sc})
        return res

    def _forward(self, data):
        out = self.forward_batched(data["image1"], data["image0"])  # the reference refines key-points in image0
        n = int(out["num_matches"][0])
        kp0, kp1, scores = out["keypoints0"][:n], out["keypoints1"][:n], out["confidence"][:n]
        top_k = self.conf["max_keypoints"]
        if top_k is not None and len(scores) > top_k:
            keep = torch.argsort(scores, descending=True)[:top_k]
            kp0, kp1, scores = kp0[keep], kp1[keep], scores[keep]
        return {"keypoints0": kp1, "keypoints1": kp0, "scores": scores}
