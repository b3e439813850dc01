"""MASt3R dense matcher plugin on the MI355X HIP backend.

Drop-in for imcui/hloc/matchers/mast3r.py: module name `mast3r`, same `default_conf` (:24-29).  The NETWORK the reference runs
through `inference(pairs, self.net, DEVICE, batch_size=1)` (:56) -- `AsymmetricMASt3R` = the DUSt3R encoder / decoder with the
'catmlp+dpt' head: DPT point maps plus 24-d unit-norm local descriptors and their confidence -- runs in libimcui_hip
(imcui_hip_dust3r_forward with desc_dim = 24).  `inference_output()` returns upstream's `{view1, view2, pred1, pred2}` with
`pts3d` / `pts3d_in_other_view`, `conf`, `desc`, `desc_conf`; every image is encoded once.

The matching step (`fast_reciprocal_NNs(desc1, desc2, subsample_or_initxy1=2, dist="dot", block_size=2**13)`, :68-75: an
iterated nearest-neighbour search between the two 512x512x24 descriptor maps) and the linspace sub-sampling (:88-92) run with
upstream's `mast3r` package when it is importable; without it `_forward` raises ImportError after the network has run (a
device-side reciprocal search is the next step of this path).
"""
from __future__ import annotations

import numpy as np
import torch

from .duster import Duster


class Mast3r(Duster):
    default_conf = {
        "name": "Mast3r",
        "model_name": "MASt3R_ViTLarge_BaseDecoder_512_catmlpdpt_metric.pth",
        "max_keypoints": 2000,
        "vit_patch_size": 16,
        "arithmetic": "fp32",
    }
    weights_subdir = "mast3r"

    def _init(self, conf):
        super()._init(conf)
        if self.net_cfg.get("desc_dim", 0) <= 0:
            raise KeyError("MASt3R weights must hold downstream_head{1,2}.head_local_features.* (the 'catmlp+dpt' head)")

    def inference_output(self, data: dict) -> dict:
        out = super().inference_output(data)
        raw = self._last_forward
        for v, pred in enumerate(("pred1", "pred2")):
            out[pred]["desc"] = raw["desc"][v]
            out[pred]["desc_conf"] = raw["desc_conf"][v]
        return out

    def _forward(self, data):
        output = self.inference_output(data)
        try:
            from mast3r.fast_nn import fast_reciprocal_NNs
        except ImportError as e:
            raise ImportError(
                "the MASt3R network ran on the HIP backend (see inference_output()); the reciprocal matching of "
                "imcui/hloc/matchers/mast3r.py:68-75 uses upstream's `mast3r` package (third_party/mast3r), which is not installed"
            ) from e
        # the reference matches the descriptors of the SECOND directed pair (image1 as view 1, image0 as view 2), mast3r.py:61-64
        desc1, desc2 = output["pred1"]["desc"][1], output["pred2"]["desc"][1]
        k0, k1 = fast_reciprocal_NNs(desc1, desc2, subsample_or_initxy1=2, device=desc1.device, dist="dot", block_size=2**13)
        if len(k0) == 0:
            return {"keypoints0": torch.zeros([0, 2]), "keypoints1": torch.zeros([0, 2])}
        limit = self.conf["max_keypoints"]
        if limit is not None and len(k0) > limit:
            pick = np.round(np.linspace(0, len(k0) - 1, limit)).astype(int)
            k0, k1 = k0[pick], k1[pick]
        return {"keypoints0": torch.from_numpy(np.ascontiguousarray(k0)), "keypoints1": torch.from_numpy(np.ascontiguousarray(k1))}
