"""DUSt3R dense matcher plugin on the MI355X HIP backend.

Drop-in for imcui/hloc/matchers/duster.py: module name `duster`, same `default_conf` (:24-29), same `preprocess` contract
(:41-56).  What runs on the device is the NETWORK the reference calls through `dust3r.inference.inference(pairs, self.net,
device, batch_size=1)` (:73) on the two directed pairs (image0, image1), (image1, image0) (`make_pairs(..., symmetrize=True)`,
:70-72): `AsymmetricCroCo3DStereo` = ViT-L encoder + two-stream ViT-B decoder + DPT point-map heads, in libimcui_hip
(imcui_hip_dust3r_forward).  `inference_output()` returns the dictionary upstream's `inference` returns -- `view1`, `view2`,
`pred1 = {pts3d, conf}`, `pred2 = {pts3d_in_other_view, conf}`, batch entries in `make_pairs`' order: (image1, image0), then
(image0, image1) (the 'complete' scene graph lists (i, j < i) first, `symmetrize` appends the swapped pairs) -- so the host-side steps of
the wrapper consume it unchanged.  Every image is encoded once (upstream encodes both images again for the swapped pair).

Host-side steps (duster.py:74-108): `global_aligner(mode=PairViewer)` (focal estimation + `cv2.solvePnPRansac`), confidence
masks, `find_reciprocal_matches` (3-D nearest neighbours with a KD-tree) and the linspace sub-sampling to `max_keypoints`.
That is RANSAC geometry on the host in the reference and stays there (north_star): `_forward` runs those steps with
upstream's own `dust3r` package when it is importable (the `third_party/dust3r` submodule of the reference checkout) and
raises a clear ImportError otherwise -- neither the package nor cv2 exist offline, so they are not restated here.

Weights: conf["state_dict"] / conf["weights_path"] (or conf["packed"], the result of an earlier `backend.pack_dust3r`: packing
the 578 M parameters takes ~20 s) with upstream's parameter names (`duster_vit_large.pth` holds them under
the key "model", which `resolve_state_dict` unwraps).  The architecture (widths, depths) is read from the tensors.
"""
from __future__ import annotations

import numpy as np
import torch

from ... import backend
from ..utils.base_model import BaseModel
from ..utils.weights import resolve_state_dict


class Duster(BaseModel):
    default_conf = {
        "name": "Dust3r",
        "model_name": "duster_vit_large.pth",
        "max_keypoints": 3000,
        "vit_patch_size": 16,
        "arithmetic": "fp32",  # HIP backend only: "fp32" = 3 x f16 split products (fp32-grade results), "fp16" = one f16 product (bf16-class, faster)
    }
    required_inputs = ["image0", "image1"]
    weights_subdir = "duster"  # sub-directory of the checkpoint repository (base_model._download_model: stem of the module file)

    def _init(self, conf):
        if conf.get("packed") is not None:  # (packed buffer, architecture) of an earlier backend.pack_dust3r call
            packed, self.net_cfg = conf["packed"]
            self.conf.pop("packed", None)
            self.register_buffer("packed", packed, persistent=False)
            self._impl = backend.DUSt3RHIP()
            return
        sd = resolve_state_dict(conf, self.weights_subdir)
        if "patch_embed.proj.weight" not in sd or not any(k.startswith("downstream_head1.dpt.") for k in sd):
            raise KeyError("DUSt3R weights must be an AsymmetricCroCo3DStereo state dict with the DPT head (keys patch_embed.proj.*, downstream_head1.dpt.*)")
        self.conf.pop("state_dict", None)
        packed, self.net_cfg = backend.pack_dust3r(sd)
        self.register_buffer("packed", packed, persistent=False)
        self._impl = backend.DUSt3RHIP()

    def forward_pairs(self, images: torch.Tensor, pairs, dump: bool = False) -> dict:
        """The network on any set of directed pairs over `images` [NI,3,H,W] in [0,1]: {"pts3d": [2,P,H,W,3], "conf": [2,P,H,W]}."""
        arith = {"fp32": 0, "fp16": 1}[self.conf.get("arithmetic", "fp32")]  # read per call: conf is mutable at run time
        return self._impl.forward(self.packed, self.net_cfg, images, pairs, dump, arith)

    def inference_output(self, data: dict) -> dict:
        """What `inference(pairs, self.net, device, batch_size=1)` returns for the symmetrised pair (duster.py:66-73)."""
        return self._symmetrised(data)[1]

    def _symmetrised(self, data: dict):
        """-> (raw network outputs [2 views, 2 directed pairs, ...], upstream's inference dictionary); no state is kept on the
        plugin object: the UI calls one model from several worker threads."""
        img0, img1 = data["image0"], data["image1"]
        if img0.shape != img1.shape or img0.shape[0] != 1:
            raise ValueError("DUSt3R expects one pair of images of one size (the wrapper's preprocess guarantees it)")
        H, W = img0.shape[-2:]
        if H % 16 or W % 16:
            raise ValueError(f"DUSt3R needs image sizes that are multiples of the patch size 16 (the wrapper's preprocess rounds to it), got {W}x{H}")
        out = self.forward_pairs(torch.cat((img0, img1), 0), [[1, 0], [0, 1]])  # make_pairs' order: (image1, image0), (image0, image1)
        norm = [(img0 - 0.5) / 0.5, (img1 - 0.5) / 0.5]
        shape = torch.tensor([[H, W], [H, W]])

        def view(a, b):  # collated views of the two directed pairs
            return {"img": torch.cat((norm[a], norm[b]), 0), "true_shape": shape, "idx": [a, b], "instance": [str(a), str(b)]}

        return out, {
            "view1": view(1, 0),
            "view2": view(0, 1),
            "pred1": {"pts3d": out["pts3d"][0], "conf": out["conf"][0]},
            "pred2": {"pts3d_in_other_view": out["pts3d"][1], "conf": out["conf"][1]},
            "loss": None,
        }

    def _forward(self, data):
        output = self.inference_output(data)
        try:
            from dust3r.cloud_opt import GlobalAlignerMode, global_aligner
            from dust3r.utils.geometry import find_reciprocal_matches, xy_grid
        except ImportError as e:
            raise ImportError(
                "the DUSt3R network ran on the HIP backend (see inference_output()); the pose / reciprocal-matching steps of "
                "imcui/hloc/matchers/duster.py:74-108 use upstream's `dust3r` package (third_party/dust3r) and cv2, which are not installed"
            ) from e
        scene = global_aligner(output, device=data["image0"].device, mode=GlobalAlignerMode.PairViewer)
        masks = [m.cpu().numpy() for m in scene.get_masks()]
        clouds = [p.detach().cpu().numpy()[m] for p, m in zip(scene.get_pts3d(), masks)]
        empty = {"keypoints0": torch.zeros([0, 2]), "keypoints1": torch.zeros([0, 2])}
        if len(clouds[1]) == 0:
            return empty
        # pixel coordinates of the confident points of either image, in the order of `clouds`
        pixels = [xy_grid(im.shape[1], im.shape[0])[m] for im, m in zip(scene.imgs, masks)]
        in_p2, nn_in_p1, _ = find_reciprocal_matches(clouds[0], clouds[1])
        k1 = pixels[1][in_p2]
        k0 = pixels[0][nn_in_p1][in_p2]
        limit = self.conf["max_keypoints"]
        if limit is not None and len(k0) > limit:  # evenly spaced subset, as the reference takes it
            pick = np.round(np.linspace(0, len(k0) - 1, limit)).astype(int)
            k0, k1 = k0[pick], k1[pick]
        return {"keypoints0": torch.from_numpy(k0), "keypoints1": torch.from_numpy(k1)}
