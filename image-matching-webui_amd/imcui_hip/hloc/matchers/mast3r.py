"""MASt3R dense matcher plugin on the MI355X HIP backend.

Drop-in for imcui/hloc/matchers/mast3r.py: module name `mast3r`, same `default_conf` (:24-29).  The NETWORK the reference runs
through `inference(pairs, self.net, DEVICE, batch_size=1)` (:56) -- `AsymmetricMASt3R` = the DUSt3R encoder / decoder with the
'catmlp+dpt' head: DPT point maps plus 24-d unit-norm local descriptors and their confidence -- runs in libimcui_hip
(imcui_hip_dust3r_forward with desc_dim = 24).  `inference_output()` returns upstream's `{view1, view2, pred1, pred2}` with
`pts3d` / `pts3d_in_other_view`, `conf`, `desc`, `desc_conf`; every image is encoded once.

The matching step -- `fast_reciprocal_NNs(desc1, desc2, subsample_or_initxy1=2, dist="dot", block_size=2**13)` (:68-75) of the
un-vendored `third_party/mast3r` (mast3r/fast_nn.py) -- is restated in `fast_reciprocal_nns` below: start from every second
pixel of image 1, hop to the nearest neighbour (largest dot product) in image 2 and back, up to 10 rounds, keep the chains that
closed on themselves (reciprocal pairs), unique pairs ordered by (position in image 1, position in image 2).  The nearest
neighbour searches (up to 65 536 queries against the 262 144 descriptors of a 512x512 map) run in libimcui_hip
(`imcui_hip_nn_argmax_f32`: similarity on the matrix cores fused with the running arg-max); the bookkeeping between two searches is
index arithmetic on device tensors, one host read of "any chain still open" per round (upstream reads the same through numpy).
Then the linspace sub-sampling to `max_keypoints` (:88-92).
"""
from __future__ import annotations

import numpy as np
import torch

from ... import backend
from .duster import Duster


class Mast3r(Duster):
    default_conf = {
        "name": "Mast3r",
        "model_name": "MASt3R_ViTLarge_BaseDecoder_512_catmlpdpt_metric.pth",
        "max_keypoints": 2000,
        "vit_patch_size": 16,
        "arithmetic": "fp32",
        # HIP backend only: arithmetic of the nearest-neighbour searches -- "auto" / "split" = 3 x f16 split products (the arithmetic the
        # network itself runs in; 4 x fewer matrix cycles, fp32-grade similarities), "fp32" = the exact-f32 matrix instruction
        # (candidates closer than ~3e-7 may resolve differently between the two).  The NETWORK has no exact-f32 mode: with
        # imcui_hip_set_precision(h, 0) the plugin raises a clear error at the first call.
        "matcher_arithmetic": "auto",
    }
    weights_subdir = "mast3r"

    def _init(self, conf):
        super()._init(conf)
        if self.net_cfg.get("desc_dim", 0) <= 0:
            raise KeyError("MASt3R weights must hold downstream_head{1,2}.head_local_features.* (the 'catmlp+dpt' head)")

    def inference_output(self, data: dict) -> dict:
        raw, out = self._symmetrised(data)
        for v, pred in enumerate(("pred1", "pred2")):
            out[pred]["desc"] = raw["desc"][v]
            out[pred]["desc_conf"] = raw["desc_conf"][v]
        return out

    def _forward(self, data):
        output = self.inference_output(data)
        # the reference matches the descriptors of the SECOND batch entry = the pair (image0 as view 1, image1 as view 2): mast3r.py:61-64
        # -> keypoints0 are pixels of image0, keypoints1 of image1
        desc1, desc2 = output["pred1"]["desc"][1], output["pred2"]["desc"][1]
        mode = self.conf.get("matcher_arithmetic", "auto")
        split = backend.get_precision(desc1.device) == 1 if mode == "auto" else mode == "split"
        k0, k1 = fast_reciprocal_nns(desc1, desc2, subsample=2, split=split)
        if len(k0) == 0:
            return {"keypoints0": torch.zeros([0, 2]), "keypoints1": torch.zeros([0, 2])}
        k0, k1 = k0.cpu().numpy(), k1.cpu().numpy()
        limit = self.conf["max_keypoints"]
        if limit is not None and len(k0) > limit:
            pick = np.round(np.linspace(0, len(k0) - 1, limit)).astype(int)
            k0, k1 = k0[pick], k1[pick]
        return {"keypoints0": torch.from_numpy(np.ascontiguousarray(k0)), "keypoints1": torch.from_numpy(np.ascontiguousarray(k1))}


def fast_reciprocal_nns(desc1: torch.Tensor, desc2: torch.Tensor, subsample: int = 2, max_iter: int = 10, nn=None, split: bool = False):
    """desc1 [H1,W1,D], desc2 [H2,W2,D] (device) -> (xy1 [K,2], xy2 [K,2]) int64 pixel (x, y) of the reciprocal matches, ordered by
    (linear position in image 1, linear position in image 2).  `nn(queries, db)` = first arg-max of the dot products (default: the
    HIP kernel); the loop is upstream's `fast_reciprocal_NNs(pts1, pts2, subsample_or_initxy1=S, ret_xy=True, pixel_tol=0)`."""
    nn = nn or (lambda q, db: backend.nn_argmax(q, db, split=split))

    def nn_of(points, which, db):
        """nearest neighbour of every listed point: chains that stand on the SAME point are searched once (the answer is a
        function of the point; upstream searches every chain, with the same result)."""
        uq, inv = torch.unique(which, return_inverse=True)
        return nn(points[uq], db)[inv]

    H1, W1, D = desc1.shape
    H2, W2, _ = desc2.shape
    dev = desc1.device
    p1, p2 = desc1.reshape(-1, D), desc2.reshape(-1, D)
    S = subsample
    ys, xs = torch.meshgrid(torch.arange(S // 2, H1, S, device=dev), torch.arange(S // 2, W1, S, device=dev), indexing="ij")
    xy1 = torch.unique((xs + W1 * ys).reshape(-1))  # sorted start positions (np.unique in upstream)
    xy2 = torch.full_like(xy1, -1)
    old_xy1, old_xy2 = xy1.clone(), xy2.clone()
    notyet = torch.ones_like(xy1, dtype=torch.bool)
    niter = 0
    while bool(notyet.any()):
        act = notyet.nonzero()[:, 0]
        xy2[act] = nn_of(p1, xy1[act], p2)
        notyet &= old_xy2 != xy2  # chains whose image-2 end did not move have converged
        act = notyet.nonzero()[:, 0]
        if len(act):
            xy1[act] = nn_of(p2, xy2[act], p1)
        notyet &= old_xy1 != xy1
        niter += 1
        if niter >= max_iter:
            break
        old_xy2.copy_(xy2)
        old_xy1.copy_(xy1)
    done = ~notyet
    key = torch.unique((xy1[done] << 32) | xy2[done])  # unique pairs, ordered by position in image 1, then in image 2
    a, b = key >> 32, key & 0xFFFFFFFF
    return torch.stack((a % W1, a // W1), 1), torch.stack((b % W2, b // W2), 1)
