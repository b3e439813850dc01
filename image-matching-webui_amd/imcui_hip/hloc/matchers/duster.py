"""DUSt3R dense matcher plugin on the MI355X HIP backend.

Drop-in for imcui/hloc/matchers/duster.py: module name `duster`, same `default_conf` (:24-29), same `preprocess` contract
(:41-56).  What runs on the device is the NETWORK the reference calls through `dust3r.inference.inference(pairs, self.net,
device, batch_size=1)` (:73) on the two directed pairs (image0, image1), (image1, image0) (`make_pairs(..., symmetrize=True)`,
:70-72): `AsymmetricCroCo3DStereo` = ViT-L encoder + two-stream ViT-B decoder + DPT point-map heads, in libimcui_hip
(imcui_hip_dust3r_forward).  `inference_output()` returns the dictionary upstream's `inference` returns -- `view1`, `view2`,
`pred1 = {pts3d, conf}`, `pred2 = {pts3d_in_other_view, conf}`, batch entries in `make_pairs`' order: (image1, image0), then
(image0, image1) (the 'complete' scene graph lists (i, j < i) first, `symmetrize` appends the swapped pairs) -- so the host-side steps of
the wrapper consume it unchanged.  Every image is encoded once (upstream encodes both images again for the swapped pair).

Host-side steps (duster.py:74-108).  `global_aligner(mode=PairViewer)` (focal estimation + `cv2.solvePnPRansac`) is RANSAC
geometry on the host in the reference and stays there (north_star): `_forward` obtains the scene through `self.aligner(output,
device)` -- upstream's own `dust3r.cloud_opt.global_aligner` when the `third_party/dust3r` submodule of the reference checkout is
importable, otherwise (round 4) the host-side restatement of PairViewer for one symmetrised pair, `pair_viewer.PairViewerScene`
(Weiszfeld focal per image, relative pose from a seeded numpy PnP-RANSAC with upstream's 100 iterations / 5 px rule, masks =
max-over-edges confidence > 3, depth maps re-projected through the recovered pinholes; parity unpinned -- neither upstream's
package nor cv2 exist offline -- tested against ground-truth geometry in tests/test_pair_viewer_cpu.py), so `_forward` needs nothing of
upstream's.  Everything AFTER the aligner is restated here as well: confidence masks -> pixel grids (`xy_grid`) -> reciprocal 3-D nearest
neighbours (`find_reciprocal_matches`: two KD-trees, image-1 points whose nearest image-0 point names them back) -> the
`np.linspace` sub-sampling to `max_keypoints` (`matches_from_scene`; tested on CPU against the brute-force restatement
`oracle/dust3r.py: duster_matches_from_scene` with the aligner mocked).

Image sizes: multiples of 16, and the two images of a pair may differ.  The reference's `Duster.preprocess` is never called by
its callers; `match_dense.match_images` / `ImagePairDataset.preprocess` resize each image on its own (resize_max 512, dfactor 16),
so two photos of different aspect ratio reach upstream's net at two sizes, which `dust3r.inference` handles by encoding the
views separately and collating the per-pair results as lists.  Same here: one size -> imcui_hip_dust3r_forward and stacked
tensors; two sizes -> imcui_hip_dust3r_forward_sizes (every sequence carries its own token grid; the heads run per size) and
lists of per-pair maps, the structure upstream's `inference` returns in that case (restated from the published code of the
un-vendored `third_party/dust3r`; `pred1["pts3d"][1]`-style indexing reads the same entry either way).

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
            backend.check_dust3r_packed(packed, self.net_cfg)  # size + format trailer: refuses a blob of another library version
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

    def forward_pairs_sizes(self, images, pairs, dump: bool = False) -> dict:
        """The same on a LIST of images [3,H_i,W_i] of up to four different sizes: nested lists [view][pair] of maps [H,W,3] / [H,W],
        the map of (view v, pair p) at the size of image pairs[p][v]."""
        arith = {"fp32": 0, "fp16": 1}[self.conf.get("arithmetic", "fp32")]
        return self._impl.forward_sizes(self.packed, self.net_cfg, images, pairs, dump, arith)

    def inference_output(self, data: dict) -> dict:
        """What `inference(pairs, self.net, device, batch_size=1)` returns for the symmetrised pair (duster.py:66-73)."""
        return self._symmetrised(data)[1]

    def _symmetrised(self, data: dict):
        """-> (raw network outputs [2 views, 2 directed pairs, ...], upstream's inference dictionary); no state is kept on the
        plugin object: the UI calls one model from several worker threads."""
        img0, img1 = data["image0"], data["image1"]
        if img0.shape[0] != 1 or img1.shape[0] != 1:
            raise ValueError("DUSt3R matches one image pair per call (batch 1, as the reference wrapper)")
        for im in (img0, img1):
            if im.shape[-2] % 16 or im.shape[-1] % 16:
                raise ValueError(f"DUSt3R needs image sizes that are multiples of the patch size 16 (the wrapper's preprocess rounds to it), got {im.shape[-1]}x{im.shape[-2]}")
        norm = [(img0 - 0.5) / 0.5, (img1 - 0.5) / 0.5]
        order = [[1, 0], [0, 1]]  # make_pairs' order: (image1, image0), (image0, image1)
        if img0.shape == img1.shape:
            H, W = img0.shape[-2:]
            out = self.forward_pairs(torch.cat((img0, img1), 0), order)
            shape = torch.tensor([[H, W], [H, W]])

            def view(a, b):  # collated views of the two directed pairs
                return {"img": torch.cat((norm[a], norm[b]), 0), "true_shape": shape, "idx": [a, b], "instance": [str(a), str(b)]}

        else:
            # two sizes (each image was resized on its own): upstream's `inference` runs such pairs one per batch and collates
            # with `lists=True` -- every value is the LIST of the per-pair entries, the batch axis dropped
            out = self.forward_pairs_sizes([img0, img1], order)

            def view(a, b):
                return {"img": [norm[a][0], norm[b][0]], "true_shape": [torch.tensor(norm[a].shape[-2:]), torch.tensor(norm[b].shape[-2:])],
                        "idx": [a, b], "instance": [str(a), str(b)]}  # fmt: skip

        return out, {
            "view1": view(1, 0),
            "view2": view(0, 1),
            "pred1": {"pts3d": out["pts3d"][0], "conf": out["conf"][0]},
            "pred2": {"pts3d_in_other_view": out["pts3d"][1], "conf": out["conf"][1]},
            "loss": None,
        }

    @staticmethod
    def aligner(output: dict, device):
        """duster.py:74: `global_aligner(output, device=device, mode=GlobalAlignerMode.PairViewer)`.  Upstream's package (host geometry:
        focal estimation, cv2.solvePnPRansac) when it is importable -- the reference's own code then runs unchanged; otherwise the
        host-side restatement of PairViewer for one symmetrised pair (`pair_viewer.PairViewerScene`: same focal estimator, same
        confidence rule, a seeded numpy PnP-RANSAC in place of cv2's; parity unpinned, see its docstring), so that `_forward` stands
        alone on a machine without third_party/dust3r.  Replaceable (tests mock it; a deployment may bind its own)."""
        try:
            from dust3r.cloud_opt import GlobalAlignerMode, global_aligner
        except ImportError:
            from .pair_viewer import PairViewerScene

            return PairViewerScene(output)
        return global_aligner(output, device=device, mode=GlobalAlignerMode.PairViewer)

    def matches_from_scene(self, imgs, masks, pts3d) -> dict:
        """duster.py:76-108 after the aligner: `imgs` [2] arrays [H, W, 3], `masks` [2] boolean [H, W] (scene.get_masks()),
        `pts3d` [2] arrays [H, W, 3] (scene.get_pts3d()) -> {"keypoints0", "keypoints1"} in pixels (x, y) of image0 / image1."""
        masks = [np.asarray(m.cpu() if torch.is_tensor(m) else m, dtype=bool) for m in masks]
        pts3d = [np.asarray(p.detach().cpu() if torch.is_tensor(p) else p) for p in pts3d]
        # (points that are not finite cannot enter a KD-tree -- scipy raises; a confident pixel with such a point is dropped)
        masks = [m & np.isfinite(p).all(-1) for m, p in zip(masks, pts3d)]
        clouds = [p[m] for p, m in zip(pts3d, masks)]
        # pixel coordinates of the confident points of either image, in the order of `clouds` (xy_grid(W, H)[mask])
        pixels = [xy_grid(im.shape[1], im.shape[0])[m] for im, m in zip(imgs, masks)]
        if len(clouds[1]) == 0:
            return {"keypoints0": torch.zeros([0, 2]), "keypoints1": torch.zeros([0, 2])}
        in_p2, nn_in_p1, _ = find_reciprocal_matches(clouds[0], clouds[1])
        k1 = pixels[1][in_p2]
        k0 = pixels[0][nn_in_p1][in_p2]
        limit = self.conf["max_keypoints"]
        if limit is not None and len(k0) > limit:  # evenly spaced subset, as the reference takes it
            pick = np.round(np.linspace(0, len(k0) - 1, limit)).astype(int)
            k0, k1 = k0[pick], k1[pick]
        return {"keypoints0": torch.from_numpy(k0), "keypoints1": torch.from_numpy(k1)}

    def _forward(self, data):
        output = self.inference_output(data)
        scene = self.aligner(output, data["image0"].device)
        return self.matches_from_scene(scene.imgs, scene.get_masks(), scene.get_pts3d())


def xy_grid(W: int, H: int) -> np.ndarray:
    """upstream dust3r.utils.geometry.xy_grid(W, H): [H, W, 2] int32 with out[j, i] = (i, j)."""
    xs, ys = np.meshgrid(np.arange(W, dtype=np.int32), np.arange(H, dtype=np.int32), indexing="xy")
    return np.stack((xs, ys), -1)


def find_reciprocal_matches(P1: np.ndarray, P2: np.ndarray):
    """upstream dust3r.utils.geometry.find_reciprocal_matches: nearest neighbours both ways with two KD-trees (Euclidean, the
    first of equal candidates as scipy's tree returns it); -> (reciprocal_in_P2 [len(P2)] bool, nn2_in_P1 [len(P2)] int, count)."""
    from scipy.spatial import cKDTree

    if len(P1) == 0:
        return np.zeros(len(P2), dtype=bool), np.zeros(len(P2), dtype=np.int64), 0
    _, nn1_in_p2 = cKDTree(P2).query(P1, workers=8)
    _, nn2_in_p1 = cKDTree(P1).query(P2, workers=8)
    reciprocal_in_p2 = nn1_in_p2[nn2_in_p1] == np.arange(len(nn2_in_p1))
    return reciprocal_in_p2, nn2_in_p1, int(reciprocal_in_p2.sum())
