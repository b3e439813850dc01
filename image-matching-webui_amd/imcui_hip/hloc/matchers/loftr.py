"""LoFTR dense matcher plugin on the MI355X HIP backend.

Drop-in for imcui/hloc/matchers/loftr.py: module name `loftr`, same `default_conf` (:13-18) and
`required_inputs` (:19); `_forward` keeps the wrapper's semantics -- image0 <-> image1 are swapped
before the net ("we refine kpts in image0", :42-51), the top-k matches by confidence are kept with
`argsort(descending)[:k]` (:58-65), key names are swapped back and `confidence` is renamed `scores`
(:67-70).  The model itself (kornia.feature.LoFTR.forward, :54) runs in libimcui_hip
(imcui_hip_loftr_forward).  Weights: conf["state_dict"] / conf["weights_path"] (kornia's `loftr_outdoor.ckpt`, or any LoFTR state dict with its names);
without either, the zoo's default entry (`weights: "outdoor"`, configs/matchers.py:249-256) resolves like the reference's
`LoFTR_(pretrained="outdoor")`: kornia when importable, else kornia's URL through the torch-hub cache (`kornia_pretrained`); a
`minima` model name (:27-36, sets temp_bug_fix) comes from `loftr/<model_name>` of the model repository like the reference's.

image0 and image1 may have different sizes (`minima_loftr`, configs/matchers.py:283, keeps each image's aspect ratio).
"""
from __future__ import annotations

import torch

from ... import backend
from ..utils.base_model import BaseModel
from ..utils.weights import resolve_state_dict


# kornia.feature.loftr.loftr `urls` (the files `LoFTR(pretrained=...)` fetches with torch.hub.load_state_dict_from_url); kornia is not
# a dependency of this package, so its download is restated: same URL, same torch-hub cache file, same `["state_dict"]` container
KORNIA_LOFTR_URLS = {
    "outdoor": "http://cmp.felk.cvut.cz/~mishkdmy/models/loftr_outdoor.ckpt",
    "indoor_new": "http://cmp.felk.cvut.cz/~mishkdmy/models/loftr_indoor_ds_new.ckpt",
    "indoor": "http://cmp.felk.cvut.cz/~mishkdmy/models/loftr_indoor.ckpt",
}


def kornia_pretrained(weights: str) -> dict:
    """What `kornia.feature.LoFTR(pretrained=weights)` loads (imcui/hloc/matchers/loftr.py:37): kornia itself when it is installed,
    otherwise its URL through torch.hub (a file already in `$TORCH_HOME/hub/checkpoints/` -- where kornia's own download leaves it --
    is used without touching the network)."""
    try:
        from kornia.feature import LoFTR as KorniaLoFTR

        return KorniaLoFTR(pretrained=weights).state_dict()
    except ImportError:
        pass
    return torch.hub.load_state_dict_from_url(KORNIA_LOFTR_URLS[weights], map_location="cpu")


class LoFTR(BaseModel):
    default_conf = {
        "weights": "outdoor",
        "match_threshold": 0.2,
        "sinkhorn_iterations": 20,
        "max_keypoints": -1,
    }
    # The reference writes match_threshold into kornia's config when the model is BUILT (`cfg["match_coarse"]["thr"] = conf["match_threshold"]`,
    # imcui/hloc/matchers/loftr.py:21-24; `CoarseMatching.__init__` keeps `self.thr`), so the UI's later `matcher.conf["match_threshold"] = ...`
    # on a cached model (imcui/ui/utils.py:921-922) never reaches the coarse matching.  False (default) = exactly that; True = re-read
    # conf["match_threshold"] on every call (what the slider intends).  Not a reference key: read with conf.get("runtime_match_threshold", False), default_conf stays the reference's.
    required_inputs = ["image0", "image1"]

    def _init(self, conf):
        self._match_threshold = float(conf["match_threshold"])  # frozen here, like `cfg["match_coarse"]["thr"]` (see default_conf)
        model_name = conf.get("model_name", None)
        minima = model_name is not None and "minima" in model_name
        # temp_bug_fix: the MINIMA checkpoint (loftr.py:27-28) and kornia's own rule for its re-trained indoor weights
        # (`LoFTR.__init__`: `if pretrained == "indoor_new": config["coarse"]["temp_bug_fix"] = True`)
        self.temp_bug_fix = minima or conf["weights"] == "indoor_new"
        fallback = None
        if conf.get("state_dict") is None and not conf.get("weights_path"):
            if minima:  # loftr.py:29-33: `loftr/<model_name>` of the model repository
                conf = {**conf, "model_name": model_name}
            else:  # loftr.py:37: `LoFTR_(pretrained=conf["weights"])` -- kornia downloads its own file; the model repository has none
                if conf["weights"] not in KORNIA_LOFTR_URLS:
                    raise ValueError(f"weights {conf['weights']!r}: kornia's LoFTR knows {sorted(KORNIA_LOFTR_URLS)}")
                weights = conf["weights"]

                def fallback():
                    try:
                        return kornia_pretrained(weights)
                    except Exception as e:  # noqa: BLE001
                        raise RuntimeError(f"kornia_pretrained({weights!r}) = {KORNIA_LOFTR_URLS[weights]}: {type(e).__name__}: {e}") from e

                conf = {**conf, "model_name": f"loftr_{conf['weights']}.ckpt"}  # last resort only (not a file the reference uses)
        sd = resolve_state_dict(conf, "loftr", fallback=fallback)
        self.conf.pop("state_dict", None)
        self.register_buffer("packed", backend.pack_loftr(sd), persistent=False)
        self._impl = backend.LoFTRHIP()

    def forward_batched(self, image0: torch.Tensor, image1: torch.Tensor) -> dict:
        """kornia LoFTR.forward(image0, image1) on a batch: fixed-capacity outputs, no host sync."""
        return self._impl.forward(self.packed, image0, image1, self.match_threshold(), self.temp_bug_fix)

    def match_threshold(self) -> float:
        """The coarse-matching threshold of this call: the value the model was built with, or conf's current one under the opt-in."""
        c = self.conf
        return float(c["match_threshold"]) if c.get("runtime_match_threshold", False) else self._match_threshold

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

    @staticmethod
    def _refuse_masks(data):
        """The reference renames `mask0` / `mask1` and hands them to the net (loftr.py:43-51); upstream uses them on padded training
        batches (coarse attention and `sim.masked_fill_(~mask, -inf)`).  No caller of the reference produces them for a
        single pair (`match_dense.ImagePairDataset`, `match_images`): the HIP path has no masked kernels and refuses the keys
        instead of silently ignoring them."""
        for k in ("mask0", "mask1"):
            if data.get(k) is not None:
                raise NotImplementedError(f"{k}: padded-batch masks are not supported by the HIP dense matchers (pass un-padded images)")

    def _forward(self, data):
        self._refuse_masks(data)
        # For consistency with hloc pairs the reference refines key-points in image0: swap the images
        out = self.forward_batched(data["image1"], data["image0"])
        n = int(out["num_matches"][0])  # ragged outputs are the reference contract (one D2H)
        kp0, kp1, scores = out["keypoints0"][:n], out["keypoints1"][:n], out["confidence"][:n]
        top_k = self.conf["max_keypoints"]
        if top_k is not None and len(scores) > top_k:
            keep = torch.argsort(scores, descending=True)[:top_k]
            kp0, kp1, scores = kp0[keep], kp1[keep], scores[keep]
        # switch the indices back
        return {"keypoints0": kp1, "keypoints1": kp0, "scores": scores, "batch_indexes": out["batch_indexes"][:n] if top_k is None else None}
