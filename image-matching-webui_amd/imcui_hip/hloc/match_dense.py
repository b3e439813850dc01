"""Batched dense matching driver on the HIP plugins (missing #6).

What `imcui/hloc/match_dense.py:196-253` (`match_dense`) does with a `DataLoader(batch_size=1)` and one `model(data)` call
per pair, done here with B pairs of one size per C-ABI call:

  * preprocessing of `ImagePairDataset.preprocess` (:155-186): gray image, `resize_max` shrink with "cv2_area" (on the
    device, `backend.preprocess_area`), `/ 255`, then the size is made divisible by `dfactor` with a bilinear resize
    (`torchvision.transforms.functional.resize`, i.e. `interpolate(mode="bilinear", antialias=True)`), and
    `scale = original size / final size`;
  * the semi-dense matcher is called the way the reference wrapper is (`model({"image0", "image1"})`, which refines the
    key-points of image0), or flipped when `name0` is a reference image with existing key-points (`existing_refs`, :221-231);
  * key-points go back to the original resolution with `(k + 0.5) * scale - 0.5` (:236-238) and every pair gets a group
    `names_to_pair(name0, name1)` with `keypoints0`, `keypoints1`, `scores` in the match file (:243-251).

The association steps that follow in the reference (`aggregate_matches`, `assign_matches`: KD-trees, binning, h5 edits)
are host code that stays the reference's; they read the file this driver writes.
"""
from __future__ import annotations

from pathlib import Path
from types import SimpleNamespace
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import backend
from .extract_features import preprocess_on_device, read_image_device, read_images_device, read_image_u8  # noqa: F401  (read_image_u8: re-exported for callers)
from .match_features import names_to_pair
from .utils.h5lite import open_h5

DEFAULT_PREPROCESSING = {"grayscale": True, "resize_max": 1024, "dfactor": 8, "cache_images": False}  # ImagePairDataset.default_conf


def preprocess_pair_image(img_u8, conf: SimpleNamespace, device) -> Tuple[torch.Tensor, np.ndarray]:
    """uint8 image (host array or device tensor) -> ([1,1,h,w] float32 on the device, scale (x, y) back to the original resolution)."""
    if not conf.grayscale:
        raise NotImplementedError("the HIP dense matchers take gray images (every dense entry of the zoo sets grayscale: True)")
    h, w = tuple(img_u8.shape[:2])
    area = SimpleNamespace(grayscale=True, resize_max=conf.resize_max if conf.resize_max and conf.resize_max < max(h, w) else None,
                           force_resize=False, interpolation="cv2_area")  # fmt: skip
    image = preprocess_on_device(img_u8, area, device)
    hh, ww = image.shape[-2:]
    size_new = (int(hh // conf.dfactor * conf.dfactor), int(ww // conf.dfactor * conf.dfactor))
    image = backend.resize_aa(image, size_new)  # F.resize(image, size=size_new, antialias=True), match_dense.py:182
    scale = np.array([w, h], dtype=np.float64) / np.array(size_new[::-1], dtype=np.float64)
    return image.contiguous(), scale


def _rescale(kpts: torch.Tensor, scale: np.ndarray) -> np.ndarray:
    k = kpts + 0.5
    if np.any(scale != 1.0):
        k = k * k.new_tensor(scale)
    return (k - 0.5).cpu().numpy()


@torch.no_grad()
def match_dense(conf: Dict, pairs: Sequence[Tuple[str, str]], image_dir: Path, match_path: Path, existing_refs: Iterable[str] = (),
                model=None, batch_size: int = 8, device="cuda", decode: str = "auto") -> Path:  # fmt: skip
    """Reference signature (:196-202) + `model` (a loaded HIP dense matcher plugin; built from conf["model"] when None),
    `batch_size` (pairs per C-ABI call) and `decode` (extract_features.read_image_device: "auto" = baseline JPEGs decoded on the
    device).  Returns the match file path."""
    if model is None:
        from . import matchers
        from .utils.base_model import dynamic_load

        model = dynamic_load(matchers, conf["model"]["name"])(conf["model"]).eval().to(device)
    device = next(model.buffers()).device
    pconf = SimpleNamespace(**{**DEFAULT_PREPROCESSING, **conf.get("preprocessing", {})})
    image_dir, match_path = Path(image_dir), Path(match_path)
    match_path.parent.mkdir(exist_ok=True, parents=True)
    existing_refs = set(existing_refs or ())
    cache: Dict[str, Tuple[torch.Tensor, np.ndarray]] = {}

    decoder = None  # one JpegDecoder (pinned staging + host threads) for the whole run

    def prefetch(chunk):
        """Read every image of `chunk` (a list of pairs) that is not cached yet through ONE `read_images_device` call: the baseline
        JPEGs among them share a batched entropy decode, one transfer and three launches per geometry (one file per call before)."""
        nonlocal decoder
        need = []
        for pr in chunk:
            for n in pr:
                if n not in cache and n not in need:
                    need.append(n)
        if not need:
            return
        if not pconf.cache_images:  # bounded cache: drop the oldest entries this chunk does not use
            keep = {n for pr in chunk for n in pr}
            for n in [n for n in cache if n not in keep]:
                if len(cache) + len(need) <= 4 * batch_size:
                    break
                cache.pop(n)
        if decode != "host" and decoder is None:
            from .utils.jpeg import JpegDecoder

            decoder = JpegDecoder(device)
        imgs = read_images_device([image_dir / n for n in need], pconf.grayscale, device, decode, decoder)
        for n, im in zip(need, imgs):
            cache[n] = preprocess_pair_image(im, pconf, device)

    def load(name):
        if name not in cache:  # (not reached after prefetch; kept for callers that extend the loop)
            cache[name] = preprocess_pair_image(read_image_device(image_dir / name, pconf.grayscale, device, decode), pconf, device)
        return cache[name]

    pending: Dict[tuple, list] = {}

    def flush(key):
        items = pending.pop(key, [])
        if not items:
            return
        preds = model.forward_pairs(torch.cat([it[2] for it in items], 0), torch.cat([it[3] for it in items], 0))
        with open_h5(match_path, "a") as fd:
            for (name0, name1, _, _, s0, s1, flip), pred in zip(items, preds):
                k0, k1 = (pred["keypoints1"], pred["keypoints0"]) if flip else (pred["keypoints0"], pred["keypoints1"])
                pair = names_to_pair(name0, name1)
                if pair in fd:
                    del fd[pair]
                grp = fd.create_group(pair)
                grp.create_dataset("keypoints0", data=_rescale(k0, s0))
                grp.create_dataset("keypoints1", data=_rescale(k1, s1))
                grp.create_dataset("scores", data=pred["scores"].cpu().numpy())

    pairs = list(pairs)
    try:
        for c0 in range(0, len(pairs), batch_size):
            chunk = pairs[c0 : c0 + batch_size]
            prefetch(chunk)
            for name0, name1 in chunk:
                im0, s0 = load(name0)
                im1, s1 = load(name1)
                flip = name0 in existing_refs  # refine the key-points of the query (image1) instead: call with the images exchanged
                a, b = (im1, im0) if flip else (im0, im1)
                key = (tuple(a.shape[-2:]), tuple(b.shape[-2:]), flip)
                pending.setdefault(key, []).append((name0, name1, a, b, s0, s1, flip))
                if len(pending[key]) >= batch_size:
                    flush(key)
        for key in list(pending):
            flush(key)
    finally:
        if decoder is not None:
            decoder.close()
    return match_path
