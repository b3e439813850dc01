"""Batched pair matching driver on the HIP plugins.

What imcui/hloc/match_features.py:118-185 (`main` / `match_from_paths`) does with `batch_size=1` and one
`model(data)` call per pair, done here with B pairs per `forward_batched` call: the pair list is parsed and
de-duplicated exactly like the reference (`parse_retrieval` utils/parsers.py:43-51, `names_to_pair` :54-59,
`find_unique_new_pairs` match_features.py:118-138), features are read through a small store interface, pairs are
grouped by image size (the C ABI normalises key-points with one (W, H) per call) and packed into fixed-stride
batches, and every pair's `matches0` (int16) / `matching_scores0` (float16) goes to a sink in the reference's
on-disk layout (`writer_fn` :73-83: group `name0/name1`).

The HDF5 store / sink open files through `utils.h5lite.open_h5`: `h5py` (the reference's own dependency) when it is
installed, the HDF5 C library through ctypes otherwise -- real HDF5 files either way, ImportError when neither exists
(there is no silent alternative format).  `DictFeatureStore` / `DictMatchSink` are the in-memory equivalents used by
pipelines that keep features resident and by the tests.
"""
from __future__ import annotations

from collections import defaultdict
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .utils.h5lite import open_h5


# ------------------------------------------------------------------ pair lists (imcui/hloc/utils/parsers.py:43-63)
def parse_retrieval(path) -> Dict[str, List[str]]:
    """`query reference` per line -> {query: [references...]} (insertion ordered)."""
    retrieval = defaultdict(list)
    with open(path, "r") as fh:
        for line in fh.read().rstrip("\n").split("\n"):
            if len(line) == 0:
                continue
            q, r = line.split()
            retrieval[q].append(r)
    return dict(retrieval)


def names_to_pair(name0: str, name1: str, separator: str = "/") -> str:
    return separator.join((name0.replace("/", "-"), name1.replace("/", "-")))


def names_to_pair_old(name0: str, name1: str) -> str:
    return names_to_pair(name0, name1, separator="_")


def find_unique_new_pairs(pairs_all: Iterable[Tuple[str, str]], done=None) -> List[Tuple[str, str]]:
    """Drop (j, i) when (i, j) is present and pairs the sink already holds (either order, old or new key style).
    Unlike the reference (which iterates a `set`), the input order is kept, so runs are reproducible."""
    seen, pairs = set(), []
    for i, j in pairs_all:
        if (j, i) in seen or (i, j) in seen:
            continue
        seen.add((i, j))
        pairs.append((i, j))
    if done is None:
        return pairs
    out = []
    for i, j in pairs:
        if any(k in done for k in (names_to_pair(i, j), names_to_pair(j, i), names_to_pair_old(i, j), names_to_pair_old(j, i))):
            continue
        out.append((i, j))
    return out


# ------------------------------------------------------------------ feature stores / match sinks
class DictFeatureStore:
    """name -> {"keypoints" [N,2], "scores" [N], "descriptors" [D,N], "image_size" (W, H)} held in memory
    (numpy or torch, any float dtype) -- the per-image groups of the reference's feature file
    (imcui/hloc/extract_features.py:221-243)."""

    def __init__(self, features: Dict[str, dict]):
        self.features = features

    def __contains__(self, name):
        return name in self.features

    def get(self, name: str) -> dict:
        return self.features[name]


class H5FeatureStore:
    """The reference's feature file: one group per image with `keypoints`, `scores`, `descriptors`, `image_size`."""

    def __init__(self, path):
        self.path = Path(path)
        if not self.path.exists():
            raise FileNotFoundError(f"Feature file {self.path}.")
        open_h5(self.path, "r").close()  # ImportError here is the loud failure: no alternative format

    def __contains__(self, name):
        with open_h5(self.path, "r") as fd:
            return name in fd

    def get(self, name: str) -> dict:
        with open_h5(self.path, "r") as fd:
            return {k: v.__array__() for k, v in fd[name].items()}


class DictMatchSink:
    """pair key -> {"matches0": int16 [N], "matching_scores0": float16 [N]} (the datasets `writer_fn` creates)."""

    def __init__(self):
        self.matches: Dict[str, dict] = {}

    def __contains__(self, pair):
        return pair in self.matches

    def put(self, pair: str, matches0: np.ndarray, scores0: Optional[np.ndarray]):
        rec = {"matches0": matches0}
        if scores0 is not None:
            rec["matching_scores0"] = scores0
        self.matches[pair] = rec


class H5MatchSink:
    """The reference's match file (imcui/hloc/match_features.py:73-83)."""

    def __init__(self, path):
        self.path = Path(path)
        self.path.parent.mkdir(exist_ok=True, parents=True)
        open_h5(self.path, "r" if self.path.exists() else "a").close()  # no HDF5 library -> ImportError here, not at the first put()

    def __contains__(self, pair):
        if not self.path.exists():
            return False
        with open_h5(self.path, "r") as fd:
            return pair in fd

    def put(self, pair: str, matches0: np.ndarray, scores0: Optional[np.ndarray]):
        with open_h5(self.path, "a") as fd:
            if pair in fd:
                del fd[pair]
            grp = fd.create_group(pair)
            grp.create_dataset("matches0", data=matches0)
            if scores0 is not None:
                grp.create_dataset("matching_scores0", data=scores0)


# ------------------------------------------------------------------ batching
def _as_tensor(x, device) -> torch.Tensor:
    t = torch.from_numpy(np.asarray(x)) if not isinstance(x, torch.Tensor) else x
    return t.to(device=device, dtype=torch.float32)


def _size_key(feat: dict) -> Tuple[int, int]:
    w, h = (int(v) for v in np.asarray(feat["image_size"]).reshape(-1)[:2])
    return w, h


def make_batches(pairs: Sequence[Tuple[str, str]], feats_q, feats_r, batch_size: int):
    """Group pairs by ((W0,H0),(W1,H1)) -- one C-ABI call normalises with one size per side -- and cut every group
    into chunks of at most `batch_size` pairs, keeping the list order inside a group.  Yields lists of pair indices."""
    groups: Dict[tuple, List[int]] = {}
    for idx, (q, r) in enumerate(pairs):
        groups.setdefault((_size_key(feats_q.get(q)), _size_key(feats_r.get(r))), []).append(idx)
    for key, idxs in groups.items():
        for s in range(0, len(idxs), batch_size):
            yield key, idxs[s : s + batch_size]


def collate(pairs, idxs, feats_q, feats_r, device):
    """Ragged per-image features -> fixed-stride batch: keypoints [B,ncap,2], scores [B,ncap], descriptors [B,ncap,D]
    (row per key-point, the layout of `forward_batched`), counts n0 / n1 [B] int32."""
    f0 = [feats_q.get(pairs[i][0]) for i in idxs]
    f1 = [feats_r.get(pairs[i][1]) for i in idxs]
    B = len(idxs)
    c0 = [int(f["keypoints"].shape[0]) for f in f0]
    c1 = [int(f["keypoints"].shape[0]) for f in f1]
    ncap = max(max(c0), max(c1), 1)
    D = int(f0[0]["descriptors"].shape[0])
    out = {}
    for side, (fs, cs) in enumerate(((f0, c0), (f1, c1))):
        k = torch.zeros(B, ncap, 2, device=device)
        s = torch.zeros(B, ncap, device=device)
        d = torch.zeros(B, ncap, D, device=device)
        for b, (f, n) in enumerate(zip(fs, cs)):
            if n == 0:
                continue
            k[b, :n] = _as_tensor(f["keypoints"], device)
            if "scores" in f:
                s[b, :n] = _as_tensor(f["scores"], device)
            d[b, :n] = _as_tensor(f["descriptors"], device).t()
        out[f"keypoints{side}"], out[f"scores{side}"], out[f"descriptors{side}"] = k, s, d
        out[f"n{side}"] = torch.tensor(cs, dtype=torch.int32, device=device)
    return out, c0, c1


def _call_batched(model, batch: dict, size0, size1) -> dict:
    """LightGlue-style plugins take (kpts, desc, n, sizes); SuperGlue-style ones also take the detector scores."""
    import inspect

    params = inspect.signature(model.forward_batched).parameters
    if "scores0" in params:
        return model.forward_batched(batch["keypoints0"], batch["keypoints1"], batch["scores0"], batch["scores1"], batch["descriptors0"],
                                     batch["descriptors1"], batch["n0"], batch["n1"], size0, size1)  # fmt: skip
    return model.forward_batched(batch["keypoints0"], batch["keypoints1"], batch["descriptors0"], batch["descriptors1"], batch["n0"],
                                 batch["n1"], size0, size1)  # fmt: skip


@torch.no_grad()
def match_from_pairs(model, pairs: Sequence[Tuple[str, str]], feats_q, feats_r, sink, batch_size: int = 32, device=None) -> int:
    """Match every pair through `model.forward_batched`, B pairs per call, and hand each result to `sink.put`
    under the reference's pair key.  Returns the number of pairs matched."""
    if device is None:
        try:
            device = next(model.buffers()).device
        except StopIteration:
            device = torch.device("cpu")
    done = 0
    for (size0, size1), idxs in make_batches(pairs, feats_q, feats_r, batch_size):
        batch, c0, _ = collate(pairs, idxs, feats_q, feats_r, device)
        pred = _call_batched(model, batch, size0, size1)
        m0 = pred["matches0"].to("cpu", torch.int16).numpy()  # -1 = unmatched; int16 on disk (N, M < 32768)
        s0 = pred["matching_scores0"].to("cpu", torch.float16).numpy() if "matching_scores0" in pred else None
        for b, i in enumerate(idxs):
            n = c0[b]
            sink.put(names_to_pair(*pairs[i]), m0[b, :n].copy(), None if s0 is None else s0[b, :n].copy())
            done += 1
    return done


def match_from_paths(model, pairs_path, match_path, feature_path_q, feature_path_ref, overwrite: bool = False, batch_size: int = 32) -> Path:
    """File-based driver with the reference's signature order (match_features.py:141-185); `model` is the loaded plugin."""
    pairs_path, match_path = Path(pairs_path), Path(match_path)
    feats_q = H5FeatureStore(feature_path_q)
    feats_r = feats_q if Path(feature_path_ref) == Path(feature_path_q) else H5FeatureStore(feature_path_ref)
    sink = H5MatchSink(match_path)
    assert pairs_path.exists(), pairs_path
    pairs = [(q, r) for q, rs in parse_retrieval(pairs_path).items() for r in rs]
    pairs = find_unique_new_pairs(pairs, None if overwrite else sink)
    if len(pairs) > 0:
        match_from_pairs(model, pairs, feats_q, feats_r, sink, batch_size=batch_size)
    return match_path
