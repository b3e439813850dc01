"""Mutual nearest-neighbour oracle  --  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the reference matcher imcui/hloc/matchers/nearest_neighbor.py
(`find_nn` :6-16, `mutual_check` :19-24, `NearestNeighbor._forward` :38-66),
written as explicit best / second-best selection rather than `topk`.

PINNED: tests/golden/nn_*.npz were produced by running the reference module itself
in the build container (tests/golden/make_golden.py); tests/test_oracle_golden.py
checks this restatement against them bit-for-bit.
"""
from __future__ import annotations

import torch

DEFAULT_CONF = {"ratio_threshold": None, "distance_threshold": None, "do_mutual_check": True}


def _best_two(sim: torch.Tensor):
    """Row-wise best and second-best similarity (values, index of best).

    Equivalent to `sim.topk(2)` of the reference (:7); ties resolve to the lowest
    column like torch's CPU topk does for the leading element.
    """
    best_val, best_idx = sim.max(dim=-1)
    if sim.shape[-1] > 1:
        masked = sim.clone()
        masked.scatter_(-1, best_idx.unsqueeze(-1), float("-inf"))
        second_val = masked.max(dim=-1).values
    else:
        second_val = None
    return best_val, best_idx, second_val


def _one_way(sim, ratio, dist):
    """reference `find_nn` (:6-16): accepted index (or -1) and score per row."""
    best_val, best_idx, second_val = _best_two(sim)
    d_best = 2 * (1 - best_val)
    accept = torch.ones_like(best_idx, dtype=torch.bool)
    if ratio:
        d_second = 2 * (1 - second_val)
        accept &= d_best <= (ratio**2) * d_second
    if dist:
        accept &= d_best <= dist**2
    matches = torch.where(accept, best_idx, torch.full_like(best_idx, -1))
    scores = torch.where(accept, (best_val + 1) / 2, torch.zeros_like(best_val))
    return matches, scores


@torch.no_grad()
def mutual_nn(data: dict, conf: dict | None = None) -> dict:
    conf = {**DEFAULT_CONF, **(conf or {})}
    d0 = data["descriptors0"].float().cpu()
    d1 = data["descriptors1"].float().cpu()
    # empty side: everything unmatched (:39-48)
    if d0.size(-1) == 0 or d1.size(-1) == 0:
        m0 = torch.full(d0.shape[:2], -1)
        return {"matches0": m0, "matching_scores0": torch.zeros_like(m0)}
    ratio = conf["ratio_threshold"]
    if d0.size(-1) == 1 or d1.size(-1) == 1:  # no second neighbour to compare with (:50-51)
        ratio = None
    sim = torch.einsum("bdn,bdm->bnm", d0, d1)  # (:52)
    m0, s0 = _one_way(sim, ratio, conf["distance_threshold"])
    if conf["do_mutual_check"]:  # (:56-62, mutual_check :19-24)
        m1, _ = _one_way(sim.transpose(1, 2), ratio, conf["distance_threshold"])
        back = torch.gather(m1, -1, m0.clamp(min=0))
        rows = torch.arange(m0.shape[-1]).expand_as(m0)
        m0 = torch.where((m0 > -1) & (back == rows), m0, torch.full_like(m0, -1))
    return {"matches0": m0, "matching_scores0": s0}
