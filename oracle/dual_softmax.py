"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's dual-softmax matcher
(`imcui/hloc/matchers/dual_softmax.py:8-41` matcher, `:44-75` plugin), batch 1 as hloc drives it.

Pinned: `tests/golden/ds_*.npz` were produced by running the reference's own module
(`tests/golden/make_golden.py`); `tests/test_oracle_golden.py` checks this restatement against them.

Semantics kept from the reference: descriptors are L2-normalised over the channel axis, the similarity is
scaled by `inv_temperature`, P = softmax over rows (dim -2) times softmax over columns (dim -1), a pair
(i, j) is a match when P[i, j] equals BOTH its row maximum and its column maximum and exceeds the threshold;
when several columns qualify for one row the LAST one wins (the reference scatters `nonzero` results in
row-major order, `:27-32`).  `matching_scores0` is float64 and `matches0` int64 there (numpy defaults).
"""
from __future__ import annotations

import torch


def dual_softmax_oracle(desc0: torch.Tensor, desc1: torch.Tensor, threshold: float = 0.2, inv_temperature: float = 20.0,
                        normalize: bool = True):
    """desc0 [B, C, N], desc1 [B, C, M] float32 -> (matches0 int64 [B, N], scores0 float64 [B, N]), per batch item."""
    B, _, N = desc0.shape
    M = desc1.shape[-1]
    m0 = torch.full((B, N), -1, dtype=torch.int64)
    s0 = torch.zeros((B, N), dtype=torch.float64)
    if N == 0 or M == 0:
        return m0, s0
    for b in range(B):
        a, c = desc0[b].float(), desc1[b].float()
        if normalize:
            a = a / a.norm(dim=0, keepdim=True)
            c = c / c.norm(dim=0, keepdim=True)
        sim = (a.t() @ c) * inv_temperature  # [N, M]
        p = torch.softmax(sim, dim=0) * torch.softmax(sim, dim=1)
        row_best = p.max(dim=1, keepdim=True).values
        col_best = p.max(dim=0, keepdim=True).values
        ok = (p == row_best) & (p == col_best) & (p > threshold)
        cols = torch.arange(M).expand(N, M)
        last = torch.where(ok, cols, torch.full_like(cols, -1)).max(dim=1).values  # largest qualifying column
        hit = last >= 0
        m0[b, hit] = last[hit]
        s0[b, hit] = p[hit, last[hit]].double()
    return m0, s0


class DualSoftMaxOracle:
    default_conf = {"match_threshold": 0.2, "inv_temperature": 20}

    def __init__(self, conf: dict | None = None):
        self.conf = {**self.default_conf, **(conf or {})}

    def __call__(self, data: dict) -> dict:
        if data["descriptors0"].size(-1) == 0 or data["descriptors1"].size(-1) == 0:
            # reference quirk kept (:52-60): the empty answer has the shape of descriptors0.shape[:2] = (B, C) and
            # integer zeros as scores
            m0 = torch.full(data["descriptors0"].shape[:2], -1)
            return {"matches0": m0, "matching_scores0": torch.zeros_like(m0)}
        m0, s0 = dual_softmax_oracle(data["descriptors0"], data["descriptors1"], self.conf["match_threshold"], self.conf["inv_temperature"])
        return {"matches0": m0, "matching_scores0": s0}
