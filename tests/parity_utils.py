"""Shared helpers of the GPU parity tests (oracle = checker, HIP path = thing under test)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from oracle.audit import assert_matches_equal_or_tied, audit_keypoint_differences  # noqa: F401  (re-exported)
from oracle.superpoint import remove_borders, simple_nms


def canonical_select(score_map_nms: torch.Tensor, thr: float, border: int, max_kpts: int):
    """The reference's selection (nonzero -> remove_borders -> topk) with ties made canonical:
    score descending, flat index ascending.  Returns (flat_idx, scores, boundary_tie)."""
    H, W = score_map_nms.shape
    kp = torch.nonzero(score_map_nms > thr)
    sc = score_map_nms[tuple(kp.t())]
    kp, sc = remove_borders(kp, sc, border, H, W)
    flat = kp[:, 0] * W + kp[:, 1]
    boundary_tie = False
    if max_kpts >= 0 and max_kpts < len(flat):
        order = torch.argsort(flat)  # ascending index first, then stable sort by score
        flat, sc = flat[order], sc[order]
        order = torch.argsort(sc, descending=True, stable=True)
        flat, sc = flat[order], sc[order]
        boundary_tie = bool(sc[max_kpts - 1] == sc[max_kpts])
        flat, sc = flat[:max_kpts], sc[:max_kpts]
    return flat, sc, boundary_tie


def oracle_select_on(dense: torch.Tensor, conf: dict):
    """Run the oracle's NMS + selection on a given dense score map [H,W] (CPU)."""
    nms = simple_nms(dense[None], conf["nms_radius"])[0]
    return canonical_select(nms, conf["keypoint_threshold"], conf["remove_borders"], conf["max_keypoints"]), nms


def synthetic_matching_problem(seed, n, m, n_out, noise=0.05, size=(640, 480)):
    """Key-points / distinctive unit descriptors with known correspondences (CPU tensors)."""
    g = torch.Generator().manual_seed(seed)
    W, H = size
    k0 = torch.rand(n, 2, generator=g) * torch.tensor([W - 8.0, H - 8.0]) + 4
    perm = torch.randperm(n, generator=g)
    perm = perm[torch.arange(m) % n]  # m > n: extra points of image 1 re-use partners (then become outliers)
    k1 = k0[perm] + torch.randn(m, 2, generator=g)
    d0 = F.normalize(torch.randn(n, 256, generator=g), dim=1)
    d1 = F.normalize(d0[perm] + noise * torch.randn(m, 256, generator=g), dim=1)
    n_out = max(n_out, m - n)
    d1[m - n_out :] = F.normalize(torch.randn(n_out, 256, generator=g), dim=1)
    k1[m - n_out :] = torch.rand(n_out, 2, generator=g) * torch.tensor([W - 8.0, H - 8.0]) + 4
    return k0, k1, d0, d1
