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


# ---- the oracle's own fp32 spread ------------------------------------------------------------
def tree_spread(a, b, keys=None) -> float:
    """Largest relative difference |a - b| / max|a| over the floating-point tensor leaves of two equally shaped nested results (dicts /
    lists / tuples); leaves whose shapes differ (a match list that flipped at a threshold) are skipped.  `keys`: restrict a top-level
    dict to these entries."""
    if torch.is_tensor(a):
        if not torch.is_tensor(b) or not a.dtype.is_floating_point or a.numel() == 0 or a.shape != b.shape:
            return 0.0
        return (a - b).abs().max().item() / max(a.abs().max().item(), 1e-30)
    if isinstance(a, dict):
        return max([tree_spread(a[k], b[k]) for k in a if k in b and (keys is None or k in keys)] + [0.0])
    if isinstance(a, (list, tuple)):
        return max([tree_spread(x, y) for x, y in zip(a, b)] + [0.0])
    return 0.0


def oracle_spread(run, threads=(1, 8, 32), keys=None):
    """`run()` = one evaluation of a CPU oracle (a nested result of tensors).  Evaluates it at several intra-op thread counts -- torch's
    CPU kernels split and order their fp32 reductions by thread count, so these are equally valid fp32 evaluations of the SAME network --
    and returns (largest relative difference between any two of them, the last result).  A parity tolerance below this spread would
    reject the reference against itself; the GPU tests use max(1e-4, 3 x spread) where they used hand-picked 2e-4 .. 5e-4 before."""
    old = torch.get_num_threads()
    outs = []
    try:
        for t in threads:
            torch.set_num_threads(t)
            outs.append(run())
    finally:
        torch.set_num_threads(old)
    worst = 0.0
    for i in range(len(outs)):
        for j in range(i + 1, len(outs)):
            worst = max(worst, tree_spread(outs[i], outs[j], keys))
    return worst, outs[-1]
