"""Audited comparisons of the parity checks  --  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Index outputs (key-points, matches) must be bit-exact (BASELINE.json north_star).  Two fp32 implementations with
different accumulation orders can still disagree where a decision is a numerical tie; these helpers turn "may differ
only by ties" into an assertion: every difference must be traced to a deciding margin of the ORACLE's own data that is
below the measured round-off, anything else fails.  Used by tests/ and __graft_entry__.smoke().
"""
from __future__ import annotations

import torch

from .superpoint import simple_nms


def assert_matches_equal_or_tied(m_hip, log_assignment, m_ref, filter_threshold, tol=1e-4, max_frac=0.002, tag="", ind0=None, ind1=None):
    """matches0 of the HIP path vs the oracle for ONE pair.

    Equal rows pass.  A differing row must be a numerically fragile decision in the ORACLE's own log-assignment
    matrix `log_assignment` [m+1, n+1] -- its row arg-max, the column arg-max of the winning column, or the
    `exp(score) > filter_threshold` test decided by a margin below `tol` (the same 1e-4 the scores are held to) --
    and there may be at most max(1, max_frac * m) of them.  Anything else fails.  With point pruning the matrix lives
    in the pruned index space: `ind0` / `ind1` (the oracle's `_ind0` / `_ind1`, original index of every surviving
    row / column) map it back; a row the oracle pruned can only be unmatched.  Returns the number of audited ties."""
    m_hip, m_ref = m_hip.long(), m_ref.long()
    bad = torch.nonzero(m_hip != m_ref).flatten().tolist()
    if not bad:
        return 0
    S = log_assignment[:-1, :-1]
    row_of = {int(o): r for r, o in enumerate(ind0.flatten().tolist())} if ind0 is not None else None
    assert len(bad) <= max(1, int(max_frac * len(m_ref))), f"{tag}: {len(bad)} differing rows"
    for i in bad:
        if row_of is not None:
            assert i in row_of, f"{tag}: row {i} was pruned by the oracle but the HIP path matched it to {int(m_hip[i])}"
        row = S[row_of[i] if row_of is not None else i]
        top = torch.topk(row, min(2, row.numel()))
        j = int(top.indices[0])
        g_row = float(top.values[0] - top.values[1]) if row.numel() > 1 else float("inf")
        col = torch.topk(S[:, j], min(2, S.shape[0]))
        g_col = float(col.values[0] - col.values[1]) if S.shape[0] > 1 else float("inf")
        g_thr = abs(float(top.values[0].exp()) - filter_threshold)
        margin = min(g_row, g_col, g_thr)
        assert margin < tol, f"{tag}: row {i}: hip {int(m_hip[i])} vs oracle {int(m_ref[i])}, deciding margin {margin:.3e} is not a tie"
    return len(bad)


def audit_keypoint_differences(flat_hip, flat_ref, dense_hip, dense_ref, conf, tag=""):
    """Every key-point index present in one set and not in the other must be explained by a round-off tie.

    The selection is a chain of exact comparisons on the dense score map (SURVEY.md section 7 hard part 2):
    `simple_nms` compares scores of pixels at most r apart (chained over 5 max-pools: a decision at p depends on
    pixels up to 5r away), then `score > keypoint_threshold`, then the k-th largest score.  The two maps differ by
    at most d = max|dense_hip - dense_ref|, so a decision can only flip where the deciding margin ON THE ORACLE'S
    MAP is below eps = 2d (+1e-7).  For every differing index p this looks for such a margin: |s(p) - thr| < eps,
    |s(p) - s_kth| < eps, or a pair of pixels (a, b) within the 5r window of p, at most r apart, whose order differs
    between the two maps (which implies |s_ref(a) - s_ref(b)| < eps).  Fails if a difference has no such cause."""
    H, W = dense_ref.shape
    r, thr, k = conf["nms_radius"], conf["keypoint_threshold"], conf["max_keypoints"]
    set_h, set_r = set(flat_hip.tolist()), set(flat_ref.tolist())
    diff = sorted(set_h ^ set_r)
    if not diff:
        return 0
    d = float((dense_hip - dense_ref).abs().max())
    eps = 2.0 * d + 1e-7
    assert eps < 1e-4, f"{tag}: dense score maps differ by {d:.3e}"
    nms_ref = simple_nms(dense_ref[None], r)[0]
    keep = nms_ref > thr
    bd = int(conf.get("remove_borders", 0) or 0)
    if bd > 0:  # `remove_borders` runs BEFORE the top-k: the k-th score is taken among the interior candidates
        keep[:bd] = False
        keep[H - bd :] = False
        keep[:, :bd] = False
        keep[:, W - bd :] = False
    cand = nms_ref[keep]
    kth = float(torch.topk(cand.flatten(), k).values[-1]) if (k >= 0 and cand.numel() > k) else None
    kth_next = float(torch.topk(cand.flatten(), k + 1).values[-1]) if (k >= 0 and cand.numel() > k) else None
    for p in diff:
        y, x = divmod(p, W)
        s = float(dense_ref[y, x])
        if abs(s - thr) < eps:
            continue
        if kth is not None and (abs(s - kth) < eps or abs(s - kth_next) < eps):
            continue
        # order flip between two pixels at most r apart inside the 5r window of p
        y0, y1, x0, x1 = max(0, y - 5 * r), min(H, y + 5 * r + 1), max(0, x - 5 * r), min(W, x + 5 * r + 1)
        a_ref, a_hip = dense_ref[y0:y1, x0:x1], dense_hip[y0:y1, x0:x1]
        found = False
        for dy in range(0, r + 1):
            for dx in range(-r, r + 1):
                if dy == 0 and dx <= 0:
                    continue
                hh, ww = a_ref.shape
                ys, xs = slice(0, hh - dy), slice(max(0, -dx), ww - max(0, dx))
                yt, xt = slice(dy, hh), slice(max(0, dx), ww - max(0, -dx))
                dr = a_ref[ys, xs] - a_ref[yt, xt]
                dh = a_hip[ys, xs] - a_hip[yt, xt]
                flip = (torch.sign(dr) != torch.sign(dh)) & (dr.abs() < eps)
                if bool(flip.any()):
                    found = True
                    break
            if found:
                break
        assert found, f"{tag}: key-point {p} (y={y}, x={x}, score {s:.6f}) differs between HIP and oracle without a round-off tie (eps {eps:.2e})"
    return len(diff)
