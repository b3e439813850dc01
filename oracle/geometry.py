"""CPU restatement of the batched RANSAC of csrc/geometry.hip ("HIP_RANSAC", an additional entry for the `ransac_zoo` of
imcui/ui/utils.py) -- TEST INFRASTRUCTURE: the checker of the kernels, never imported by the product path.

PARITY UNPINNED with respect to the reference's default verifier: `proc_ransac_matches` (imcui/ui/utils.py:424-456) calls
cv2.findHomography / cv2.findFundamentalMat with USAC_MAGSAC; cv2 is not installed here and its samplers cannot be reproduced, so this
restates OUR specification step by step (same counter-based sampler, same minimal solvers, same error measures, same sequential
stopping rule, same local optimisation), in float64 numpy, one pair at a time.  The GPU tests compare the kernels with it (models to
1e-8, masks equal up to audited threshold ties) and both with ground-truth geometry.
"""
from __future__ import annotations

import math

import numpy as np

MASK64 = (1 << 64) - 1
MAX_HYP = 16384


def geo_hash(seed: int, b: int, k: int, j: int, attempt: int) -> int:
    x = (seed * 0x9E3779B97F4A7C15 + ((b << 40) ^ (k << 8) ^ j) + attempt * 0xD1B54A32D192ED03) & MASK64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & MASK64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & MASK64
    x ^= x >> 31
    return x


def sample(seed: int, b: int, k: int, m: int, n: int) -> list[int]:
    idx: list[int] = []
    for j in range(m):
        attempt = 0
        while True:
            c = geo_hash(seed, b, k, j, attempt) % n
            if attempt >= 16:
                while c in idx:
                    c = (c + 1) % n
            if c not in idx:
                idx.append(c)
                break
            attempt += 1
    return idx


def _hartley(p):
    c = p.mean(0)
    d = np.sqrt(((p - c) ** 2).sum(1)).mean()
    s = math.sqrt(2.0) / max(d, 1e-12)
    return (p - c) * s, np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])


def _rank2(F):
    w, V = np.linalg.eigh(F.T @ F)
    v = V[:, 0]
    return F - np.outer(F @ v, v)


def normalise_model(M, geometry):
    if not np.isfinite(M).all():
        return None
    if geometry == 0:
        if abs(M[2, 2]) < 1e-12:
            return None
        return M / M[2, 2]
    nn = math.sqrt((M * M).sum())
    if not nn > 1e-150:
        return None
    big = np.unravel_index(np.argmax(np.abs(M)), M.shape)
    return M * ((-1.0 if M[big] < 0 else 1.0) / nn)


def minimal_model(p0, p1, geometry):
    """4 (homography) / 8 (fundamental) correspondences -> 3x3 float64 or None (degenerate sample)."""
    a, T0 = _hartley(p0.astype(np.float64))
    c, T1 = _hartley(p1.astype(np.float64))
    x, y, u, v = a[:, 0], a[:, 1], c[:, 0], c[:, 1]
    one, zero = np.ones_like(x), np.zeros_like(x)
    if geometry == 0:
        A = np.zeros((8, 8))
        rhs = np.zeros(8)
        A[0::2] = np.stack([x, y, one, zero, zero, zero, -u * x, -u * y], 1)
        A[1::2] = np.stack([zero, zero, zero, x, y, one, -v * x, -v * y], 1)
        rhs[0::2], rhs[1::2] = u, v
        # the device eliminates with partial pivoting and gives up below a pivot of 1e-10: mirror the test on the LU pivots
        try:
            import scipy.linalg as sl

            lu, _ = sl.lu_factor(A)
            if np.abs(np.diag(lu)).min() < 1e-10:
                return None
            h = sl.lu_solve((lu, _), rhs)
        except Exception:  # noqa: BLE001
            return None
        M = np.linalg.inv(T1) @ np.append(h, 1.0).reshape(3, 3) @ T0
    else:
        A = np.stack([u * x, u * y, u, v * x, v * y, v, x, y, one], 1)
        # null vector by complete pivoting as on the device (a tiny pivot = a degenerate sample)
        A = A.copy()
        perm = list(range(9))
        for col in range(8):
            sub = np.abs(A[col:, col:])
            pr, pc = np.unravel_index(np.argmax(sub), sub.shape)
            pr, pc = pr + col, pc + col
            if abs(A[pr, pc]) < 1e-10:
                return None
            A[[col, pr]] = A[[pr, col]]
            A[:, [col, pc]] = A[:, [pc, col]]
            perm[col], perm[pc] = perm[pc], perm[col]
            f = A[col + 1 :, col] / A[col, col]
            A[col + 1 :, col:] -= np.outer(f, A[col, col:])
        z = np.zeros(9)
        z[8] = 1.0
        for r in range(7, -1, -1):
            z[r] = (-A[r, 8] - A[r, r + 1 : 8] @ z[r + 1 : 8]) / A[r, r]
        f9 = np.zeros(9)
        for t in range(9):
            f9[perm[t]] = z[t]
        M = T1.T @ _rank2(f9.reshape(3, 3)) @ T0
    return normalise_model(M, geometry)


def errors2(M, p0, p1, geometry):
    x, y, u, v = (p0[:, 0].astype(np.float64), p0[:, 1].astype(np.float64), p1[:, 0].astype(np.float64), p1[:, 1].astype(np.float64))
    a = M[0, 0] * x + M[0, 1] * y + M[0, 2]
    b = M[1, 0] * x + M[1, 1] * y + M[1, 2]
    c = M[2, 0] * x + M[2, 1] * y + M[2, 2]
    if geometry == 0:
        bad = np.abs(c) < 1e-12
        cc = np.where(bad, 1.0, c)
        e = (a / cc - u) ** 2 + (b / cc - v) ** 2
        return np.where(bad, 1e300, e)
    e = u * a + v * b + c
    ta = M[0, 0] * u + M[1, 0] * v + M[2, 0]
    tb = M[0, 1] * u + M[1, 1] * v + M[2, 1]
    den = a * a + b * b + ta * ta + tb * tb
    return np.where(den > 1e-300, e * e / np.where(den > 1e-300, den, 1.0), 1e300)


def refit(p0, p1, mask, geometry):
    m = 4 if geometry == 0 else 8
    if mask.sum() < m:
        return None
    a, T0 = _hartley(p0[mask].astype(np.float64))
    c, T1 = _hartley(p1[mask].astype(np.float64))
    x, y, u, v = a[:, 0], a[:, 1], c[:, 0], c[:, 1]
    one, zero = np.ones_like(x), np.zeros_like(x)
    if geometry == 0:
        R = np.concatenate([np.stack([-x, -y, -one, zero, zero, zero, u * x, u * y, u], 1), np.stack([zero, zero, zero, -x, -y, -one, v * x, v * y, v], 1)], 0)
    else:
        R = np.stack([u * x, u * y, u, v * x, v * y, v, x, y, one], 1)
    w, V = np.linalg.eigh(R.T @ R)
    M = V[:, 0].reshape(3, 3)
    M = np.linalg.inv(T1) @ M @ T0 if geometry == 0 else T1.T @ _rank2(M) @ T0
    return normalise_model(M, geometry)


REFIT_ROUNDS = 3


def ransac(p0, p1, geometry: int, reproj_threshold: float, confidence: float, max_iter: int, seed: int = 0, pair_index: int = 0, return_counts: bool = False):
    """One pair.  -> (model 3x3 float64 or None, mask [n] bool, info dict(inliers, used, best_k))."""
    n = len(p0)
    m = 4 if geometry == 0 else 8
    K = min(max_iter, MAX_HYP)
    none = (None, np.zeros(n, bool), dict(inliers=0, used=0, best_k=-1))
    if n < m:
        return none
    thr2 = reproj_threshold * reproj_threshold
    conf = min(max(confidence, 0.0), 0.999999999)
    lc = math.log(1.0 - conf)
    best, bestk, needed, used, models, counts = 0, -1, K, 0, {}, []
    k = 0
    while k < K and k < needed:
        used = k + 1
        idx = sample(seed, pair_index, k, m, n)
        M = minimal_model(p0[idx], p1[idx], geometry)
        c = -1 if M is None else int((errors2(M, p0, p1, geometry) < thr2).sum())
        counts.append(c)
        if c > best:
            best, bestk, models[k] = c, k, M
            wm = (c / n) ** m
            if wm >= 1.0 - 1e-15:
                needed = 0
            elif wm > 1e-300:
                # log1p: 1.0 - wm == 1.0 below 2^-53 (same fix as csrc/geometry.hip); no negative denominator = no bound
                den = math.log1p(-wm)
                it = math.ceil(lc / den) if den < 0.0 else K
                needed = int(max(it, 0)) if it < K else K
        k += 1
    if bestk < 0 or best < m:
        return none
    M = models[bestk]
    mask = errors2(M, p0, p1, geometry) < thr2
    for _ in range(REFIT_ROUNDS):  # local optimisation: least-squares refits on the current inliers (csrc/geometry.hip: GEO_REFIT_ROUNDS)
        M2 = refit(p0, p1, mask, geometry)
        if M2 is None:
            continue
        mask2 = errors2(M2, p0, p1, geometry) < thr2
        if mask2.sum() >= mask.sum():
            M, mask = M2, mask2
    info = dict(inliers=int(mask.sum()), used=used, best_k=bestk)
    if return_counts:
        info["counts"] = counts
    return M, mask, info
