"""Batched geometric verification on the device: "HIP_RANSAC", an additional method for the reference's
`ransac_zoo` (imcui/ui/utils.py), with the call surface of the functions it stands beside:

  * `proc_ransac_matches(mkpts0, mkpts1, ransac_method, reproj_threshold, confidence, max_iter, geometry_type)` -> (M, mask)   (:424-456)
  * `compute_geometry(pred, ...)` -> {"Fundamental", "mask_f", "Homography", "mask_h"}                                          (:532-610)
  * `filter_matches(pred, ...)` -> pred with "mmkeypoints0_orig", "mmkeypoints1_orig", "mmconf", "H", "geom_info"               (:459-530)

and `ransac_batched`, the form the pipelines use: B pairs, both estimates, no host round trip until the caller asks for one.  The
reference's own methods (cv2 USAC / MAGSAC, poselib) keep running on the host, unchanged; cv2's samplers cannot be reproduced, so this
is a new method, not a re-implementation (csrc/geometry.hip states the algorithm; oracle/geometry.py is its CPU restatement).
`H1` / `H2` of `compute_geometry` come from cv2.stereoRectifyUncalibrated in the reference; they are added when cv2 is importable.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np
import torch

from . import backend

DEFAULT_RANSAC_REPROJ_THRESHOLD = 8
DEFAULT_RANSAC_CONFIDENCE = 0.9999
DEFAULT_RANSAC_MAX_ITER = 10000
DEFAULT_MIN_NUM_MATCHES = 4
GEOMETRY = {"Homography": 0, "Fundamental": 1}
_ws = backend._Workspace()
_lock = __import__("threading").Lock()  # the workspace table is shared by the UI's worker threads (buffers are per device and stream)


def ransac_batched(mkpts0: torch.Tensor, mkpts1: torch.Tensor, counts: Optional[torch.Tensor] = None, geometry_type: str = "Homography",
                   reproj_threshold: float = DEFAULT_RANSAC_REPROJ_THRESHOLD, confidence: float = DEFAULT_RANSAC_CONFIDENCE,
                   max_iter: int = DEFAULT_RANSAC_MAX_ITER, seed: int = 0) -> Dict[str, torch.Tensor]:  # fmt: skip
    """mkpts0 / mkpts1 [B,N,2] float32 on the device (row i of pair b = one correspondence, the first counts[b] rows valid) ->
    {"model" [B,3,3] float64, "mask" [B,N] bool, "num_inliers" [B] int32, "iterations" [B] int32, "ok" [B] bool}, all on the device."""
    import ctypes as C

    if geometry_type not in GEOMETRY:
        raise NotImplementedError(geometry_type)
    dev = mkpts0.device
    hd = backend.get_handle(dev)
    p0, p1 = mkpts0.contiguous().float(), mkpts1.contiguous().float()
    if p0.dim() != 3 or p0.shape != p1.shape or p0.shape[-1] != 2:
        raise backend.ImcuiHipError(f"ransac_batched expects two [B,N,2] tensors, got {tuple(p0.shape)} and {tuple(p1.shape)}")
    B, N, _ = p0.shape
    cnt = torch.full((B,), N, dtype=torch.int32, device=dev) if counts is None else counts.to(device=dev, dtype=torch.int32).contiguous()
    model = torch.zeros((B, 3, 3), dtype=torch.float64, device=dev)
    mask = torch.zeros((B, max(N, 1)), dtype=torch.uint8, device=dev)
    info = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    if B > 0 and N > 0:
        nbytes = hd.lib.imcui_hip_ransac_workspace_bytes(B, N, int(max_iter))
        with _lock:
            ws = _ws.get(nbytes, dev)
        with torch.cuda.device(dev):
            rc = hd.lib.imcui_hip_ransac(hd.h, backend._ptr(p0), backend._ptr(p1), backend._ptr(cnt), B, N, GEOMETRY[geometry_type], float(reproj_threshold),
                                         float(confidence), int(max_iter), C.c_ulonglong(int(seed) & ((1 << 64) - 1)), backend._ptr(model), backend._ptr(mask),
                                         backend._ptr(info), backend._ptr(ws), ws.numel(), backend._stream_ptr())  # fmt: skip
            hd.check(rc, "imcui_hip_ransac")
    return {"model": model, "mask": mask[:, :N].bool(), "num_inliers": info[:, 0], "iterations": info[:, 1], "ok": info[:, 3].bool()}


def proc_ransac_matches(mkpts0: np.ndarray, mkpts1: np.ndarray, ransac_method: str = "HIP_RANSAC", ransac_reproj_threshold: float = 3.0,
                        ransac_confidence: float = 0.99, ransac_max_iter: int = 2000, geometry_type: str = "Homography", device="cuda", seed: int = 0):  # fmt: skip
    """The reference's per-pair call (:424-456) for the method "HIP_RANSAC": numpy in, (M [3,3] float64 | None, mask [N] bool | None) out,
    like `_filter_matches_opencv` (:326-378)."""
    if not ransac_method.startswith("HIP"):
        raise NotImplementedError(f"{ransac_method}: the host methods of the reference's ransac_zoo are not replaced")
    k0 = torch.as_tensor(np.asarray(mkpts0, dtype=np.float32)).reshape(1, -1, 2).to(device)
    k1 = torch.as_tensor(np.asarray(mkpts1, dtype=np.float32)).reshape(1, -1, 2).to(device)
    out = ransac_batched(k0, k1, None, geometry_type, ransac_reproj_threshold, ransac_confidence, ransac_max_iter, seed)
    if not bool(out["ok"][0]):
        return None, None
    return out["model"][0].cpu().numpy(), out["mask"][0].cpu().numpy()


def compute_geometry(pred: Dict[str, Any], ransac_method: str = "HIP_RANSAC", ransac_reproj_threshold: float = DEFAULT_RANSAC_REPROJ_THRESHOLD,
                     ransac_confidence: float = DEFAULT_RANSAC_CONFIDENCE, ransac_max_iter: int = DEFAULT_RANSAC_MAX_ITER, device="cuda") -> Dict[str, Any]:  # fmt: skip
    """imcui/ui/utils.py:532-610: fundamental matrix, homography (+ the rectifying pair when cv2 is there) of one pair's matches."""
    if "mkeypoints0_orig" in pred and "mkeypoints1_orig" in pred:
        mkpts0, mkpts1 = pred["mkeypoints0_orig"], pred["mkeypoints1_orig"]
    elif "line_keypoints0_orig" in pred and "line_keypoints1_orig" in pred:
        mkpts0, mkpts1 = pred["line_keypoints0_orig"], pred["line_keypoints1_orig"]
    else:
        return {}
    if len(mkpts0) < 2 * DEFAULT_MIN_NUM_MATCHES:
        return {}
    geo: Dict[str, Any] = {}
    F, mask_f = proc_ransac_matches(mkpts0, mkpts1, ransac_method, ransac_reproj_threshold, ransac_confidence, ransac_max_iter, "Fundamental", device)
    if F is not None:
        geo["Fundamental"], geo["mask_f"] = F.tolist(), mask_f
    H, mask_h = proc_ransac_matches(mkpts0, mkpts1, ransac_method, ransac_reproj_threshold, ransac_confidence, ransac_max_iter, "Homography", device)
    if H is not None:
        geo["Homography"], geo["mask_h"] = H.tolist(), mask_h
        if F is not None and "image0_orig" in pred:
            try:
                import cv2

                h0, w0 = pred["image0_orig"].shape[:2]
                _, H1, H2 = cv2.stereoRectifyUncalibrated(np.asarray(mkpts0).reshape(-1, 2), np.asarray(mkpts1).reshape(-1, 2), F, imgSize=(w0, h0))
                geo["H1"], geo["H2"] = H1.tolist(), H2.tolist()
            except ImportError:
                pass  # the rectifying homographies are cv2's (host) in the reference
            except Exception:  # noqa: BLE001  (cv2.error: "StereoRectifyUncalibrated failed, skip!" in the reference)
                pass
    return geo


def filter_matches(pred: Dict[str, Any], ransac_method: str = "HIP_RANSAC", ransac_reproj_threshold: float = DEFAULT_RANSAC_REPROJ_THRESHOLD,
                   ransac_confidence: float = DEFAULT_RANSAC_CONFIDENCE, ransac_max_iter: int = DEFAULT_RANSAC_MAX_ITER, device="cuda") -> Dict[str, Any]:  # fmt: skip
    """imcui/ui/utils.py:459-530 for key-point matches: the homography inliers become `mmkeypoints*_orig` / `mmconf`."""

    def null(p):
        p["mmkeypoints0_orig"], p["mmkeypoints1_orig"], p["mmconf"], p["H"] = np.array([]), np.array([]), np.array([]), None
        return p

    if "mkeypoints0_orig" not in pred or "mkeypoints1_orig" not in pred:
        return null(pred)
    mkpts0, mkpts1 = pred["mkeypoints0_orig"], pred["mkeypoints1_orig"]
    if mkpts0 is None or len(mkpts0) < DEFAULT_MIN_NUM_MATCHES:
        return null(pred)
    geo = compute_geometry(pred, ransac_method, ransac_reproj_threshold, ransac_confidence, ransac_max_iter, device)
    if "Homography" in geo:
        mask = geo["mask_h"]
        pred["mmkeypoints0_orig"], pred["mmkeypoints1_orig"], pred["mmconf"] = mkpts0[mask], mkpts1[mask], pred["mconf"][mask]
        pred["H"] = np.array(geo["Homography"])
    else:
        null(pred)
    geo.pop("mask_h", None)
    geo.pop("mask_f", None)
    pred["geom_info"] = geo
    return pred


def verify_matches_batched(out: Dict[str, torch.Tensor], geometry_type: str = "Homography", reproj_threshold: float = DEFAULT_RANSAC_REPROJ_THRESHOLD,
                           confidence: float = DEFAULT_RANSAC_CONFIDENCE, max_iter: int = DEFAULT_RANSAC_MAX_ITER, seed: int = 0) -> Dict[str, torch.Tensor]:  # fmt: skip
    """The fixed-stride outputs of `SuperPointLightGluePipeline` (keypoints0/1 [B,K,2], matches0 [B,K], num_keypoints0 [B]) -> the matched
    point pairs compacted per pair on the device ([B,K,2] x 2 + counts) and their batched verification; nothing touches the host."""
    k0, k1, m0 = out["keypoints0"], out["keypoints1"], out["matches0"].long()
    B, K, _ = k0.shape
    valid = (m0 > -1) & (torch.arange(K, device=k0.device)[None] < out["num_keypoints0"].long()[:, None])
    order = torch.argsort((~valid).to(torch.int8), dim=1, stable=True)  # matched rows first, original order kept
    a = torch.gather(k0, 1, order[..., None].expand(-1, -1, 2))
    partner = torch.gather(m0.clamp(min=0), 1, order)
    b = torch.gather(k1, 1, partner[..., None].expand(-1, -1, 2))
    counts = valid.sum(1).to(torch.int32)
    res = ransac_batched(a, b, counts, geometry_type, reproj_threshold, confidence, max_iter, seed)
    res.update(mkeypoints0=a, mkeypoints1=b, num_matches=counts, order=order)
    return res
