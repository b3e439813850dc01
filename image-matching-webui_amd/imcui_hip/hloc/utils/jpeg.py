"""Baseline-JPEG decode with the pixels reconstructed on the device (include/imcui_hip.h, the JPEG section).

`read_image` of the reference (imcui/hloc/utils/io.py:11-21) is `cv2.imread`: one host decode per image.  Here the Huffman bit stream
is decoded by the library's host routine (`imcui_hip_jpeg_entropy_decode`: re-entrant C++, called through ctypes, which releases the
GIL -- `JpegDecoder` runs it on a thread pool) into quantised DCT coefficients, those go to the device, and dequantisation + inverse DCT
+ chroma up-sampling + colour conversion run there (`imcui_hip_jpeg_reconstruct`), bit-exact against libjpeg's default path (what
cv2 / PIL run).  gray=True returns the luma plane of the file, which is what `cv2.imread(IMREAD_GRAYSCALE)` hands the extractor for a JPEG.

Files the device path does not take (progressive, CMYK, 4:4:0, an EXIF orientation other than upright, non-JPEG) raise
`JpegUnsupported`; the drivers keep the host reader for those.
"""
from __future__ import annotations

import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from ... import backend
from ...lib_loader import load_library

INFO_INTS = 24


class JpegUnsupported(ValueError):
    """Not a baseline JPEG this decoder takes (the caller falls back to its host reader)."""


def jpeg_info(data: bytes) -> list[int]:
    lib = load_library()
    info = (C.c_int * INFO_INTS)()
    rc = lib.imcui_hip_jpeg_info(data, len(data), info)
    if rc != 0:
        raise JpegUnsupported(f"imcui_hip_jpeg_info: status {rc} ({'unsupported JPEG variant' if rc == -4 else 'not a JPEG / damaged header'})")
    return list(info)


def entropy_decode(data: bytes, pinned: bool = False):
    """-> (info, coefficients int16 [host tensor], quantisation tables uint16-as-int16 [3 * 64] [host tensor])."""
    lib = load_library()
    info = (C.c_int * INFO_INTS)()
    rc = lib.imcui_hip_jpeg_info(data, len(data), info)
    if rc != 0:
        raise JpegUnsupported(f"imcui_hip_jpeg_info: status {rc}")
    if info[8] != 1:
        raise JpegUnsupported(f"EXIF orientation {info[8]}: cv2.imread rotates such files; not done on the device")
    n = lib.imcui_hip_jpeg_coef_count(info)
    coef = torch.empty(n, dtype=torch.int16, pin_memory=pinned)
    qt = torch.empty(192, dtype=torch.int16, pin_memory=pinned)
    rc = lib.imcui_hip_jpeg_entropy_decode(data, len(data), coef.data_ptr(), qt.data_ptr())
    if rc != 0:
        raise JpegUnsupported(f"imcui_hip_jpeg_entropy_decode: status {rc} ({'unsupported JPEG variant' if rc == -4 else 'damaged bit stream'})")
    return info, coef, qt


def reconstruct(info, coef: torch.Tensor, qt: torch.Tensor, gray: bool, device) -> torch.Tensor:
    """Coefficients (host or device tensors) -> uint8 [H,W] (gray) or [H,W,3] (RGB) on `device`."""
    device = torch.device(device)
    hd = backend.get_handle(device)
    lib = hd.lib
    W, H = info[0], info[1]
    if gray and info[2] == 3:  # the luma plane alone: the chroma coefficients stay on the host
        ny = info[5] * info[9] * info[6] * info[10] * 64
        coef = coef[:ny]
    coef_d = coef.to(device, non_blocking=True)
    qt_d = qt.to(device, non_blocking=True)
    out = torch.empty((H, W) if gray else (H, W, 3), dtype=torch.uint8, device=device)
    nbytes = lib.imcui_hip_jpeg_workspace_bytes(info, int(gray))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        rc = lib.imcui_hip_jpeg_reconstruct(hd.h, backend._ptr(coef_d), backend._ptr(qt_d), info, int(gray), backend._ptr(out), backend._ptr(ws), nbytes,
                                            backend._stream_ptr())  # fmt: skip
        hd.check(rc, "imcui_hip_jpeg_reconstruct")
    return out


def decode_jpeg(data: bytes, gray: bool, device) -> torch.Tensor:
    info, coef, qt = entropy_decode(data)
    return reconstruct(info, coef, qt, gray, device)


class JpegDecoder:
    """Batch decode: the bit streams on `threads` host threads, the pixels on the device."""

    def __init__(self, device, threads: int = 8):
        self.device = torch.device(device)
        self.pool = ThreadPoolExecutor(max_workers=max(1, threads))

    def decode_batch(self, blobs, gray: bool):
        """blobs: list of `bytes`; -> list of uint8 device tensors (JpegUnsupported instances for the files the device path refuses)."""
        def host(b):
            try:
                return entropy_decode(b, pinned=True)
            except JpegUnsupported as e:
                return e

        staged = list(self.pool.map(host, blobs))
        return [s if isinstance(s, JpegUnsupported) else reconstruct(*s, gray, self.device) for s in staged]

    def close(self):
        self.pool.shutdown(wait=True)


def is_jpeg(data: bytes) -> bool:
    return len(data) > 3 and data[0] == 0xFF and data[1] == 0xD8
