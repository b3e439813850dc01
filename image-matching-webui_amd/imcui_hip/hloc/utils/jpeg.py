"""Baseline-JPEG decode with the pixels reconstructed on the device (include/imcui_hip.h, the JPEG section).

`read_image` of the reference (imcui/hloc/utils/io.py:11-21) is `cv2.imread`: one host decode per image.  Here the Huffman bit stream
is decoded by the library's host routine (`imcui_hip_jpeg_entropy_decode`: re-entrant C++, called through ctypes, which releases the
GIL; `JpegDecoder` hands whole batches to the library's own thread pool, `imcui_hip_jpeg_entropy_decode_batch`) into quantised DCT coefficients, those go to the device, and dequantisation + inverse DCT
+ chroma up-sampling + colour conversion run there (`imcui_hip_jpeg_reconstruct`), bit-exact against libjpeg's default path (what
cv2 / PIL run).  gray=True returns the luma plane of the file, which is what `cv2.imread(IMREAD_GRAYSCALE)` hands the extractor for a JPEG.

An EXIF orientation is applied on the device after the reconstruction (`imcui_hip_orient_u8`), as cv2.imread does inside its decoder.

Files the device path does not take (arithmetic-coded, CMYK, 4:4:0, non-JPEG; progressive Huffman files ARE taken since round 5) raise
`JpegUnsupported`; the drivers keep the host reader for those.
"""
from __future__ import annotations

import ctypes as C
import numpy as np
import torch

from ... import backend
from ...lib_loader import load_library

INFO_INTS = 24
MAX_PIXELS = 1 << 27  # 128 Mpx: a size field beyond this is a damaged or hostile header, not a photograph (the staging buffer is sized by it)


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
    if info[0] * info[1] > MAX_PIXELS:
        raise JpegUnsupported(f"{info[0]} x {info[1]} pixels: beyond MAX_PIXELS")
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


def apply_orientation(img: torch.Tensor, orientation: int) -> torch.Tensor:
    """EXIF orientation 2..8 on a decoded uint8 device image [H,W] / [H,W,C], as cv2.imread / PIL's exif_transpose apply it."""
    if orientation == 1:
        return img
    hd = backend.get_handle(img.device)
    H, W = img.shape[:2]
    Cc = img.shape[2] if img.dim() == 3 else 1
    shape = ((W, H) if orientation >= 5 else (H, W)) + ((Cc,) if img.dim() == 3 else ())
    out = torch.empty(shape, dtype=torch.uint8, device=img.device)
    with torch.cuda.device(img.device):
        hd.check(hd.lib.imcui_hip_orient_u8(hd.h, backend._ptr(img.contiguous()), H, W, Cc, int(orientation), backend._ptr(out), backend._stream_ptr()), "imcui_hip_orient_u8")
    return out


def decode_jpeg(data: bytes, gray: bool, device) -> torch.Tensor:
    info, coef, qt = entropy_decode(data)
    return apply_orientation(reconstruct(info, coef, qt, gray, device), info[8])


class JpegDecoder:
    """Batch decode: the bit streams on `threads` host threads OF THE LIBRARY (`imcui_hip_jpeg_entropy_decode_batch`: no interpreter
    lock, no per-file allocation) straight into a pinned staging buffer laid out plane-major per geometry group, ONE transfer and three
    kernel launches per group of equally shaped files (`imcui_hip_jpeg_reconstruct_batch`).  Two staging buffers alternate; a buffer is
    rewritten only after the transfer that last read it has completed."""

    def __init__(self, device, threads: int = 8):
        self.device = torch.device(device)
        self.threads = max(1, int(threads))
        self._stage = [None, None]
        self._done = [None, None]
        self._turn = 0

    def _staging(self, shorts: int) -> torch.Tensor:
        k = self._turn & 1
        self._turn += 1
        if self._done[k] is not None:
            self._done[k].synchronize()  # the transfer that last read this buffer
        if self._stage[k] is None or self._stage[k].numel() < shorts:
            self._stage[k] = torch.empty(max(shorts, 1 << 20), dtype=torch.int16, pin_memory=True)
        return self._stage[k], k

    def decode_batch(self, blobs, gray: bool):
        """blobs: list of `bytes`; -> list of uint8 device tensors ([H,W] or [H,W,3]; views into one tensor per geometry group), with a
        `JpegUnsupported` instance in the place of every file the device path refuses."""
        lib = load_library()
        dev = self.device
        n = len(blobs)
        results: list = [None] * n
        infos: list = [None] * n
        info_of: dict = {}  # file -> its EXIF orientation (the geometry key leaves it out: rotated files batch with upright ones)
        groups: dict = {}
        for i, b in enumerate(blobs):
            info = (C.c_int * INFO_INTS)()
            rc = lib.imcui_hip_jpeg_info(b, len(b), info)
            if rc != 0:
                results[i] = JpegUnsupported(f"imcui_hip_jpeg_info: status {rc}")
            elif info[0] * info[1] > MAX_PIXELS:
                results[i] = JpegUnsupported(f"{info[0]} x {info[1]} pixels: beyond MAX_PIXELS")
            else:
                infos[i] = info
                info_of[i] = info[8]
                groups.setdefault(tuple(info[:8]) + tuple(info[9:21]), []).append(i)
        if not groups:
            return results
        # staging layout (int16 elements): per group [Y planes of its files | Cb planes | Cr planes], then the tables of all files
        plan, off = [], 0
        for key, idx in groups.items():
            info = infos[idx[0]]
            ny = info[5] * info[9] * info[6] * info[10] * 64
            ncb = info[5] * info[13] * info[6] * info[14] * 64 if info[2] == 3 else 0
            plan.append((idx, info, ny, ncb, off))
            off += len(idx) * (ny + 2 * ncb)
        qt_off = off
        total = off + n * 192
        stage, slot = self._staging(total)
        base = stage.data_ptr()
        data_p = (C.c_char_p * n)(*[b if infos[i] is not None else None for i, b in enumerate(blobs)])
        sizes = (C.c_size_t * n)(*[len(b) for b in blobs])
        planes = (C.c_void_p * (3 * n))()
        for idx, info, ny, ncb, goff in plan:
            m = len(idx)
            for k, i in enumerate(idx):
                planes[3 * i] = base + 2 * (goff + k * ny)
                if ncb:
                    planes[3 * i + 1] = base + 2 * (goff + m * ny + k * ncb)
                    planes[3 * i + 2] = base + 2 * (goff + m * (ny + ncb) + k * ncb)
        status = (C.c_int * n)()
        rc = lib.imcui_hip_jpeg_entropy_decode_batch(data_p, sizes, n, planes, base + 2 * qt_off, status, self.threads)
        if rc != 0:
            raise backend.ImcuiHipError(f"imcui_hip_jpeg_entropy_decode_batch failed ({rc})")
        hd = backend.get_handle(dev)
        qt_d = stage[qt_off:total].to(dev, non_blocking=True)
        for idx, info, ny, ncb, goff in plan:
            m = len(idx)
            chroma = info[2] == 3 and not gray
            span = m * (ny + (2 * ncb if chroma else 0))
            coef_d = stage[goff : goff + span].to(dev, non_blocking=True)
            W, H = info[0], info[1]
            out = torch.empty((m, H, W) if gray else (m, H, W, 3), dtype=torch.uint8, device=dev)
            # the tables of this group's files, in group order
            qsel = qt_d.view(n, 192)[torch.tensor(idx, device=dev)] if m != n else qt_d.view(n, 192)
            nbytes = lib.imcui_hip_jpeg_workspace_bytes_batch(info, int(gray), m)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            cy = coef_d.data_ptr()
            ccb = cy + 2 * m * ny if chroma else None
            ccr = cy + 2 * m * (ny + ncb) if chroma else None
            with torch.cuda.device(dev):
                rc = lib.imcui_hip_jpeg_reconstruct_batch(hd.h, cy, ccb, ccr, backend._ptr(qsel.contiguous()), info, m, int(gray), backend._ptr(out), backend._ptr(ws),
                                                          nbytes, backend._stream_ptr())  # fmt: skip
                hd.check(rc, "imcui_hip_jpeg_reconstruct_batch")
            for k, i in enumerate(idx):
                results[i] = apply_orientation(out[k], info_of[i]) if status[i] == 0 else JpegUnsupported(
                    f"imcui_hip_jpeg_entropy_decode: status {status[i]} ({'unsupported JPEG variant' if status[i] == -4 else 'damaged bit stream'})")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._done[slot] = ev
        return results

    def close(self):
        for ev in self._done:
            if ev is not None:
                ev.synchronize()
        self._stage = [None, None]


def is_jpeg(data: bytes) -> bool:
    return len(data) > 3 and data[0] == 0xFF and data[1] == 0xD8
