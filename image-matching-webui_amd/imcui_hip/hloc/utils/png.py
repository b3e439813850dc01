"""PNG files -> uint8 device tensors through the library (csrc/png.hip): chunk walk + zlib inflate on the library's host threads (C ABI, no
interpreter lock) into pinned staging, ONE transfer, the scan-line filters undone on the device as a per-image wavefront.

Replaces, for the batch drivers, the PNG half of `read_image` = `cv2.imread` (imcui/hloc/utils/io.py:11-21): the reference's own evaluation
images (imcui/datasets/wxbs_benchmark/**.png) are PNG.  Output = what `extract_features.read_image_u8` returns for the file: [H,W] for gray
(and gray + alpha) files, [H,W,3] RGB for RGB / RGBA / palette files (alpha dropped).  Bit-exact; the checker is PIL.  Interlaced files,
bit depths other than 8 and files whose eXIf chunk asks for a rotation / mirror (cv2.imread and PIL apply it on load) raise `PngUnsupported`
(the caller keeps its host reader)."""
from __future__ import annotations

import ctypes as C

import torch

from ... import backend
from ...lib_loader import load_library

INFO_INTS = 8
MAX_PIXELS = 1 << 28


class PngUnsupported(ValueError):
    """The device path does not take this file (interlaced, 16-bit, damaged ...): use the host reader."""


def is_png(data: bytes) -> bool:
    return len(data) > 8 and data[:8] == b"\x89PNG\r\n\x1a\n"


def png_info(data: bytes):
    info = (C.c_int * INFO_INTS)()
    rc = load_library().imcui_hip_png_info(data, len(data), info)
    if rc != 0:
        raise PngUnsupported(f"imcui_hip_png_info: status {rc}")
    return info


class PngDecoder:
    """`decode_batch(blobs)`: every file inflated on `threads` host threads straight into one pinned staging buffer, one transfer, one or a few
    launches (24 images each).  Two staging buffers alternate; a buffer is rewritten only after the transfer that last read it completed."""

    def __init__(self, device, threads: int = 8):
        self.device = torch.device(device)
        self.threads = max(1, int(threads))
        self._stage = [None, None]
        self._done = [None, None]
        self._turn = 0

    def _staging(self, nbytes: int):
        k = self._turn & 1
        self._turn += 1
        if self._done[k] is not None:
            self._done[k].synchronize()
        if self._stage[k] is None or self._stage[k].numel() < nbytes:
            self._stage[k] = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8, pin_memory=True)
        return self._stage[k], k

    def decode_batch(self, blobs):
        """blobs: list of `bytes` -> list of uint8 device tensors ([H,W] or [H,W,3]), a `PngUnsupported` in the place of every refused file."""
        lib = load_library()
        dev = self.device
        n = len(blobs)
        results: list = [None] * n
        infos: list = [None] * n
        for i, b in enumerate(blobs):
            info = (C.c_int * INFO_INTS)()
            rc = lib.imcui_hip_png_info(b, len(b), info)
            if rc != 0:
                results[i] = PngUnsupported(f"imcui_hip_png_info: status {rc}")
            elif info[0] * info[1] > MAX_PIXELS:
                results[i] = PngUnsupported(f"{info[0]} x {info[1]} pixels: beyond MAX_PIXELS")
            else:
                infos[i] = info
        live = [i for i in range(n) if infos[i] is not None]
        if not live:
            return results
        m = len(live)
        raw_sizes = [lib.imcui_hip_png_raw_bytes(infos[i]) for i in live]
        raw_off, off = [], 0
        for s in raw_sizes:
            raw_off.append(off)
            off += (s + 255) // 256 * 256
        pal_off = off
        total = off + 768 * m
        stage, slot = self._staging(total)
        base = stage.data_ptr()
        data_p = (C.c_char_p * m)(*[blobs[i] for i in live])
        sizes = (C.c_size_t * m)(*[len(blobs[i]) for i in live])
        raws = (C.c_void_p * m)(*[base + o for o in raw_off])
        rbytes = (C.c_size_t * m)(*raw_sizes)
        status = (C.c_int * m)()
        rc = lib.imcui_hip_png_inflate_batch(data_p, sizes, m, raws, rbytes, base + pal_off, status, self.threads)
        if rc != 0:
            raise backend.ImcuiHipError(f"imcui_hip_png_inflate_batch failed ({rc})")
        for k, i in enumerate(live):
            if status[k] != 0:
                results[i] = PngUnsupported(f"imcui_hip_png_inflate: status {status[k]} (damaged stream)")
        good = [k for k in range(m) if status[k] == 0]  # a refused file's staging is uninitialised: it gets no device job
        if not good:
            return results
        g = len(good)
        hd = backend.get_handle(dev)
        staged = stage[:total].to(dev, non_blocking=True)
        out_sizes = [infos[live[k]][0] * infos[live[k]][1] * infos[live[k]][4] for k in good]
        out_off, off = [], 0
        for s in out_sizes:
            out_off.append(off)
            off += (s + 255) // 256 * 256
        out = torch.empty(max(off, 1), dtype=torch.uint8, device=dev)
        info_flat = (C.c_int * (INFO_INTS * g))()
        for q, k in enumerate(good):
            for w in range(INFO_INTS):
                info_flat[q * INFO_INTS + w] = infos[live[k]][w]
        nbytes = lib.imcui_hip_png_workspace_bytes(info_flat, g)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        # palettes are indexed by job: compact the live files' palette slots the same way the jobs are
        pal_ptr = staged.data_ptr() + pal_off
        if g != m:
            pal = staged[pal_off : pal_off + 768 * m].view(m, 768)[torch.tensor(good, device=dev)].contiguous()
            pal_ptr = pal.data_ptr()
        with torch.cuda.device(dev):
            rc = lib.imcui_hip_png_reconstruct_batch(hd.h, backend._ptr(staged), (C.c_size_t * g)(*[raw_off[k] for k in good]), info_flat, pal_ptr, g, backend._ptr(out),
                                                     (C.c_size_t * g)(*out_off), backend._ptr(ws), nbytes, backend._stream_ptr())  # fmt: skip
            hd.check(rc, "imcui_hip_png_reconstruct_batch")
        for q, k in enumerate(good):
            i = live[k]
            W, H, ch = infos[i][0], infos[i][1], infos[i][4]
            t = out[out_off[q] : out_off[q] + out_sizes[q]]
            results[i] = t.view(H, W) if ch == 1 else t.view(H, W, 3)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._done[slot] = ev
        return results

    def close(self):
        for ev in self._done:
            if ev is not None:
                ev.synchronize()
        self._stage = [None, None]


def decode_png(data: bytes, device) -> torch.Tensor:
    """One file (convenience; the drivers use `PngDecoder.decode_batch`)."""
    dec = PngDecoder(device, threads=1)
    r = dec.decode_batch([data])[0]
    dec.close()
    if isinstance(r, PngUnsupported):
        raise r
    return r
