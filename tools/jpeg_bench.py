"""Throughput of the JPEG path alone: N synthetic 640x480 4:2:0 quality-90 files -> luma planes on the device.
Stages timed separately: the Huffman stage on T host threads (no device work), the device reconstruction on resident coefficients
(HIP events), and the whole path (host threads -> pinned staging -> H2D -> kernels).    python tools/jpeg_bench.py [threads]"""
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-matching-webui_amd")]
import torch  # noqa: E402
from PIL import Image  # noqa: E402

from imcui_hip.hloc.utils import jpeg as J  # noqa: E402
from imcui_hip.synth import make_pair  # noqa: E402

dev = torch.device("cuda:0")
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 32
blobs = []
for i in range(16):
    g = (make_pair(300 + i, 480, 640)[0][0, 0] * 255).round().to(torch.uint8).numpy()
    buf = io.BytesIO()
    Image.fromarray(g).convert("RGB").save(buf, "JPEG", quality=90, subsampling="4:2:0")
    blobs.append(buf.getvalue())
blobs = blobs * 16  # 256 files
print(f"{len(blobs)} files, {sum(map(len, blobs)) / len(blobs) / 1e3:.1f} KB each, {threads} host threads")
import ctypes as C  # noqa: E402

from imcui_hip import load_library  # noqa: E402

lib = load_library()
n = len(blobs)
info = (C.c_int * 24)()
assert lib.imcui_hip_jpeg_info(blobs[0], len(blobs[0]), info) == 0
ny, ncb = info[5] * info[9] * info[6] * info[10] * 64, info[5] * info[13] * info[6] * info[14] * 64
stage = torch.empty(n * (ny + 2 * ncb + 192), dtype=torch.int16, pin_memory=True)
base = stage.data_ptr()
planes = (C.c_void_p * (3 * n))()
for i in range(n):
    planes[3 * i], planes[3 * i + 1], planes[3 * i + 2] = base + 2 * i * ny, base + 2 * (n * ny + i * ncb), base + 2 * (n * (ny + ncb) + i * ncb)
status = (C.c_int * n)()
data_p, sizes = (C.c_char_p * n)(*blobs), (C.c_size_t * n)(*[len(b) for b in blobs])
# 1. Huffman stage alone: the library's own threads writing into pinned memory
for t in sorted({1, 8, threads}):
    lib.imcui_hip_jpeg_entropy_decode_batch(data_p, sizes, n, planes, base + 2 * n * (ny + 2 * ncb), status, t)
    t0 = time.perf_counter()
    lib.imcui_hip_jpeg_entropy_decode_batch(data_p, sizes, n, planes, base + 2 * n * (ny + 2 * ncb), status, t)
    dt = time.perf_counter() - t0
    assert all(v == 0 for v in status)
    print(f"entropy decode, {t:3d} host thread(s): {n / dt:8.0f} images/s ({dt / n * 1e3 * t:.2f} ms per image and thread, {sum(map(len, blobs)) / dt / 1e6:.0f} MB/s of bit stream)")
# 2. device reconstruction alone (coefficients of the whole batch resident)
hd = backend = None
from imcui_hip import backend  # noqa: E402

hd = backend.get_handle(dev)
coef_d, qt_d = stage[: n * (ny + 2 * ncb)].to(dev), stage[n * (ny + 2 * ncb) :].to(dev)
for gray in (True, False):
    out = torch.empty((n, 480, 640) if gray else (n, 480, 640, 3), dtype=torch.uint8, device=dev)
    nbytes = lib.imcui_hip_jpeg_workspace_bytes_batch(info, int(gray), n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    cy = coef_d.data_ptr()

    def run():
        hd.check(lib.imcui_hip_jpeg_reconstruct_batch(hd.h, cy, cy + 2 * n * ny, cy + 2 * n * (ny + ncb), backend._ptr(qt_d), info, n, int(gray), backend._ptr(out),
                                                      backend._ptr(ws), nbytes, backend._stream_ptr()), "reconstruct")  # fmt: skip

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byt = n * (ny * 2 + 480 * 640) if gray else n * ((ny + 2 * ncb) * 2 + 480 * 640 * 3)
    print(f"device reconstruction of {n} files in one call ({'luma only' if gray else 'RGB: 3 IDCT planes + up-sampling + colour'}): {n / ms * 1e3:8.0f} images/s "
          f"({ms / n * 1e3:.2f} us per image, {byt / ms / 1e6:.0f} GB/s of coefficients read + pixels written)")
# 3. the whole path: bytes -> device tensors
dec = J.JpegDecoder(dev, threads=threads)
for gray in (True, False):
    dec.decode_batch(blobs, gray)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = dec.decode_batch(blobs, gray)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"files -> {'gray' if gray else 'RGB '} on the device, whole path ({threads} host threads, pinned staging, one transfer + three launches per batch): {n / dt:8.0f} images/s")
dec.close()
