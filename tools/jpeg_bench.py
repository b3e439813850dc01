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
dec = J.JpegDecoder(dev, threads=threads)
# 1. Huffman stage alone
for rep in range(2):
    t0 = time.perf_counter()
    staged = list(dec.pool.map(lambda b: J.entropy_decode(b, pinned=False), blobs))
    dt = time.perf_counter() - t0
print(f"entropy decode on {threads} threads: {len(blobs) / dt:8.0f} images/s ({dt / len(blobs) * 1e3 * threads:.2f} ms per image and thread)")
# 2. device reconstruction alone (coefficients resident)
res = [(info, c.to(dev), q.to(dev)) for info, c, q in staged[:64]]
for gray in (True, False):
    for _ in range(2):
        for info, c, q in res:
            J.reconstruct(info, c, q, gray, dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for info, c, q in res:
        J.reconstruct(info, c, q, gray, dev)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"device reconstruction ({'luma only' if gray else 'RGB: 3 IDCT planes + up-sampling + colour'}): {len(res) / ms * 1e3:8.0f} images/s ({ms / len(res) * 1e3:.1f} us per image, "
          f"launch-bound: one image per call)")
# 3. the whole path
for gray in (True, False):
    dec.decode_batch(blobs[:32], gray)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = dec.decode_batch(blobs, gray)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"files -> {'gray' if gray else 'RGB '} on the device, whole path: {len(blobs) / dt:8.0f} images/s")
dec.close()
