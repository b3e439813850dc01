"""Fixtures for the PNG decoder tests, made in the build container where /root/reference exists (the GPU box has no reference tree):

  tests/golden/png_reference_files.npz -- the BYTES of six PNG files of the reference repository (the five smallest of
  imcui/datasets/wxbs_benchmark/.WxBS/v1.1 -- 8-bit gray -- and .EVD/EVD/1/mag.png -- 8-bit RGB) together with what PIL decodes them to.

    python tests/golden/make_png_fixtures.py
"""
import io
import os
import subprocess

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
files = sorted(subprocess.check_output(["find", "/root/reference", "-iname", "*.png"]).decode().split(), key=os.path.getsize)
rgb = [f for f in files if Image.open(f).mode == "RGB"][:1]
gray = [f for f in files if Image.open(f).mode == "L"][:4]
out = {}
for i, f in enumerate(gray + rgb):
    data = open(f, "rb").read()
    im = Image.open(io.BytesIO(data))
    out[f"name{i}"] = np.array(os.path.relpath(f, "/root/reference"))
    out[f"bytes{i}"] = np.frombuffer(data, dtype=np.uint8)
    out[f"pixels{i}"] = np.array(im if im.mode == "L" else im.convert("RGB"))
np.savez_compressed(os.path.join(HERE, "png_reference_files.npz"), **out)
print({k: v.shape for k, v in out.items() if k.startswith("pixels")})
