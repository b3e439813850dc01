"""Fixtures for the JPEG decoder tests, made in the build container where /root/reference exists (the GPU box has no reference tree):

  tests/golden/jpeg_reference_files.npz -- the BYTES of six small JPEG files of the reference repository
  (imcui/datasets/sacre_coeur/mapping_scale/*.jpg, ~18 KB each) together with what PIL (libjpeg-turbo 3.x, the decoder family behind
  the reference's cv2.imread) decodes them to: RGB and gray (= libjpeg's JCS_GRAYSCALE output, the luma plane).

    python tests/golden/make_jpeg_fixtures.py
"""
import glob
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
files = sorted(glob.glob("/root/reference/imcui/datasets/sacre_coeur/mapping_scale/*.jpg"), key=os.path.getsize)[:6]
out = {}
for i, f in enumerate(files):
    data = open(f, "rb").read()
    rgb = np.array(Image.open(io.BytesIO(data)).convert("RGB"))
    im = Image.open(io.BytesIO(data))
    im.draft("L", im.size)
    assert im.mode == "L" and im.size == (rgb.shape[1], rgb.shape[0])
    out[f"name{i}"] = np.array(os.path.relpath(f, "/root/reference"))
    out[f"bytes{i}"] = np.frombuffer(data, dtype=np.uint8)
    out[f"rgb{i}"] = rgb
    out[f"gray{i}"] = np.array(im)
np.savez_compressed(os.path.join(HERE, "jpeg_reference_files.npz"), **out)
print({k: v.shape for k, v in out.items() if k.startswith("rgb")})
