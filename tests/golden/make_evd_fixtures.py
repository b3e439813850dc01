"""Decode the reference's real image pairs into committed fixtures for the GPU parity / AUC leg.

    python tests/golden/make_evd_fixtures.py        # needs /root/reference (the build container), PIL

Sources (read-only):
  * the 15 EVD pairs with ground-truth homographies:  imcui/datasets/wxbs_benchmark/.EVD/EVD/{1,2}/*.png, h/*.txt
    (x2 ~ H x1; checked by warping image 1 onto image 2)
  * the reference's own test pair:                   tests/data/*.jpg   (no ground truth)
Every image goes through the reference's host preprocessing for the `superpoint_max` / `loftr` confs
(imcui/hloc/extract_features.py:106-160, configs/extractors.py:29-45): RGB -> gray with OpenCV's 8-bit fixed point
(oracle/preprocess.py:rgb_to_gray_u8), then `force_resize` to 640 x 480 with area interpolation
(oracle/preprocess.py:area_resize_f32; images that would GROW along a side are resized with PIL bilinear, the
reference's own fallback is INTER_LINEAR).  The result is rounded to uint8 and stored PNG-compressed in
tests/golden/evd_pairs.npz together with the homographies rescaled to the 640 x 480 frames.  The fixtures are
model-boundary INPUTS (the tests divide by 255): both the HIP path and the oracle read exactly these bytes.
"""
import glob
import io
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.preprocess import area_resize_f32, rgb_to_gray_u8  # noqa: E402

REF = "/root/reference"
EVD = os.path.join(REF, "imcui/datasets/wxbs_benchmark/.EVD/EVD")
W, H = 640, 480


def load_gray(path):
    im = Image.open(path)
    if im.mode == "L":
        return np.asarray(im).astype(np.uint8)
    return rgb_to_gray_u8(np.asarray(im.convert("RGB")))


def to_model_size(gray):
    h, w = gray.shape
    if w >= W and h >= H:
        out = area_resize_f32(gray.astype(np.float32), (W, H))
    else:  # a side would grow: the reference falls back to INTER_LINEAR (extract_features.py:30-31)
        out = np.asarray(Image.fromarray(gray).resize((W, H), Image.BILINEAR), dtype=np.float32)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8), (w, h)


def png_bytes(arr):
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format="PNG", optimize=True)
    return np.frombuffer(buf.getvalue(), dtype=np.uint8)


def main():
    names, blobs0, blobs1, hs, sizes = [], [], [], [], []
    for p1 in sorted(glob.glob(os.path.join(EVD, "1", "*.png"))):
        name = os.path.splitext(os.path.basename(p1))[0]
        g0, s0 = to_model_size(load_gray(p1))
        g1, s1 = to_model_size(load_gray(os.path.join(EVD, "2", name + ".png")))
        hm = np.loadtxt(os.path.join(EVD, "h", name + ".txt"))
        a0 = np.diag([W / s0[0], H / s0[1], 1.0])
        a1 = np.diag([W / s1[0], H / s1[1], 1.0])
        hm = a1 @ hm @ np.linalg.inv(a0)
        names.append(name)
        blobs0.append(png_bytes(g0))
        blobs1.append(png_bytes(g1))
        hs.append(hm / hm[2, 2])
        sizes.append([*s0, *s1])
    jpgs = sorted(glob.glob(os.path.join(REF, "tests/data/*.jpg")))
    g0, s0 = to_model_size(load_gray(jpgs[0]))
    g1, s1 = to_model_size(load_gray(jpgs[1]))
    names.append("tests_data_jpg")
    blobs0.append(png_bytes(g0))
    blobs1.append(png_bytes(g1))
    hs.append(np.full((3, 3), np.nan))  # no ground truth for this pair
    sizes.append([*s0, *s1])
    out = os.path.join(HERE, "evd_pairs.npz")
    np.savez(out, names=np.array(names), homographies=np.stack(hs), original_sizes=np.array(sizes),
             **{f"img0_{i}": b for i, b in enumerate(blobs0)}, **{f"img1_{i}": b for i, b in enumerate(blobs1)})  # fmt: skip
    print(out, os.path.getsize(out) / 1e6, "MB", len(names), "pairs")


if __name__ == "__main__":
    main()
