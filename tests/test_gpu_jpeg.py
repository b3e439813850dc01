"""JPEG decode with the pixels reconstructed on the device (csrc/jpeg.hip through the C ABI) against PIL = libjpeg-turbo, the decoder
family behind the reference's `cv2.imread` (imcui/hloc/utils/io.py:11-21): BIT-EXACT, RGB and gray (GPU box only)."""
import io
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from test_jpeg_cpu import CASES, GOLD, encode, exif_jpeg, pil_decode, smooth_image

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("w,h,sub,q,rst", CASES + [(640, 480, "4:2:0", 90, 0), (641, 479, "4:2:2", 85, 7), (1024, 1024, "4:4:4", 95, 0)])
def test_device_decode_equals_pil(w, h, sub, q, rst):
    from imcui_hip.hloc.utils.jpeg import decode_jpeg

    kw = dict(quality=q, subsampling=sub)
    if rst:
        kw["restart_marker_blocks"] = rst
    data = encode(smooth_image(w * 100 + h, h, w), **kw)
    for gray in (False, True):
        got = decode_jpeg(data, gray, DEV).cpu().numpy()
        want = pil_decode(data, gray)
        assert got.shape == want.shape and np.array_equal(got, want), (gray, int(np.abs(got.astype(int) - want.astype(int)).max()))


def test_device_decode_of_gray_files_and_optimised_tables():
    from imcui_hip.hloc.utils.jpeg import decode_jpeg

    for data in (encode(smooth_image(5, 45, 70, 1), quality=80), encode(smooth_image(6, 64, 64), quality=95, optimize=True, subsampling="4:2:0")):
        for gray in (False, True):
            assert np.array_equal(decode_jpeg(data, gray, DEV).cpu().numpy(), pil_decode(data, gray))


def test_device_decode_of_reference_repository_files():
    """Six JPEG files of the reference repository: bytes and PIL's decode as committed by tests/golden/make_jpeg_fixtures.py."""
    from imcui_hip.hloc.utils.jpeg import JpegDecoder

    z = np.load(GOLD)
    dec = JpegDecoder(DEV, threads=4)
    blobs = [z[f"bytes{i}"].tobytes() for i in range(6)]
    for gray, key in ((False, "rgb"), (True, "gray")):
        outs = dec.decode_batch(blobs, gray)
        for i, o in enumerate(outs):
            assert np.array_equal(o.cpu().numpy(), z[f"{key}{i}"]), (str(z[f"name{i}"]), key)
    dec.close()


def test_batch_of_640x480_files_twice_bitwise_and_refusals(tmp_path):
    """32 files through the thread pool, decoded twice: identical; unsupported files come back as JpegUnsupported (the drivers then
    read them on the host), and `read_image_device` does that fall-back."""
    from imcui_hip.hloc.extract_features import read_image_device
    from imcui_hip.hloc.utils.jpeg import JpegDecoder, JpegUnsupported

    blobs = [encode(smooth_image(100 + i, 480, 640), quality=70 + i % 25, subsampling=("4:2:0", "4:2:2", "4:4:4")[i % 3]) for i in range(32)]
    # (progressive Huffman files decode on the device since round 5; a frame announced as arithmetic-coded stands for the refused kind)
    prog = encode(smooth_image(1, 40, 40), quality=80, progressive=True).replace(b"\xff\xc2", b"\xff\xca", 1)
    dec = JpegDecoder(DEV, threads=8)
    a = dec.decode_batch(blobs + [prog], True)
    b = dec.decode_batch(blobs + [prog], True)
    dec.close()
    assert isinstance(a[-1], JpegUnsupported) and isinstance(b[-1], JpegUnsupported)
    for i, (x, y) in enumerate(zip(a[:-1], b[:-1])):
        assert torch.equal(x, y) and np.array_equal(x.cpu().numpy(), pil_decode(blobs[i], True))
    # the drivers' reader: JPEG (baseline and progressive) and PNG on the device, a CMYK JPEG through the host reader
    (tmp_path / "a.jpg").write_bytes(blobs[0])
    from PIL import Image

    cmyk = io.BytesIO()
    Image.fromarray(smooth_image(2, 40, 40)).convert("CMYK").save(cmyk, "JPEG")
    (tmp_path / "p.jpg").write_bytes(cmyk.getvalue())
    Image.fromarray(smooth_image(9, 30, 50)).save(tmp_path / "c.png")
    assert np.array_equal(read_image_device(tmp_path / "a.jpg", True, DEV).cpu().numpy(), pil_decode(blobs[0], True))
    assert read_image_device(tmp_path / "p.jpg", False, DEV).shape == (40, 40, 3)
    assert read_image_device(tmp_path / "c.png", False, DEV).shape == (30, 50, 3)
    with pytest.raises(JpegUnsupported):
        read_image_device(tmp_path / "p.jpg", True, DEV, decode="device")


@pytest.mark.parametrize("w,h,sub,q,rst", [(64, 64, "4:2:0", 80, 0), (123, 77, "4:2:2", 75, 0), (200, 150, "4:2:0", 60, 3), (640, 480, "4:2:0", 90, 0), (641, 479, "4:4:4", 92, 5)])
def test_device_decode_of_progressive_files_equals_pil(w, h, sub, q, rst):
    """Progressive Huffman files (SOF2, libjpeg's standard ten-scan script with spectral selection and successive approximation; round 5): the host
    stage accumulates the scans into the coefficient planes of the equivalent sequential file, the device reconstruction is the same -- bit-exact
    against PIL, RGB and gray, alone and in a batch next to baseline files."""
    from imcui_hip.hloc.utils.jpeg import JpegDecoder, decode_jpeg

    kw = dict(quality=q, subsampling=sub, progressive=True)
    if rst:
        kw["restart_marker_blocks"] = rst
    data = encode(smooth_image(w + 3 * h, h, w), **kw)
    assert b"\xff\xc2" in data[:1200]
    for gray in (False, True):
        assert np.array_equal(decode_jpeg(data, gray, DEV).cpu().numpy(), pil_decode(data, gray)), gray
    base = encode(smooth_image(w + 3 * h, h, w), quality=q, subsampling=sub)
    dec = JpegDecoder(DEV, threads=4)
    outs = dec.decode_batch([base, data, base, data], False)
    dec.close()
    for b, o in zip([base, data, base, data], outs):
        assert np.array_equal(o.cpu().numpy(), pil_decode(b, False))


def test_extract_features_from_jpeg_files_equals_the_plugin_on_pil_gray(tmp_path):
    """`extract_features.main` on JPEG files (decode = "auto": pixels reconstructed on the device) writes the same key-points as the
    SuperPoint plugin called on PIL's gray decode of each file -- the reference's `cv2.imread(IMREAD_GRAYSCALE)` input."""
    from imcui_hip.hloc import extract_features as ef
    from imcui_hip.hloc.extractors.superpoint import SuperPoint
    from imcui_hip.hloc.utils.h5lite import open_h5
    from imcui_hip.synth import make_pair
    from imcui_hip.synth_weights import superpoint_state_dict

    names = []
    for i in range(3):
        img = (make_pair(50 + i, 240, 320, n_blobs=400)[0][0, 0] * 255).round().to(torch.uint8).numpy()
        rgb = np.stack([img, np.roll(img, 3, 0), np.roll(img, 5, 1)], -1)
        (tmp_path / f"im{i}.jpg").write_bytes(encode(rgb, quality=92, subsampling="4:2:0"))
        names.append(f"im{i}.jpg")
    mconf = {"name": "superpoint", "nms_radius": 3, "max_keypoints": 512, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": superpoint_state_dict(0)}
    model = SuperPoint(dict(mconf)).eval().to(DEV)
    conf = {"output": "feats", "model": mconf, "preprocessing": {"grayscale": True, "resize_max": None}}
    path = ef.main(conf, tmp_path, tmp_path, as_half=False, image_list=names, model=model, batch_size=2)
    with open_h5(path, "r") as fd:
        for n in names:
            gray = pil_decode((tmp_path / n).read_bytes(), True)
            # (the driver's own uint8 -> float32 step, so that both sides feed the extractor the same bits)
            ref = model({"image": ef.preprocess_on_device(gray, SimpleNamespace(grayscale=True, resize_max=None, force_resize=False, interpolation="cv2_area"), torch.device(DEV))})
            k = np.asarray(fd[n]["keypoints"])
            assert k.shape[0] > 100 and np.array_equal(k, ref["keypoints"][0].cpu().numpy())
            assert np.array_equal(np.asarray(fd[n]["scores"]), ref["scores"][0].cpu().numpy())


def test_exif_orientation_is_applied_on_the_device():
    """cv2.imread applies the EXIF orientation inside its decoder; the device path does it after the reconstruction
    (`imcui_hip_orient_u8`): all eight values, RGB and gray, single-file and batch entry points, against PIL's exif_transpose."""
    from PIL import Image, ImageOps

    from imcui_hip.hloc.utils.jpeg import JpegDecoder, decode_jpeg

    blobs = [exif_jpeg(smooth_image(40 + o, 37, 53), o, quality=88, subsampling="4:2:0") for o in range(1, 9)]
    dec = JpegDecoder(DEV, threads=2)
    for gray in (False, True):
        batch = dec.decode_batch(blobs, gray)
        for o, data in enumerate(blobs, 1):
            im = ImageOps.exif_transpose(Image.open(io.BytesIO(data)))
            want = np.array(im.convert("RGB"))
            if gray:  # the luma plane of the file in its EXIF orientation
                from oracle.jpeg import orient

                want = np.ascontiguousarray(orient(pil_decode(data, True), o))
            assert np.array_equal(decode_jpeg(data, gray, DEV).cpu().numpy(), want), (o, gray)
            assert np.array_equal(batch[o - 1].cpu().numpy(), want), (o, gray)
    dec.close()
