"""PNG decode on the device (csrc/png.hip through the C ABI) against PIL, bit for bit (GPU box only): files of the reference repository
(committed fixtures), PIL-encoded files of every colour type the path takes, an image taller than a workgroup (two wavefront strips), mixed
batches with a damaged file, and the drivers' `read_image_device` / `read_images_device` against the host reader."""
import numpy as np
import pytest
import torch

from oracle.png import libpng_rgb_to_gray
from test_png_cpu import encode, pil_pixels, reference_files, synthetic_files, texture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_reference_files_bit_exact():
    from imcui_hip.hloc.utils.png import PngDecoder

    files = reference_files()
    dec = PngDecoder(DEV, threads=4)
    outs = dec.decode_batch([d for _, d, _ in files])
    for (name, _, want), got in zip(files, outs):
        assert isinstance(got, torch.Tensor), (name, got)
        assert got.dtype == torch.uint8 and tuple(got.shape) == want.shape, name
        assert np.array_equal(got.cpu().numpy(), want), name
    dec.close()


def test_every_colour_type_and_strips_bit_exact():
    from imcui_hip.hloc.utils.png import PngDecoder, decode_png

    files = synthetic_files() + [("tall_rgba", encode(texture(31, 2100, 9, 4), "RGBA")), ("wide_gray", encode(texture(32, 3, 4000, 1)[:, :, 0], "L")),
                                 ("big", encode(texture(33, 1200, 1600, 3), "RGB"))]  # fmt: skip
    dec = PngDecoder(DEV, threads=8)
    for rounds in range(2):  # (the second round re-uses a staging buffer)
        outs = dec.decode_batch([d for _, d in files])
        for (name, data), got in zip(files, outs):
            assert isinstance(got, torch.Tensor), (name, got)
            assert np.array_equal(got.cpu().numpy(), pil_pixels(data)), name
    dec.close()
    for name, data in files[:3]:
        assert np.array_equal(decode_png(data, DEV).cpu().numpy(), pil_pixels(data)), name


def test_more_files_than_one_launch_and_a_damaged_one():
    from imcui_hip.hloc.utils.png import PngDecoder, PngUnsupported

    blobs = [encode(texture(100 + i, 20 + 3 * i, 30 + 5 * (i % 7), 3 if i % 3 else 1)[..., 0] if i % 3 == 0 else texture(100 + i, 20 + 3 * i, 30 + 5 * (i % 7), 3),
                    "L" if i % 3 == 0 else "RGB") for i in range(53)]  # fmt: skip
    bad = bytearray(blobs[17])
    bad[len(bad) // 2] ^= 0x55
    blobs[17] = bytes(bad)
    blobs[30] = b"\x89PNG\r\n\x1a\n" + b"\0" * 40
    dec = PngDecoder(DEV, threads=4)
    outs = dec.decode_batch(blobs)
    dec.close()
    for i, (b, got) in enumerate(zip(blobs, outs)):
        if i in (17, 30):
            assert isinstance(got, PngUnsupported), i
        else:
            assert np.array_equal(got.cpu().numpy(), pil_pixels(b)), i


def test_drivers_read_png_files_on_the_device(tmp_path):
    """`read_image_device` / `read_images_device` with PNG files equal the host reader `read_image_u8`, gray and colour requests, and
    `decode='device'` refuses what the device path does not take."""
    from PIL import Image

    from imcui_hip.hloc.extract_features import read_image_device, read_image_u8, read_images_device

    paths = []
    for i, (name, data) in enumerate(synthetic_files()[:7]):
        p = tmp_path / f"{name}.png"
        p.write_bytes(data)
        paths.append(p)
    jp = tmp_path / "photo.jpg"
    Image.fromarray(texture(40, 64, 80, 3)).save(jp, "JPEG", quality=90)
    paths.append(jp)
    for gray in (True, False):
        batch = read_images_device(paths, gray, torch.device(DEV), decode="auto")
        for p, got in zip(paths, batch):
            want = read_image_u8(p, gray)
            if p.suffix == ".png":
                if gray:  # cv2.imread(IMREAD_GRAYSCALE): gray files as stored, colour files through libpng's rgb_to_gray (oracle/png.py)
                    rgb = read_image_u8(p, False)
                    assert want.ndim == 2 and np.array_equal(want, libpng_rgb_to_gray(rgb)), p.name
                assert np.array_equal(got.cpu().numpy(), want), (p.name, gray)
                assert np.array_equal(read_image_device(p, gray, torch.device(DEV), decode="device").cpu().numpy(), want), (p.name, gray)
            else:
                assert got.shape[:2] == want.shape[:2]
    i16 = tmp_path / "deep.png"
    Image.fromarray(texture(41, 20, 20, 1)[:, :, 0].astype(np.uint16) << 8).save(i16, "PNG")
    assert np.array_equal(read_image_device(i16, True, torch.device(DEV)).cpu().numpy(), read_image_u8(i16, True))  # (auto: the host reader takes it)
    with pytest.raises(ValueError):
        read_image_device(i16, True, torch.device(DEV), decode="device")
