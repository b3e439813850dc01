"""Batched feature-extraction driver on the HIP plugins (SURVEY.md section 8f rank 2 + 3).

What `imcui/hloc/extract_features.py:174-248` (`main`) does one image per `model(data)` call, done here with B images
per `forward_batched` call and the preprocessing on the device:

  * image list / skipping of already exported names: `ImageDataset.__init__` (:53-77) and `list_h5_names`
    (utils/io.py:24-37) semantics;
  * preprocessing (`ImageDataset.__getitem__` :79-103): decode (baseline JPEG: Huffman on the host, pixels on the device, bit-exact
    against libjpeg -- `read_image_device`, utils/jpeg.py; other formats: host, PIL or cv2), gray conversion, optional
    `resize_max` / `force_resize` with "cv2_area" interpolation, `/ 255` -- the arithmetic runs on the GPU
    (`backend.preprocess_area`: cv2's 8-bit RGB2GRAY fixed point + INTER_AREA decimation + float32 division);
  * images of equal preprocessed size are batched; key-points are mapped back to the original resolution with
    `(k + 0.5) * scales - 0.5` (:212-215), `uncertainty = detection_noise * scales.mean()` (:219), float32 -> float16
    when `as_half` (:221-225);
  * one HDF5 group per image with a dataset per tensor + `image_size`, `keypoints.attrs["uncertainty"]` (:227-235),
    through `utils.h5lite.open_h5` (h5py when installed, the HDF5 C library otherwise).

A resize that GROWS a side runs cv2.INTER_LINEAR like the reference (`resize_image` :29-31), also on the device
(`backend.preprocess_linear`); `superpoint_max` (force_resize to 640 x 480) therefore accepts images of any size.

Gray of a colour FILE (`cv2.imread(IMREAD_GRAYSCALE)` converts inside the decoder): JPEG = the file's luma plane (device decoder) and
PNG = libpng's `(9797 R + 19234 G + 3737 B) >> 15` (`_png_as_read_image`); a host reader without cv2 (PIL) hands back RGB and the device
kernel applies `cvtColor`'s fixed-point formula, as the UI path (`extract`, :158-161) does with its RGB input.
"""
from __future__ import annotations

from pathlib import Path
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from .. import backend
from .utils.h5lite import open_h5

GLOBS = ["*.jpg", "*.png", "*.jpeg", "*.JPG", "*.PNG"]  # ImageDataset.default_conf["globs"]
DEFAULT_PREPROCESSING = {"globs": GLOBS, "grayscale": False, "resize_max": None, "force_resize": False, "interpolation": "cv2_area"}


def list_h5_names(path) -> List[str]:
    """Names of the groups that hold datasets (imcui/hloc/utils/io.py:24-37)."""
    names = []
    with open_h5(path, "r") as fd:

        def visit(name, obj):
            if hasattr(obj, "shape") and hasattr(obj, "dtype"):  # a dataset
                names.append(name.rsplit("/", 1)[0].strip("/"))

        fd.visititems(visit)
    return list(dict.fromkeys(names))


def read_image_u8(path, grayscale: bool = False) -> np.ndarray:
    """`read_image(path, grayscale)` of imcui/hloc/utils/io.py:11-21 as uint8: [H,W] when `grayscale`, else [H,W,3] RGB.
    cv2 (when installed) with the reference's own flags -- IMREAD_GRAYSCALE / IMREAD_COLOR: the decoder applies the EXIF
    orientation, reduces 16-bit files to 8 bits and drops alpha, and for `grayscale` hands back ITS gray (libjpeg's luma for
    JPEG), exactly what the reference feeds the extractor.  Without cv2, PIL reproduces those conversions: EXIF transpose,
    16-bit -> 8-bit by dropping the low byte (cv2's 1/256 scaling), palette / alpha / CMYK -> RGB; the gray of a colour PNG is libpng's
    (what cv2's PNG reader returns, `_png_as_read_image`), the gray of other colour files is left to the device kernel (cv2's RGB2GRAY
    fixed point), which is what the UI path (`extract`, :158-161) does with its RGB input."""
    try:
        import cv2
    except ImportError:
        cv2 = None
    if cv2 is not None:
        img = cv2.imread(str(path), cv2.IMREAD_GRAYSCALE if grayscale else cv2.IMREAD_COLOR)
        if img is None:
            raise ValueError(f"Cannot read image {path}.")
        if not grayscale and img.ndim == 3:
            img = img[:, :, ::-1]  # BGR to RGB
        return np.ascontiguousarray(img, dtype=np.uint8)
    from PIL import Image, ImageOps

    try:
        im = Image.open(str(path))
        im.load()
    except Exception as e:  # noqa: BLE001
        raise ValueError(f"Cannot read image {path}.") from e
    is_png = (im.format or "").upper() == "PNG"  # (the transposed copy below has no format)
    im = ImageOps.exif_transpose(im)
    if im.mode in ("I;16", "I;16B", "I;16L", "I"):
        arr = np.asarray(im).astype(np.int64)
        arr = (arr >> 8) if arr.max() > 255 else arr
        return np.ascontiguousarray(arr.clip(0, 255).astype(np.uint8))  # a 16-bit gray file: [H,W]
    if im.mode == "L":
        arr = np.array(im, dtype=np.uint8)
        return arr if grayscale else np.repeat(arr[:, :, None], 3, axis=2)  # IMREAD_COLOR replicates gray files
    if im.mode == "LA":
        im = im.convert("L")
        arr = np.array(im, dtype=np.uint8)
        return arr if grayscale else np.repeat(arr[:, :, None], 3, axis=2)
    if im.mode != "RGB":
        im = im.convert("RGB")  # palette, RGBA (alpha dropped), CMYK
    arr = np.array(im, dtype=np.uint8)  # a writable copy (torch.from_numpy)
    if grayscale and is_png:  # cv2's PNG reader converts inside libpng (see `_png_as_read_image`): the same gray as the device decoder's
        c = arr.astype(np.int32)
        return ((9797 * c[..., 0] + 19234 * c[..., 1] + 3737 * c[..., 2]) >> 15).astype(np.uint8)
    return arr  # other colour files: gray conversion runs on the device


def read_image_device(path, grayscale: bool, device, decode: str = "auto") -> torch.Tensor:
    """`read_image(path, grayscale)` with the result on the device: uint8 [H,W] (grayscale) or [H,W,3] RGB.
    decode = "auto": a baseline JPEG goes through the library's decoder -- Huffman on the host, inverse DCT / up-sampling / colour on the
    device, bit-exact against libjpeg's default path (utils/jpeg.py); with `grayscale` the result is the file's luma plane, which is
    what the reference's `cv2.imread(IMREAD_GRAYSCALE)` returns for a JPEG.  A PNG (8-bit, not interlaced: every PNG of the reference repository)
    is inflated on the host and un-filtered on the device (utils/png.py), bit-exact.  Files neither path takes (arithmetic-coded or CMYK JPEG,
    interlaced or 16-bit PNG, other formats) are read on the host (`read_image_u8`) and uploaded.  "host": always the host reader; "device": refuse
    instead of falling back."""
    if decode not in ("auto", "host", "device"):
        raise ValueError(f"decode = {decode!r}: auto | host | device")
    if decode != "host":
        from .utils.jpeg import JpegUnsupported, decode_jpeg, is_jpeg

        try:
            data = Path(path).read_bytes()
        except OSError as e:
            raise ValueError(f"Cannot read image {path}.") from e
        from .utils.png import PngUnsupported, decode_png, is_png

        if is_jpeg(data):
            try:
                return decode_jpeg(data, grayscale, device)
            except JpegUnsupported:
                if decode == "device":
                    raise
        elif is_png(data):
            try:
                return _png_as_read_image(decode_png(data, device), grayscale)
            except PngUnsupported:
                if decode == "device":
                    raise
        elif decode == "device":
            raise ValueError(f"{path}: neither a JPEG nor a PNG file (decode='device')")
    return torch.from_numpy(read_image_u8(path, grayscale)).to(device)


def _png_as_read_image(t: torch.Tensor, grayscale: bool) -> torch.Tensor:
    """The device PNG decoder's output as `cv2.imread` hands the file back: gray files stay [H,W] for `grayscale` and are replicated to three
    channels otherwise (IMREAD_COLOR); colour files stay RGB for IMREAD_COLOR and, for IMREAD_GRAYSCALE, are reduced to gray the way OpenCV's
    PNG reader does it -- INSIDE libpng (`png_set_rgb_to_gray(png_ptr, 1, 0.299, 0.587)`, grfmt_png.cpp), not with `cvtColor`:
    libpng turns the two weights into 15-bit integers by truncation (29900 * 32768 / 100000 = 9797, 58700 * 32768 / 100000 = 19234, blue =
    32768 - both = 3737) and TRUNCATES the sum, `(9797 R + 19234 G + 3737 B) >> 15` (`png_do_rgb_to_gray`, the 8-bit branch without gamma tables;
    R = G = B passes through, which the formula gives by itself since the weights sum to 2^15).  `cvtColor`'s 9798 / 19235 / 3735 with rounding
    differs from it by one level on many pixels."""
    if t.dim() == 2 and not grayscale:
        return t[:, :, None].expand(-1, -1, 3).contiguous()
    if t.dim() == 3 and grayscale:
        c = t.to(torch.int32)
        return ((9797 * c[..., 0] + 19234 * c[..., 1] + 3737 * c[..., 2]) >> 15).to(torch.uint8)
    return t


def read_images_device(paths, grayscale: bool, device, decode: str = "auto", decoder=None) -> list:
    """`read_image_device` for a list of files: the baseline JPEGs among them go through ONE `JpegDecoder.decode_batch` call (bit streams on
    the library's host threads into pinned staging, one transfer + three launches per group of equally shaped files), everything else --
    and every file the device path refuses -- through the host reader.  -> list of uint8 device tensors in the order of `paths`."""
    if decode not in ("auto", "host", "device"):
        raise ValueError(f"decode = {decode!r}: auto | host | device")
    out: list = [None] * len(paths)
    if decode != "host":
        from .utils.jpeg import JpegDecoder, JpegUnsupported, is_jpeg
        from .utils.png import PngDecoder, PngUnsupported, is_png

        blobs, where, pblobs, pwhere = [], [], [], []
        for i, p in enumerate(paths):
            try:
                data = Path(p).read_bytes()
            except OSError as e:
                raise ValueError(f"Cannot read image {p}.") from e
            if is_jpeg(data):
                blobs.append(data)
                where.append(i)
            elif is_png(data):
                pblobs.append(data)
                pwhere.append(i)
            elif decode == "device":
                raise ValueError(f"{p}: neither a JPEG nor a PNG file (decode='device')")
        if pblobs:  # PNG: zlib on the library's host threads, the scan-line filters undone on the device (utils/png.py)
            pdec = PngDecoder(device)
            for i, r in zip(pwhere, pdec.decode_batch(pblobs)):
                if isinstance(r, PngUnsupported):
                    if decode == "device":
                        raise r
                else:
                    out[i] = _png_as_read_image(r, grayscale)
            pdec.close()
        if blobs:
            own = decoder is None
            dec = JpegDecoder(device) if own else decoder
            for i, r in zip(where, dec.decode_batch(blobs, grayscale)):
                if isinstance(r, JpegUnsupported):
                    if decode == "device":
                        raise r
                else:
                    out[i] = r
            if own:
                dec.close()
    for i, p in enumerate(paths):
        if out[i] is None:
            out[i] = torch.from_numpy(read_image_u8(p, grayscale)).to(device)
    return out


def image_names(root: Path, conf: SimpleNamespace, paths=None) -> List[str]:
    """`ImageDataset.__init__` (:53-77): glob the root or take an explicit list; every name must exist."""
    root = Path(root)
    if paths is None:
        found = []
        for g in conf.globs:
            found += list(root.glob("**/" + g))
        if len(found) == 0:
            raise ValueError(f"Could not find any image in root: {root}.")
        return [p.relative_to(root).as_posix() for p in sorted(set(found))]
    if isinstance(paths, (Path, str)):
        with open(paths, "r") as fh:
            names = [ln.strip() for ln in fh if ln.strip() and not ln.startswith("#")]
    else:
        names = [p.as_posix() if isinstance(p, Path) else p for p in paths]
    for n in names:
        if not (root / n).exists():
            raise ValueError(f"Image {n} does not exists in root: {root}.")
    return names


def target_size(size, conf: SimpleNamespace):
    """(w, h) after `ImageDataset.__getitem__`'s resize rule (:85-91), or None when the image is kept."""
    if conf.resize_max and (conf.force_resize or max(size) > conf.resize_max):
        scale = conf.resize_max / max(size)
        return tuple(int(round(x * scale)) for x in size)
    return None


def preprocess_on_device(img_u8, conf: SimpleNamespace, device) -> torch.Tensor:
    """uint8 [H,W] / [H,W,3] (host array or device tensor) -> float32 [1,1,h,w] in [0,1] on the device, following :79-99."""
    if not conf.grayscale:
        raise NotImplementedError("the HIP extractors take gray images (SuperPoint conf `grayscale: True`)")
    if conf.interpolation != "cv2_area":
        raise NotImplementedError(f"interpolation {conf.interpolation!r}: only cv2_area runs on the device")
    t = (img_u8 if torch.is_tensor(img_u8) else torch.from_numpy(img_u8)).to(device)[None]
    h, w = t.shape[1:3]
    new = target_size((w, h), conf)
    if new is None:
        new = (w, h)
    if new[0] > w or new[1] > h:
        return backend.preprocess_linear(t, new)  # resize_image :29-31: INTER_AREA becomes INTER_LINEAR as soon as a side grows
    if new == (w, h) and t.dim() == 4:
        return backend.rgb_to_gray(t) if (h * w) % 4 == 0 else backend.preprocess_area(t, new)
    return backend.preprocess_area(t, new)


@torch.no_grad()
def main(conf: Dict, image_dir: Path, export_dir: Optional[Path] = None, as_half: bool = True,
         image_list: Optional[Union[Path, Sequence[str]]] = None, feature_path: Optional[Path] = None, overwrite: bool = False,
         model=None, batch_size: int = 32, device="cuda", decode: str = "auto", decode_threads: int = 8) -> Path:  # fmt: skip
    """Reference signature (:174-182) + `model` (a loaded HIP extractor plugin; built from conf["model"] when None),
    `batch_size` (images per C-ABI call), `decode` (`read_images_device`: "auto" = baseline JPEGs decoded on the device, 2 x batch_size
    files per `JpegDecoder.decode_batch` call) and `decode_threads` (host threads of the Huffman stage).  Returns the feature file path."""
    pconf = SimpleNamespace(**{**DEFAULT_PREPROCESSING, **conf.get("preprocessing", {})})
    image_dir = Path(image_dir)
    names = image_names(image_dir, pconf, image_list)
    if feature_path is None:
        feature_path = Path(export_dir, conf["output"] + ".h5")
    feature_path = Path(feature_path)
    feature_path.parent.mkdir(exist_ok=True, parents=True)
    skip = set(list_h5_names(feature_path) if feature_path.exists() and not overwrite else ())
    names = [n for n in names if n not in skip]
    if len(names) == 0:
        return feature_path
    if model is None:
        from . import extractors
        from .utils.base_model import dynamic_load

        model = dynamic_load(extractors, conf["model"]["name"])(conf["model"]).eval().to(device)
    device = next(model.buffers()).device
    noise = getattr(model, "detection_noise", 1)

    pending: Dict[tuple, list] = {}  # preprocessed (h, w) -> [(name, image tensor, original size)]

    def flush(key):
        items = pending.pop(key, [])
        if not items:
            return
        batch = torch.cat([it[1] for it in items], 0)
        if hasattr(model, "forward_checked"):  # counts + selection status in one copy; capacity overflow retried, failures raise
            out, counts = model.forward_checked(batch)
        else:
            out = model.forward_batched(batch)
            counts = out["num_keypoints"].tolist()
        kp, sc, de = out["keypoints"].cpu().numpy(), out["scores"].cpu().numpy(), out["descriptors"].cpu().numpy()
        with open_h5(feature_path, "a") as fd:
            for b, (name, img, original_size) in enumerate(items):
                n = counts[b]
                size = np.array(img.shape[-2:][::-1])
                scales = (original_size / size).astype(np.float32)
                pred = {
                    "keypoints": (kp[b, :n] + 0.5) * scales[None] - 0.5,
                    "scores": sc[b, :n],
                    "descriptors": np.ascontiguousarray(de[b, :n].T),  # [256, N] like the reference plugin
                    "image_size": original_size,
                }
                uncertainty = noise * scales.mean()
                if as_half:
                    pred = {k: (v.astype(np.float16) if v.dtype == np.float32 else v) for k, v in pred.items()}
                if name in fd:
                    del fd[name]
                grp = fd.create_group(name)
                for k, v in pred.items():
                    grp.create_dataset(k, data=v)
                grp["keypoints"].attrs["uncertainty"] = uncertainty

    decoder = None
    if decode != "host":
        from .utils.jpeg import JpegDecoder

        decoder = JpegDecoder(device, threads=decode_threads)
    chunk = max(1, 2 * batch_size)  # files read (and, for JPEG, decoded in one batch) ahead of the extractor
    for c0 in range(0, len(names), chunk):
        part = names[c0 : c0 + chunk]
        for name, img_u8 in zip(part, read_images_device([image_dir / n for n in part], pconf.grayscale, device, decode, decoder)):
            original_size = np.array(tuple(img_u8.shape[:2][::-1]))
            img = preprocess_on_device(img_u8, pconf, device)
            key = tuple(img.shape[-2:])
            pending.setdefault(key, []).append((name, img, original_size))
            if len(pending[key]) >= batch_size:
                flush(key)
    for key in list(pending):
        flush(key)
    if decoder is not None:
        decoder.close()
    return feature_path
