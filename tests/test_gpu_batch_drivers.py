"""File-based hloc batch drivers on the GPU: images on disk -> device-side preprocessing ->
SuperPoint in batches -> feature .h5 -> pair list -> LightGlue in batches -> match .h5, compared with the one-image /
one-pair-per-call plugin path the reference runs (imcui/hloc/extract_features.py:199-243, match_features.py:172-185)."""
import os

import numpy as np
import pytest
import torch

from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict
from oracle.preprocess import area_resize_f32, rgb_to_gray_u8
from test_gpu_real_images import load_pairs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("src_hw,dst_wh,channels", [((750, 1000), (640, 480), 3), ((673, 1013), (640, 425), 3), ((960, 1280), (640, 480), 1),
                                                     ((1024, 1536), (512, 512), 3), ((485, 641), (640, 480), 1)])  # fmt: skip
def test_device_preprocessing_equals_the_restated_host_path(src_hw, dst_wh, channels):
    """gray (cv2 fixed point) -> float32 -> INTER_AREA -> / 255 on the device vs oracle/preprocess.py: bit-exact, for
    fractional factors (decimation tables), integer factors (block mean) and mixed ones."""
    from imcui_hip import backend

    g = np.random.default_rng(src_hw[0] + dst_wh[0])
    img = g.integers(0, 256, size=(2, *src_hw, channels), dtype=np.uint8)
    out = backend.preprocess_area(torch.from_numpy(img).cuda(), dst_wh).cpu().numpy()
    assert out.shape == (2, 1, dst_wh[1], dst_wh[0])
    for b in range(2):
        gray = rgb_to_gray_u8(img[b]) if channels == 3 else img[b, ..., 0]
        ref = area_resize_f32(gray.astype(np.float32), dst_wh) / np.float32(255.0)
        assert np.array_equal(out[b, 0], ref), np.abs(out[b, 0] - ref).max()


def test_extract_then_match_from_files_equals_the_per_call_plugins(tmp_path):
    from PIL import Image

    from imcui_hip.hloc import extract_features as ef
    from imcui_hip.hloc import match_features as mf
    from imcui_hip.hloc.extractors.superpoint import SuperPoint
    from imcui_hip.hloc.matchers.lightglue import LightGlue
    from imcui_hip.hloc.utils.h5lite import open_h5

    names, img0, img1, _ = load_pairs()
    root = tmp_path / "images"
    (root / "db").mkdir(parents=True)
    (root / "query").mkdir()
    files = []
    for i in range(6):  # 12 real 640 x 480 gray images as PNG files, one stored as RGB and one at a size that needs the area resize
        for side, img in (("db", img0), ("query", img1)):
            arr = (img[i, 0] * 255).round().to(torch.uint8).numpy()
            if i == 1 and side == "db":
                arr = np.stack([arr, arr, arr], -1)
            if i == 2 and side == "query":
                arr = np.asarray(Image.fromarray(arr).resize((1280, 960), Image.BICUBIC))
            Image.fromarray(arr).save(root / side / f"{names[i]}.png")
            files.append(f"{side}/{names[i]}.png")
    conf = {"output": "feats-superpoint", "model": {"name": "superpoint", "nms_radius": 3, "max_keypoints": 1024, "keypoint_threshold": 0.005},
            "preprocessing": {"grayscale": True, "resize_max": 640}}  # fmt: skip
    sp = SuperPoint({**conf["model"], "state_dict": superpoint_state_dict(0)}).eval().to("cuda:0")
    feature_path = ef.main(conf, root, tmp_path / "out", model=sp, batch_size=5)
    assert feature_path == tmp_path / "out" / "feats-superpoint.h5"
    assert sorted(ef.list_h5_names(feature_path)) == sorted(files)
    # every image: the file content equals the plugin called on that one image (reference flow), at as_half precision
    from types import SimpleNamespace

    pconf = SimpleNamespace(**{**ef.DEFAULT_PREPROCESSING, **conf["preprocessing"]})
    with open_h5(feature_path, "r") as fd:
        for f in files:
            raw = ef.read_image_u8(root / f)
            image = ef.preprocess_on_device(raw, pconf, torch.device("cuda:0"))
            assert tuple(image.shape[-2:]) == (480, 640)
            with torch.no_grad():
                pred = sp({"image": image})
            scales = (np.array(raw.shape[:2][::-1]) / np.array([640, 480])).astype(np.float32)
            kp = ((pred["keypoints"][0].cpu().numpy() + 0.5) * scales[None] - 0.5).astype(np.float16)
            grp = fd[f]
            assert grp["keypoints"].__array__().dtype == np.float16 and grp["descriptors"].__array__().shape == (256, kp.shape[0])
            assert np.array_equal(grp["keypoints"].__array__(), kp)
            assert np.array_equal(grp["scores"].__array__(), pred["scores"][0].cpu().numpy().astype(np.float16))
            assert np.array_equal(grp["descriptors"].__array__(), pred["descriptors"][0].cpu().numpy().astype(np.float16))
            assert tuple(grp["image_size"].__array__()) == tuple(raw.shape[:2][::-1])
            assert float(grp["keypoints"].attrs["uncertainty"]) == pytest.approx(2.0 * scales.mean())
    # a second call exports nothing new
    assert ef.main(conf, root, tmp_path / "out", model=sp) == feature_path
    # ---- matching from the files
    pairs_path = tmp_path / "pairs.txt"
    pairs = [(f"query/{names[i]}.png", f"db/{names[i]}.png") for i in range(6)] + [(f"query/{names[0]}.png", f"db/{names[3]}.png")]
    pairs_path.write_text("".join(f"{q} {r}\n" for q, r in pairs))
    lg = LightGlue({"depth_confidence": 0.95, "width_confidence": 0.99, "match_threshold": 0.1, "state_dict": lightglue_state_dict(0)}).eval().to("cuda:0")
    match_path = mf.match_from_paths(lg, pairs_path, tmp_path / "out" / "matches.h5", feature_path, feature_path, batch_size=4)
    store = mf.H5FeatureStore(feature_path)
    total = 0
    with open_h5(match_path, "r") as fd:
        for q, r in pairs:
            f0, f1 = store.get(q), store.get(r)
            data = {"image0": torch.empty((1, 1) + tuple(int(v) for v in f0["image_size"])[::-1]), "image1": torch.empty((1, 1) + tuple(int(v) for v in f1["image_size"])[::-1])}
            for side, f in (("0", f0), ("1", f1)):
                data["keypoints" + side] = torch.from_numpy(f["keypoints"].astype(np.float32))[None].cuda()
                data["scores" + side] = torch.from_numpy(f["scores"].astype(np.float32))[None].cuda()
                data["descriptors" + side] = torch.from_numpy(f["descriptors"].astype(np.float32))[None].cuda()
            with torch.no_grad():
                pred = lg(data)
            grp = fd[mf.names_to_pair(q, r)]
            m = grp["matches0"].__array__()
            assert m.dtype == np.int16 and np.array_equal(m, pred["matches0"][0].cpu().numpy().astype(np.int16)), (q, r)
            assert np.array_equal(grp["matching_scores0"].__array__(), pred["matching_scores0"][0].cpu().numpy().astype(np.float16))
            total += int((m >= 0).sum())
    assert total > 10


@pytest.mark.parametrize("matcher,ext", [("loftr", "png"), ("eloftr", "png"), ("loftr", "jpg")])
def test_dense_driver_equals_one_pair_per_call(tmp_path, matcher, ext):
    """`match_dense` on image files (imcui/hloc/match_dense.py:196-253), B pairs per C-ABI call, against the reference flow:
    one `model({"image0", "image1"})` call per pair on the same preprocessed tensors, key-points rescaled to the original
    resolution, groups `name0/name1` with keypoints0 / keypoints1 / scores.  Includes a pair that is flipped because its
    first image is an existing reference, an image that needs the area resize, and one whose size needs the dfactor resize.
    The driver reads the files of a chunk of pairs in one batched call (`read_images_device`): PNG files take the host reader, the JPEG
    files the device decoder, whose pixels equal the host reader's bit for bit -- so the per-pair flow below (host reader) must agree."""
    from types import SimpleNamespace

    from PIL import Image

    from imcui_hip.hloc import match_dense as md
    from imcui_hip.hloc.match_features import names_to_pair
    from imcui_hip.hloc.utils.h5lite import open_h5
    from imcui_hip.synth import make_shifted_pair
    from imcui_hip.synth_weights import eloftr_state_dict, loftr_state_dict

    root = tmp_path / "images"
    root.mkdir()
    names = []
    for i, (hw, shift) in enumerate([((256, 320), (16, 8)), ((256, 320), (8, 24)), ((256, 320), (-16, 0)), ((512, 640), (32, 16)), ((262, 325), (16, 8))]):
        i0, i1, _ = make_shifted_pair(40 + i, hw[0], hw[1], shift, 900)
        for side, img in (("a", i0), ("b", i1)):
            Image.fromarray((img[0, 0] * 255).round().to(torch.uint8).numpy()).save(root / f"{side}{i}.{ext}", **({"quality": 95} if ext == "jpg" else {}))
        names.append((f"a{i}.{ext}", f"b{i}.{ext}"))
    df = 32 if matcher == "eloftr" else 8
    conf = {"model": {"name": matcher, "match_threshold": 0.2, "max_keypoints": 500}, "preprocessing": {"grayscale": True, "resize_max": 320, "dfactor": df}}
    if matcher == "eloftr":
        from imcui_hip.hloc.matchers.eloftr import ELoFTR as Model

        sd = eloftr_state_dict(0)
    else:
        from imcui_hip.hloc.matchers.loftr import LoFTR as Model

        sd = loftr_state_dict(0)
    model = Model({**conf["model"], "state_dict": sd}).eval().to("cuda:0")
    existing = {f"a1.{ext}"}
    path = md.match_dense(conf, names, root, tmp_path / "dense.h5", existing_refs=existing, model=model, batch_size=3)
    pconf = SimpleNamespace(**{**md.DEFAULT_PREPROCESSING, **conf["preprocessing"]})
    total = 0
    with open_h5(path, "r") as fd:
        for n0, n1 in names:
            im0, s0 = md.preprocess_pair_image(md.read_image_u8(root / n0), pconf, torch.device("cuda:0"))
            im1, s1 = md.preprocess_pair_image(md.read_image_u8(root / n1), pconf, torch.device("cuda:0"))
            assert im0.shape[-1] % df == 0 and im0.shape[-2] % df == 0 and max(im0.shape[-2:]) <= 320
            with torch.no_grad():
                if n0 in existing:
                    pred = model({"image0": im1, "image1": im0})
                    pred = {**pred, "keypoints0": pred["keypoints1"], "keypoints1": pred["keypoints0"]}
                else:
                    pred = model({"image0": im0, "image1": im1})
            grp = fd[names_to_pair(n0, n1)]
            k0 = ((pred["keypoints0"] + 0.5) * pred["keypoints0"].new_tensor(s0) - 0.5).cpu().numpy()
            k1 = ((pred["keypoints1"] + 0.5) * pred["keypoints1"].new_tensor(s1) - 0.5).cpu().numpy()
            # the same set of matches; the order inside a pair may differ where confidences tie at the top-k sort
            got = np.concatenate([grp["keypoints0"].__array__(), grp["keypoints1"].__array__(), grp["scores"].__array__()[:, None]], 1)
            want = np.concatenate([k0, k1, pred["scores"].cpu().numpy()[:, None]], 1)
            assert got.shape == want.shape and got.dtype == np.float32
            assert np.allclose(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])], atol=1e-4)
            total += len(got)
    assert total > 300
