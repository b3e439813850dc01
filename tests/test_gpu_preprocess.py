"""Device preprocessing kernels of round 3 (GPU box only, through the C ABI): the growing resize (cv2.INTER_LINEAR) against its
numpy restatement, the dfactor resize against torch's own anti-aliased bilinear kernel -- both bit for bit -- and the batch
extractor on image files smaller than the `superpoint_max` force-resize target."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import preprocess as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("src_hw,dst_wh,channels", [((240, 320), (640, 480), 1), ((240, 320), (640, 480), 3), ((100, 37), (64, 250), 1),
                                                     ((480, 640), (800, 300), 3), ((31, 17), (32, 32), 1)])
def test_growing_resize_equals_the_restated_cv2_linear(src_hw, dst_wh, channels):
    """`resize_image(..., "cv2_area")` with a growing side = cv2.INTER_LINEAR on the float image (extract_features.py:29-31),
    then / 255: bit for bit against oracle/preprocess.py (parity unpinned: cv2 is absent), mixed grow / shrink included."""
    from imcui_hip import backend

    g = np.random.default_rng(src_hw[0] * 3 + channels)
    shape = (2, *src_hw) if channels == 1 else (2, *src_hw, 3)
    img = (g.random(shape) * 255).astype(np.uint8)
    out = backend.preprocess_linear(torch.from_numpy(img).to(DEV), dst_wh).cpu().numpy()
    for b in range(2):
        gray = img[b] if channels == 1 else P.rgb_to_gray_u8(img[b])
        want = P.resize_image_cv2_area(gray.astype(np.float32), dst_wh) / np.float32(255.0)
        assert out[b, 0].shape == want.shape
        assert np.array_equal(out[b, 0], want.astype(np.float32)), np.abs(out[b, 0] - want).max()


@pytest.mark.parametrize("H,W,h,w", [(487, 653, 480, 648), (300, 517, 296, 512), (96, 100, 48, 37), (33, 47, 32, 40), (517, 389, 512, 384),
                                      (1030, 771, 1024, 768), (31, 31, 64, 64)])
def test_dfactor_resize_equals_torch_antialias(H, W, h, w):
    """`F.resize(image, size_new, antialias=True)` (extract_features.py:142-148, match_dense.py:182): the HIP kernel against
    ATen's CPU kernel run right here -- the arithmetic the reference itself executes -- bit for bit, batch and channel planes."""
    from imcui_hip import backend

    g = torch.Generator().manual_seed(H + W)
    img = torch.rand(2, 3, H, W, generator=g)
    ref = F.interpolate(img, size=(h, w), mode="bilinear", align_corners=False, antialias=True)
    got = backend.resize_aa(img.to(DEV), (h, w)).cpu()
    assert got.shape == ref.shape
    assert torch.equal(got, ref), (got - ref).abs().max().item()
    same = backend.resize_aa(img.to(DEV), (H, W))
    assert torch.equal(same.cpu(), img)  # torchvision returns the image when the size already matches


def test_superpoint_max_accepts_small_images(tmp_path):
    """`superpoint_max` force-resizes EVERY image to 640 x 480 (configs/extractors.py:29-45): a 320 x 240 file (round 2 raised
    NotImplementedError), a 400 x 200 one (one side grows, one shrinks) and a larger one go through the batch extractor and
    equal the plugin called on the restated preprocessing."""
    from types import SimpleNamespace

    from PIL import Image

    from imcui_hip.hloc import extract_features as ef
    from imcui_hip.hloc.extractors.superpoint import SuperPoint
    from imcui_hip.hloc.utils.h5lite import open_h5
    from imcui_hip.synth import make_pair_batch
    from imcui_hip.synth_weights import superpoint_state_dict

    img0, img1, _ = make_pair_batch(2, 1, 480, 640, n_blobs=300)
    root = tmp_path / "images"
    root.mkdir()
    sizes = {"small.png": (320, 240), "mixed.png": (400, 200), "large.png": (1000, 750), "exact.png": (640, 480)}
    for (name, wh), im in zip(sizes.items(), [img0[0, 0], img1[0, 0], img0[0, 0], img1[0, 0]]):
        arr = (im * 255).round().to(torch.uint8).numpy()
        Image.fromarray(arr).resize(wh, Image.BICUBIC).save(root / name)
    conf = {"output": "feats-superpoint-max", "model": {"name": "superpoint", "nms_radius": 3, "max_keypoints": 1024, "keypoint_threshold": 0.005},
            "preprocessing": {"grayscale": True, "resize_max": 1024, "force_resize": True, "width": 640, "height": 480}}  # fmt: skip
    sp = SuperPoint({**conf["model"], "state_dict": superpoint_state_dict(0)}).eval().to(DEV)
    path = ef.main(conf, root, tmp_path / "out", model=sp, batch_size=3)
    pconf = SimpleNamespace(**{**ef.DEFAULT_PREPROCESSING, **conf["preprocessing"]})
    with open_h5(path, "r") as fd:
        for name, (w, h) in sizes.items():
            raw = ef.read_image_u8(root / name, True)
            new = ef.target_size((w, h), pconf)  # ImageDataset.__getitem__: resize_max / max(size), also when it grows (force_resize)
            want = P.resize_image_cv2_area(raw.astype(np.float32), new) / np.float32(255.0) if new != (w, h) else raw.astype(np.float32) / np.float32(255.0)
            image = ef.preprocess_on_device(raw, pconf, torch.device(DEV))
            assert np.array_equal(image[0, 0].cpu().numpy(), want.astype(np.float32)), name
            with torch.no_grad():
                pred = sp({"image": image})
            scales = (np.array([w, h]) / np.array(image.shape[-2:][::-1])).astype(np.float32)
            kp = ((pred["keypoints"][0].cpu().numpy() + 0.5) * scales[None] - 0.5).astype(np.float16)
            assert kp.shape[0] > 50, name
            assert np.array_equal(fd[name]["keypoints"].__array__(), kp), name
            assert np.array_equal(fd[name]["image_size"].__array__(), np.array([w, h]))
