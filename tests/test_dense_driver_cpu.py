"""Host logic of the batched dense-matching driver (imcui_hip/hloc/match_dense.py) on CPU with stand-ins for the device pieces:
every image of a chunk of `batch_size` pairs is read through ONE `read_images_device` call, images already cached are not read again,
the cache stays bounded when `cache_images` is off, pairs are grouped by preprocessed size, a pair whose first image is an existing
reference goes through the matcher with its images exchanged (imcui/hloc/match_dense.py:196-253), and the groups land in the file
under `names_to_pair`."""
import numpy as np
import torch

from imcui_hip.hloc import match_dense as md
from imcui_hip.hloc.match_features import names_to_pair
from imcui_hip.hloc.utils.h5lite import open_h5


class _StubDense(torch.nn.Module):
    """`forward_pairs` stand-in: for every pair one 'match' whose coordinates encode the mean grey level of either image, so the file
    tells which tensor went in as image0 / image1."""

    def __init__(self):
        super().__init__()
        self.register_buffer("dummy", torch.zeros(1))
        self.calls = []

    def forward_pairs(self, image0, image1):
        self.calls.append((tuple(image0.shape), tuple(image1.shape)))
        out = []
        for b in range(image0.shape[0]):
            a, c = float(image0[b].mean()), float(image1[b].mean())
            out.append({"keypoints0": torch.tensor([[a, a]]), "keypoints1": torch.tensor([[c, c]]), "scores": torch.tensor([0.5])})
        return out


def _install_standins(monkeypatch, sizes):
    reads = []

    def read_images_device(paths, grayscale, device, decode="auto", decoder=None):
        reads.append([p.name for p in paths])
        return [torch.full(sizes[p.name] + (1,), int(p.stem[1:]), dtype=torch.uint8) for p in paths]

    def preprocess_pair_image(img_u8, conf, device):
        h, w = img_u8.shape[:2]
        return img_u8[None, None, :, :, 0].float(), np.array([2.0, 2.0])  # "original resolution" = twice the tensor's

    monkeypatch.setattr(md, "read_images_device", read_images_device)
    monkeypatch.setattr(md, "preprocess_pair_image", preprocess_pair_image)
    monkeypatch.setattr(md, "read_image_device", lambda *a, **k: (_ for _ in ()).throw(AssertionError("one-file reader used: the chunk was not prefetched")))
    return reads


def test_dense_driver_chunked_reads_cache_and_flip(tmp_path, monkeypatch):
    names = [f"i{k}.jpg" for k in range(9)]
    sizes = {n: ((8, 16) if k < 6 else (16, 8)) for k, n in enumerate(names)}
    reads = _install_standins(monkeypatch, sizes)
    pairs = [("i0.jpg", "i1.jpg"), ("i0.jpg", "i2.jpg"), ("i3.jpg", "i1.jpg"), ("i4.jpg", "i5.jpg"), ("i6.jpg", "i7.jpg"), ("i8.jpg", "i6.jpg"), ("i2.jpg", "i6.jpg")]
    model = _StubDense()
    conf = {"model": {"name": "stub"}, "preprocessing": {"grayscale": True, "resize_max": 64, "dfactor": 8}}
    path = md.match_dense(conf, pairs, tmp_path, tmp_path / "m.h5", existing_refs={"i3.jpg"}, model=model, batch_size=2, decode="host")
    # one read call per chunk of two pairs, only the images not cached yet (the last chunk finds i2 and i6 in the cache: no call)
    assert reads == [["i0.jpg", "i1.jpg", "i2.jpg"], ["i3.jpg", "i4.jpg", "i5.jpg"], ["i6.jpg", "i7.jpg", "i8.jpg"]]
    assert all(len(r) <= 4 for r in reads)
    with open_h5(path, "r") as fd:
        for n0, n1 in pairs:
            grp = fd[names_to_pair(n0, n1)]
            k0, k1 = grp["keypoints0"].__array__(), grp["keypoints1"].__array__()
            v0, v1 = float(n0[1:-4]), float(n1[1:-4])
            # (k + 0.5) * 2 - 0.5 with k = the image's grey level: the flipped pair (i3 is an existing reference) must still come out in
            # (name0, name1) order
            assert np.allclose(k0, (v0 + 0.5) * 2 - 0.5) and np.allclose(k1, (v1 + 0.5) * 2 - 0.5), (n0, n1, k0, k1)
            assert np.allclose(grp["scores"].__array__(), 0.5)
    # batches never mix preprocessed sizes or flip states, and hold at most batch_size pairs
    assert all(s0[0] <= 2 and s0[0] == s1[0] for s0, s1 in model.calls)
    shapes = {(s0[-2:], s1[-2:]) for s0, s1 in model.calls}
    assert ((8, 16), (8, 16)) in shapes and ((16, 8), (16, 8)) in shapes and ((8, 16), (16, 8)) in shapes


def test_dense_driver_cache_is_bounded(tmp_path, monkeypatch):
    names = [f"i{k}.jpg" for k in range(40)]
    sizes = {n: (8, 8) for n in names}
    reads = _install_standins(monkeypatch, sizes)
    pairs = [(names[2 * k], names[2 * k + 1]) for k in range(20)] + [(names[0], names[1])]  # the last pair was evicted long ago: read again
    live = []
    orig = md.preprocess_pair_image

    def counting(img, conf, device):
        live.append(1)
        return orig(img, conf, device)

    monkeypatch.setattr(md, "preprocess_pair_image", counting)
    md.match_dense({"model": {"name": "stub"}, "preprocessing": {"cache_images": False}}, pairs, tmp_path, tmp_path / "m.h5", model=_StubDense(), batch_size=2,
                   decode="host")
    assert sum(len(r) for r in reads) == 42 and reads[-1] == ["i0.jpg", "i1.jpg"]
    reads.clear()
    md.match_dense({"model": {"name": "stub"}, "preprocessing": {"cache_images": True}}, pairs, tmp_path, tmp_path / "m2.h5", model=_StubDense(), batch_size=2,
                   decode="host")
    assert sum(len(r) for r in reads) == 40  # everything stays cached
