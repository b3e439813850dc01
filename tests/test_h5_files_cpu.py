"""The hloc on-disk format written by the batch drivers is REAL HDF5: the feature file of
imcui/hloc/extract_features.py:221-243 and the match file of imcui/hloc/match_features.py:73-83, through
`imcui_hip.hloc.utils.h5lite.open_h5` (h5py when installed, the HDF5 C library through ctypes otherwise), exercised
file-based end to end with a stand-in matcher, and read back by the real h5py where an interpreter that has it exists
(/opt/conda/bin/python3.9 in the build image)."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

from imcui_hip.hloc import match_features as mf
from imcui_hip.hloc.utils import h5lite
from test_match_driver_cpu import _features, _StubMatcher

try:
    h5lite.open_h5(os.devnull + ".h5", "r")
except ImportError:
    pytest.skip("neither h5py nor libhdf5 here", allow_module_level=True)
except Exception:  # noqa: BLE001  (file missing: the library itself is available)
    pass

H5PY_PYTHON = next((p for p in ("/opt/conda/bin/python3.9",) if os.path.exists(p)), None)


def _write_features(path, feats):
    with h5lite.open_h5(path, "a") as fd:
        for name, f in feats.items():
            grp = fd.create_group(name)
            for k, v in f.items():
                grp.create_dataset(k, data=v)
            grp["keypoints"].attrs["uncertainty"] = 2.0


def test_feature_and_match_files_round_trip(tmp_path):
    names = ["db/a.jpg", "db/b.jpg", "query/c.jpg", "query/d.jpg"]
    feats = _features(3, names, [(40, (640, 480)), (35, (640, 480)), (50, (640, 480)), (22, (320, 240))])
    fpath, mpath, ppath = tmp_path / "feats.h5", tmp_path / "out" / "matches.h5", tmp_path / "pairs.txt"
    _write_features(fpath, feats)
    ppath.write_text("query/c.jpg db/a.jpg\nquery/c.jpg db/b.jpg\nquery/d.jpg db/a.jpg\ndb/a.jpg query/c.jpg\n")
    store = mf.H5FeatureStore(fpath)
    assert "db/a.jpg" in store and "db/zzz.jpg" not in store
    got = store.get("query/d.jpg")
    assert got["keypoints"].dtype == np.float16 and got["descriptors"].shape == (64, 22) and tuple(got["image_size"]) == (320, 240)
    model = _StubMatcher()
    model.register_buffer("dummy", torch.zeros(1))
    assert mf.match_from_paths(model, ppath, mpath, fpath, fpath, batch_size=2) == mpath
    ref = mf.DictMatchSink()
    pairs = [("query/c.jpg", "db/a.jpg"), ("query/c.jpg", "db/b.jpg"), ("query/d.jpg", "db/a.jpg")]  # the reversed duplicate is dropped
    mf.match_from_pairs(_StubMatcher(), pairs, mf.DictFeatureStore(feats), mf.DictFeatureStore(feats), ref, batch_size=1, device=torch.device("cpu"))
    with h5lite.open_h5(mpath, "r") as fd:
        assert sorted(fd.keys()) == ["query-c.jpg", "query-d.jpg"]
        for q, r in pairs:
            key = mf.names_to_pair(q, r)
            grp = fd[key]
            m, s = grp["matches0"].__array__(), grp["matching_scores0"].__array__()
            assert m.dtype == np.int16 and s.dtype == np.float16 and m.shape == (feats[q]["keypoints"].shape[0],)
            assert np.array_equal(m, ref.matches[key]["matches0"]) and np.array_equal(s, ref.matches[key]["matching_scores0"])
    # a second run finds every pair done (either key order) and leaves the file alone
    before = os.path.getmtime(mpath)
    calls = len(model.calls)
    mf.match_from_paths(model, ppath, mpath, fpath, fpath, batch_size=2)
    assert len(model.calls) == calls and os.path.getmtime(mpath) == before


@pytest.mark.skipif(H5PY_PYTHON is None, reason="no interpreter with the real h5py in this environment")
def test_files_written_without_h5py_are_read_by_the_real_h5py(tmp_path):
    """Pins the ctypes writer to the reference's own reader: h5py (3.3.0 under /opt/conda) opens the files, sees the
    groups / dtypes / attribute the reference's readers expect (utils/io.py: `get_keypoints` reads
    `dset.attrs.get("uncertainty")`, `get_matches` reads matches0 / matching_scores0) and returns identical values."""
    feats = _features(5, ["a/x.png", "y.png"], [(30, (640, 480)), (17, (800, 600))])
    fpath, mpath = tmp_path / "f.h5", tmp_path / "m.h5"
    with h5lite.File(fpath, "a") as fd:  # force the ctypes implementation even where h5py exists
        for name, f in feats.items():
            grp = fd.create_group(name)
            for k, v in f.items():
                grp.create_dataset(k, data=v)
            grp["keypoints"].attrs["uncertainty"] = 1.5
    with h5lite.File(mpath, "a") as fd:
        grp = fd.create_group(mf.names_to_pair("a/x.png", "y.png"))
        grp.create_dataset("matches0", data=np.array([3, -1, 0, 7], dtype=np.int16))
        grp.create_dataset("matching_scores0", data=np.array([0.9, 0.0, 0.25, 0.5], dtype=np.float16))
    code = (
        "import h5py, json, numpy as np\n"
        f"f = h5py.File({str(fpath)!r}, 'r', libver='latest'); m = h5py.File({str(mpath)!r}, 'r', libver='latest')\n"
        "names = []\n"
        "f.visititems(lambda _, o: names.append(o.parent.name.strip('/')) if isinstance(o, h5py.Dataset) else None)\n"
        "g = f['a/x.png']; p = m['a-x.png/y.png']\n"
        "print(json.dumps({'names': sorted(set(names)), 'kdtype': str(g['keypoints'].dtype), 'kshape': list(g['keypoints'].shape),\n"
        "  'unc': float(g['keypoints'].attrs.get('uncertainty')), 'ksum': float(g['keypoints'][()].astype('f8').sum()),\n"
        "  'dsum': float(g['descriptors'][()].astype('f8').sum()), 'size': [int(v) for v in g['image_size'][()]],\n"
        "  'm': p['matches0'][()].tolist(), 'mdtype': str(p['matches0'].dtype), 's': p['matching_scores0'][()].astype('f8').tolist(),\n"
        "  'sdtype': str(p['matching_scores0'].dtype)}))\n"
    )
    r = subprocess.run([H5PY_PYTHON, "-W", "ignore", "-c", code], capture_output=True, text=True, timeout=120, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    f = feats["a/x.png"]
    assert got["names"] == ["a/x.png", "y.png"] and got["kdtype"] == "float16" and got["kshape"] == [30, 2] and got["unc"] == 1.5
    assert got["ksum"] == float(f["keypoints"].astype("f8").sum()) and got["dsum"] == float(f["descriptors"].astype("f8").sum())
    assert got["size"] == [640, 480] and got["m"] == [3, -1, 0, 7] and got["mdtype"] == "int16"
    assert got["s"] == [float(np.float16(0.9)), 0.0, 0.25, 0.5] and got["sdtype"] == "float16"


def test_extract_driver_host_logic(tmp_path):
    """Image listing, resize rule and the skip-existing logic of the extraction driver (no GPU: the device calls are not reached)."""
    from types import SimpleNamespace

    from PIL import Image

    from imcui_hip.hloc import extract_features as ef

    root = tmp_path / "imgs"
    (root / "sub").mkdir(parents=True)
    for name, size in (("a.png", (64, 48)), ("sub/b.jpg", (100, 80))):
        Image.fromarray(np.zeros((size[1], size[0], 3), dtype=np.uint8)).save(root / name)
    conf = SimpleNamespace(**ef.DEFAULT_PREPROCESSING)
    assert ef.image_names(root, conf) == ["a.png", "sub/b.jpg"]
    assert ef.image_names(root, conf, ["sub/b.jpg"]) == ["sub/b.jpg"]
    with pytest.raises(ValueError):
        ef.image_names(root, conf, ["missing.png"])
    assert ef.read_image_u8(root / "a.png").shape == (48, 64, 3)
    conf.resize_max = 1600
    assert ef.target_size((1000, 750), conf) is None  # below resize_max and not forced
    conf.force_resize = True
    assert ef.target_size((1000, 750), conf) == (1600, 1200)
    conf.resize_max, conf.force_resize = 640, False
    assert ef.target_size((1000, 750), conf) == (640, 480) and ef.target_size((1013, 673), conf) == (640, 425)
    # names already in the feature file are skipped: with everything exported, main() returns before touching a device
    fpath = tmp_path / "feats.h5"
    _write_features(fpath, _features(1, ["a.png", "sub/b.jpg"], [(5, (64, 48)), (6, (100, 80))]))
    assert sorted(ef.list_h5_names(fpath)) == ["a.png", "sub/b.jpg"]
    out = ef.main({"output": "x", "preprocessing": {"grayscale": True}}, root, feature_path=fpath, model=object())
    assert out == fpath


def test_read_image_follows_the_reference_reader(tmp_path):
    """`read_image` (imcui/hloc/utils/io.py:11-21) decodes with IMREAD_GRAYSCALE / IMREAD_COLOR: the EXIF orientation is applied,
    16-bit files are reduced to 8 bits, gray / gray+alpha / palette files come back as 3-channel RGB in colour mode.  The reader of
    the batch drivers must do the same with whichever decoder is installed."""
    from PIL import Image

    from imcui_hip.hloc import extract_features as ef

    g = np.random.default_rng(0)
    rgb = (g.random((40, 60, 3)) * 255).astype(np.uint8)
    # EXIF orientation 6 (rotate 90 degrees clockwise on display): a 60 x 40 file shows as 40 x 60
    im = Image.fromarray(rgb)
    exif = Image.Exif()
    exif[0x0112] = 6
    im.save(tmp_path / "rot.png", exif=exif)
    out = ef.read_image_u8(tmp_path / "rot.png")
    assert out.shape == (60, 40, 3) and out.dtype == np.uint8
    assert np.array_equal(out, np.rot90(rgb, k=-1))
    # 16-bit gray PNG: value / 256, not a wrap-around
    g16 = (g.random((20, 30)) * 65535).astype(np.uint16)
    Image.fromarray(g16).save(tmp_path / "deep.png")
    d = ef.read_image_u8(tmp_path / "deep.png", grayscale=True)
    assert d.shape == (20, 30) and np.array_equal(d, (g16 >> 8).astype(np.uint8))
    # gray + alpha and palette files: colour mode returns 3 channels, gray mode 2-D
    la = Image.fromarray(np.stack([rgb[..., 0], np.full((40, 60), 128, np.uint8)], -1), mode="LA")
    la.save(tmp_path / "la.png")
    assert ef.read_image_u8(tmp_path / "la.png").shape == (40, 60, 3)
    assert np.array_equal(ef.read_image_u8(tmp_path / "la.png", grayscale=True), rgb[..., 0])
    Image.fromarray(rgb).convert("P").save(tmp_path / "pal.png")
    assert ef.read_image_u8(tmp_path / "pal.png").shape == (40, 60, 3)
    with pytest.raises(ValueError, match="Cannot read image"):
        ef.read_image_u8(tmp_path / "missing.png")


def test_resize_tables_of_the_library_equal_the_restatements():
    """The host tap tables the device kernels consume (imcui_hip_linear_table, imcui_hip_aa_table; no GPU needed) against
    oracle/preprocess.py -- the anti-aliased table is thereby pinned to torch (tests/test_oracle_preprocess.py)."""
    from imcui_hip import backend
    from oracle import preprocess as P

    for s, d in ((240, 480), (320, 640), (100, 37), (517, 512)):
        i0, i1, w1 = backend.linear_table_host(s, d, True)
        xi, xw = P._linear_table(s, d)
        x0, x1, a1 = xi.copy(), xi + 1, xw.copy()
        x0[xi < 0], x1[xi < 0], a1[xi < 0] = 0, 0, 0.0
        x0[xi >= s - 1], x1[xi >= s - 1], a1[xi >= s - 1] = s - 1, s - 1, 0.0
        assert np.array_equal(i0, x0) and np.array_equal(i1, x1) and np.array_equal(w1, a1)
        j0, j1, v1 = backend.linear_table_host(s, d, False)
        assert np.array_equal(j0, np.clip(xi, 0, s - 1)) and np.array_equal(j1, np.clip(xi + 1, 0, s - 1)) and np.array_equal(v1, xw)
    for s, d in ((487, 480), (653, 648), (100, 37), (33, 32), (31, 64), (1030, 1024)):
        first, count, w = backend.aa_table_host(s, d)
        tab = P.aa_table(s, d)
        for i, (xmin, ws) in enumerate(tab):
            assert first[i] == xmin and count[i] == len(ws) and np.array_equal(w[i, : len(ws)], ws) and not w[i, len(ws) :].any()
