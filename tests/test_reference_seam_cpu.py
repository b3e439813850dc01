"""The drop-in seam exercised with the REFERENCE's own loader (imcui/hloc/utils/base_model.py:46-55 `dynamic_load`)
and its own `BaseModel`, on the two-line overlay modules INTEGRATION.md section 2 tells a maintainer to add.

Runs in a subprocess from a scratch directory with /root/reference on PYTHONPATH (importing `imcui.hloc` truncates
./log.txt, hloc/__init__.py:29-30).  Skipped where the reference checkout does not exist (the GPU box)."""
import os
import subprocess
import sys
import textwrap

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

OVERLAYS = {  # verbatim from INTEGRATION.md section 2
    "extractors/superpoint.py": "from imcui_hip.hloc.extractors.superpoint import SuperPoint as _HipSuperPoint\nclass SuperPoint(_HipSuperPoint):\n    pass\n",
    "matchers/lightglue.py": "from imcui_hip.hloc.matchers.lightglue import LightGlue as _HipLightGlue\nclass LightGlue(_HipLightGlue):\n    pass\n",
    "matchers/loftr.py": "from imcui_hip.hloc.matchers.loftr import LoFTR as _HipLoFTR\nclass LoFTR(_HipLoFTR):\n    pass\n",
    "matchers/superglue.py": "from imcui_hip.hloc.matchers.superglue import SuperGlue as _HipSuperGlue\nclass SuperGlue(_HipSuperGlue):\n    pass\n",
    "matchers/duster.py": "from imcui_hip.hloc.matchers.duster import Duster as _HipDuster\nclass Duster(_HipDuster):  # AsymmetricCroCo3DStereo state dict (duster_vit_large.pth); the network runs in libimcui_hip, the aligner stays upstream's\n    pass\n",
    "matchers/mast3r.py": "from imcui_hip.hloc.matchers.mast3r import Mast3r as _HipMast3r\nclass Mast3r(_HipMast3r):  # AsymmetricMASt3R state dict ('catmlp+dpt' head); network and reciprocal matching on the device\n    pass\n",
    "matchers/dual_softmax.py": "from imcui_hip.hloc.matchers.dual_softmax import DualSoftMax as _HipDS\nclass DualSoftMax(_HipDS):\n    pass\n",
    "matchers/nearest_neighbor.py": "from imcui_hip.hloc.matchers.nearest_neighbor import NearestNeighbor as _HipNN\nclass NearestNeighbor(_HipNN):\n    pass\n",
}

DRIVER = textwrap.dedent(
    """
    import sys, torch
    from imcui.hloc.utils.base_model import BaseModel, dynamic_load          # the reference's own seam
    import overlay.extractors as extractors, overlay.matchers as matchers
    from imcui_hip import ImcuiHipError
    from imcui_hip.synth_weights import dust3r_state_dict, lightglue_state_dict, loftr_state_dict, superglue_state_dict, superpoint_state_dict

    def model_size(m):  # imcui/ui/modelcache.py:84-87
        return sum(p.numel() * p.element_size() for p in m.parameters()) + sum(b.numel() * b.element_size() for b in m.buffers())

    SP = dynamic_load(extractors, "superpoint")
    LG = dynamic_load(matchers, "lightglue")
    LF = dynamic_load(matchers, "loftr")
    SG = dynamic_load(matchers, "superglue")
    DS = dynamic_load(matchers, "dual_softmax")
    NN = dynamic_load(matchers, "nearest_neighbor")
    DU = dynamic_load(matchers, "duster")      # exactly one BaseModel subclass DEFINED in the overlay module (base_model.py:49-55)
    MA = dynamic_load(matchers, "mast3r")
    assert DU.__module__ == "overlay.matchers.duster" and MA.__module__ == "overlay.matchers.mast3r"
    small = {"enc_dim": 128, "enc_depth": 1, "dec_dim": 64, "dec_depth": 4}
    du = DU({"state_dict": dust3r_state_dict(0, small)}).eval().to("cpu")
    ma = MA({"state_dict": dust3r_state_dict(0, {**small, "desc_dim": 24})}).eval().to("cpu")
    assert du.conf["max_keypoints"] == 3000 and ma.conf["max_keypoints"] == 2000 and du.conf["vit_patch_size"] == 16  # duster.py:24-29, mast3r.py:24-29
    assert ma.net_cfg["desc_dim"] == 24 and du.net_cfg["desc_dim"] == 0
    try:
        MA({"state_dict": dust3r_state_dict(0, small)})
    except KeyError as e:
        assert "head_local_features" in str(e)
    else:
        raise SystemExit("a DUSt3R state dict was accepted as MASt3R weights")
    for cls, mod in ((SP, "overlay.extractors.superpoint"), (LG, "overlay.matchers.lightglue"), (LF, "overlay.matchers.loftr"),
                     (SG, "overlay.matchers.superglue"), (DS, "overlay.matchers.dual_softmax"), (NN, "overlay.matchers.nearest_neighbor")):
        assert issubclass(cls, BaseModel) and cls.__module__ == mod, (cls, cls.__mro__)
    # construction + the callers' `.eval().to(DEVICE)` (ui/utils.py:123,138) on a CPU-only host
    sp = SP({"max_keypoints": 512, "state_dict": superpoint_state_dict(0)}).eval().to("cpu")
    lg = LG({"match_threshold": 0.3, "state_dict": lightglue_state_dict(0)}).eval().to("cpu")
    lf = LF({"state_dict": loftr_state_dict(0)}).eval().to("cpu")
    sg = SG({"state_dict": superglue_state_dict(0)}).eval().to("cpu")
    assert sp.conf["nms_radius"] == 4 and sp.conf["max_keypoints"] == 512 and lg.conf["filter_threshold"] == 0.3
    # byte counts the ARC model cache would book: the packed weights are registered buffers
    sizes = {n: model_size(m) for n, m in (("sp", sp), ("lg", lg), ("lf", lf), ("sg", sg))}
    assert sizes["sp"] >= 4 * 1300865 and sizes["lg"] >= 4 * 11_000_000 and sizes["lf"] >= 4 * 11_000_000 and sizes["sg"] >= 4 * 11_000_000, sizes
    assert all(model_size(m) == sum(b.numel() * b.element_size() for b in m.buffers()) for m in (sp, lg, lf, sg))  # no parameters, only buffers
    # no CPU fallback: a CPU tensor fails loudly and cleanly through the reference's BaseModel.forward
    for model, data in ((sp, {"image": torch.zeros(1, 1, 64, 64)}),
                        (lf, {"image0": torch.zeros(1, 1, 64, 64), "image1": torch.zeros(1, 1, 64, 64)}),
                        (du, {"image0": torch.zeros(1, 3, 64, 64), "image1": torch.zeros(1, 3, 64, 64)}),
                        (ma, {"image0": torch.zeros(1, 3, 64, 64), "image1": torch.zeros(1, 3, 64, 64)})):
        try:
            model(data)
        except ImcuiHipError as e:
            assert "no CPU fallback" in str(e) or "ROCm" in str(e), e
        else:
            raise SystemExit("a CPU tensor did not raise ImcuiHipError")
    # required_inputs are asserted by the reference's forward (base_model.py:21-25)
    try:
        lg({"image0": torch.zeros(1, 1, 8, 8)})
    except AssertionError as e:
        assert "Missing key" in str(e)
    else:
        raise SystemExit("missing key not detected")
    # the pure-host plugins run end to end on CPU-free inputs (empty descriptor sets)
    out = NN({})({"descriptors0": torch.zeros(1, 128, 5), "descriptors1": torch.zeros(1, 128, 0)})
    assert (out["matches0"] == -1).all()
    print("SEAM_OK", sizes)
    """
)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "imcui")), reason="needs the reference checkout (build container only)")
def test_overlay_modules_load_through_the_references_own_dynamic_load(tmp_path, lib):
    pkg = tmp_path / "overlay"
    for sub in ("", "extractors", "matchers"):
        (pkg / sub).mkdir(parents=True, exist_ok=True)
        (pkg / sub / "__init__.py").write_text("")
    for rel, src in OVERLAYS.items():
        (pkg / rel).write_text(src)
    (tmp_path / "driver.py").write_text(DRIVER)
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([str(tmp_path), REF, os.path.join(ROOT, "image-matching-webui_amd"), ROOT])
    r = subprocess.run([sys.executable, "driver.py"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SEAM_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_integration_md_shows_the_overlays_this_test_uses():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for src in OVERLAYS.values():
        first = src.splitlines()[0]
        assert first in text, first
