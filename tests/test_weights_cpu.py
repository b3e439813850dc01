"""Checkpoint FILES through conf["weights_path"] (no GPU): every container format the upstream projects save, the reference's
default LoFTR route (kornia's download, restated), the hub-failure message -- and the lifetime of `match_threshold` on a cached
model (imcui/ui/utils.py:921-922 mutates conf after construction; the reference's networks never see it).

Reference: imcui/hloc/utils/base_model.py:37-43 (`_download_model`), matchers/loftr.py:21-38, eloftr.py:49-61, superglue.py:32-37,
lightglue.py:37-52, duster.py:31-37, mast3r.py:36-41, extractors/superpoint.py:45-54."""
import argparse
import os

import pytest
import torch

from imcui_hip import synth_weights as W
from imcui_hip.hloc.utils import weights as weights_mod
from imcui_hip.hloc.utils.weights import load_checkpoint_file, resolve_state_dict, unwrap_checkpoint

SMALL_DUST3R = {"enc_dim": 512, "enc_depth": 2, "dec_dim": 256, "dec_depth": 4}


def _save(tmp_path, name, obj):
    p = os.path.join(str(tmp_path), name)
    torch.save(obj, p)
    return p


def _no_network(monkeypatch):
    """Any attempt to go on-line fails the way it does on an air-gapped box."""
    import huggingface_hub

    def refuse(*a, **k):
        raise ConnectionError("offline test: no hub")

    monkeypatch.setattr(huggingface_hub, "hf_hub_download", refuse)
    monkeypatch.setattr(torch.hub, "load_state_dict_from_url", refuse)


def test_unwrap_checkpoint_containers():
    sd = {"a.weight": torch.ones(2), "model.weight": torch.zeros(1)}  # a parameter NAMED model.* is not a container
    assert unwrap_checkpoint(sd) is sd
    assert unwrap_checkpoint({"state_dict": sd, "epoch": 3}) is sd
    assert unwrap_checkpoint({"model": sd, "args": argparse.Namespace(x=1)}) is sd
    assert unwrap_checkpoint({"model": {"state_dict": sd}}) is sd
    for bad in ({}, {"epoch": 3}, [1, 2], {"state_dict": {"w": "not a tensor"}}):
        with pytest.raises(TypeError, match="no state dict"):
            unwrap_checkpoint(bad)


def test_missing_weights_path_is_an_error_not_a_download(tmp_path, monkeypatch):
    _no_network(monkeypatch)
    with pytest.raises(FileNotFoundError, match="nowhere.pth"):
        resolve_state_dict({"weights_path": str(tmp_path / "nowhere.pth"), "model_name": "superpoint_v1.pth"}, "superglue")


def test_hub_failure_message_names_the_file_and_the_two_conf_keys(monkeypatch):
    _no_network(monkeypatch)
    with pytest.raises(RuntimeError) as e:
        resolve_state_dict({"model_name": "superpoint_lightglue.pth", "weights": "outdoor"}, "lightglue")
    msg = str(e.value)
    assert "lightglue/superpoint_lightglue.pth" in msg and "weights_path" in msg and "state_dict" in msg and "offline test" in msg


def test_superpoint_superglue_lightglue_files(lib, tmp_path):
    """Bare `.pth` state dicts (superpoint_v1.pth, superglue_outdoor.pth, superpoint_lightglue.pth) and LightGlue's pre-rename block
    names `self_attn.{i}` / `cross_attn.{i}` (renamed on load upstream)."""
    from imcui_hip.hloc.extractors.superpoint import SuperPoint
    from imcui_hip.hloc.matchers.lightglue import LightGlue
    from imcui_hip.hloc.matchers.superglue import SuperGlue

    for cls, sd, name in ((SuperPoint, W.superpoint_state_dict(0), "superpoint_v1.pth"), (SuperGlue, W.superglue_state_dict(0), "superglue_outdoor.pth"),
                          (LightGlue, W.lightglue_state_dict(0), "superpoint_lightglue.pth")):  # fmt: skip
        want = cls({"state_dict": sd}).packed
        got = cls({"weights_path": _save(tmp_path, name, sd)})
        assert torch.equal(got.packed, want), name
        assert "state_dict" not in got.conf
    sd = W.lightglue_state_dict(0)
    old = {}
    for k, v in sd.items():
        parts = k.split(".")
        if parts[0] == "transformers" and parts[2] in ("self_attn", "cross_attn"):
            k = ".".join([parts[2], parts[1], *parts[3:]])
        old[k] = v
    assert any(k.startswith("self_attn.0.") for k in old) and not any(k.startswith("transformers.0.self_attn") for k in old)
    got = LightGlue({"weights_path": _save(tmp_path, "old_names.pth", old)})
    assert torch.equal(got.packed, LightGlue({"state_dict": sd}).packed)
    # the reference's wrapper leaves the downloaded path in conf["weights"] (lightglue.py:48): an existing file there is honoured
    got = LightGlue({"weights": _save(tmp_path, "via_weights_key.pth", sd)})
    assert torch.equal(got.packed, LightGlue({"state_dict": sd}).packed)


def test_loftr_and_eloftr_lightning_checkpoints(lib, tmp_path):
    """`.ckpt` = {"state_dict": ..., + non-tensor Lightning fields} (needs the full unpickler, as eloftr.py:54 `weights_only=False`)."""
    from imcui_hip.hloc.matchers import eloftr as E
    from imcui_hip.hloc.matchers.loftr import LoFTR

    sd = W.loftr_state_dict(0)
    want = LoFTR({"state_dict": sd}).packed
    ckpt = {"state_dict": sd, "epoch": 29, "hyper_parameters": argparse.Namespace(lr=1e-3), "pytorch-lightning_version": "1.3.5"}
    assert torch.equal(LoFTR({"weights_path": _save(tmp_path, "loftr_outdoor.ckpt", ckpt)}).packed, want)
    assert torch.equal(LoFTR({"weights_path": _save(tmp_path, "loftr_bare.ckpt", sd)}).packed, want)
    assert torch.equal(LoFTR({"state_dict": {"state_dict": sd}}).packed, want)  # the container handed over in memory

    port = W.eloftr_state_dict(0)
    want = E.ELoFTR({"state_dict": port}).packed
    up = E.port_to_upstream_names(port)  # the names of eloftr_outdoor.ckpt
    ckpt = {"state_dict": {**up, "matcher.pos_encoding.sin": torch.zeros(3)}, "epoch": 29, "hparams_name": "args", "hyper_parameters": argparse.Namespace(x=1)}
    assert torch.equal(E.ELoFTR({"weights_path": _save(tmp_path, "eloftr_outdoor.ckpt", ckpt)}).packed, want)
    assert torch.equal(E.ELoFTR({"weights_path": _save(tmp_path, "eloftr_port.pth", port)}).packed, want)


def test_dust3r_and_mast3r_model_containers(lib, tmp_path):
    """`duster_vit_large.pth` / the MASt3R file = {"args": Namespace, "model": state dict} (`from_pretrained` reads ckpt["model"])."""
    from imcui_hip.hloc.matchers.duster import Duster
    from imcui_hip.hloc.matchers.mast3r import Mast3r

    for cls, cfg, name in ((Duster, SMALL_DUST3R, "duster_vit_large.pth"), (Mast3r, {**SMALL_DUST3R, "desc_dim": 24}, "mast3r.pth")):
        sd = W.dust3r_state_dict(0, cfg)
        want = cls({"state_dict": sd})
        got = cls({"weights_path": _save(tmp_path, name, {"args": argparse.Namespace(model="AsymmetricCroCo3DStereo(...)"), "model": sd, "epoch": 0})})
        assert torch.equal(got.packed, want.packed) and got.net_cfg == want.net_cfg, name


def test_default_loftr_entry_resolves_like_kornia(lib, tmp_path, monkeypatch):
    """imcui/hloc/configs/matchers.py:249-256 (`loftr`: weights "outdoor", no model_name) -> `LoFTR_(pretrained="outdoor")` (loftr.py:37):
    kornia's URL through torch.hub.  The file kornia's download leaves in the torch-hub cache is found without a network; without it the
    error names kornia's URL and the conf keys that take a local file."""
    from imcui_hip.hloc.matchers import loftr as L

    sd = W.loftr_state_dict(0)
    want = L.LoFTR({"state_dict": sd}).packed
    monkeypatch.setenv("TORCH_HOME", str(tmp_path))
    os.makedirs(tmp_path / "hub" / "checkpoints")
    torch.save({"state_dict": sd}, tmp_path / "hub" / "checkpoints" / "loftr_outdoor.ckpt")
    import huggingface_hub

    def refuse(*a, **k):
        raise ConnectionError("offline test: no hub")

    monkeypatch.setattr(huggingface_hub, "hf_hub_download", refuse)
    assert torch.equal(L.LoFTR({}).packed, want)  # the zoo's default entry
    assert L.LoFTR({}).temp_bug_fix is False
    # indoor_new: kornia sets temp_bug_fix for its re-trained indoor weights
    torch.save({"state_dict": sd}, tmp_path / "hub" / "checkpoints" / "loftr_indoor_ds_new.ckpt")
    assert L.LoFTR({"weights": "indoor_new"}).temp_bug_fix is True
    with pytest.raises(ValueError, match="kornia's LoFTR knows"):
        L.LoFTR({"weights": "no_such_weights"})
    # nothing cached, nothing reachable: the message says what was tried and what to pass instead
    monkeypatch.setattr(torch.hub, "load_state_dict_from_url", refuse)
    with pytest.raises(RuntimeError) as e:
        L.LoFTR({"weights": "indoor"})
    assert "weights_path" in str(e.value) and "kornia_pretrained" in str(e.value)
    # a MINIMA model name goes to the model repository like the reference (loftr.py:29-33) and sets temp_bug_fix
    seen = []

    def fake_hub(repo_type, repo_id, filename):
        seen.append((repo_id, filename))
        return _save(tmp_path, "minima.ckpt", {"state_dict": sd})

    monkeypatch.setattr(huggingface_hub, "hf_hub_download", fake_hub)
    m = L.LoFTR({"model_name": "minima_loftr.ckpt"})
    assert seen == [(weights_mod.MODEL_REPO_ID, "loftr/minima_loftr.ckpt")] and m.temp_bug_fix is True and torch.equal(m.packed, want)


class _Spy:
    def __init__(self):
        self.seen = []

    def forward(self, *a, **k):
        self.seen.append(a)
        return {}


def test_match_threshold_is_frozen_at_init_for_loftr_eloftr_superglue(lib):
    """The reference bakes the threshold into the network at `_init` (loftr.py:21-24 `cfg["match_coarse"]["thr"]`, eloftr.py:51-52,
    superglue.py:37 `SG(conf)` copies the conf): mutating a cached model's conf changes nothing.  The plugins do the same by default;
    conf["runtime_match_threshold"] = True re-reads the conf per call.  default_conf stays the reference's dictionary."""
    from imcui_hip.hloc.matchers.eloftr import ELoFTR
    from imcui_hip.hloc.matchers.loftr import LoFTR
    from imcui_hip.hloc.matchers.superglue import SuperGlue

    img = torch.zeros(1, 1, 32, 32)
    for cls, sd in ((LoFTR, W.loftr_state_dict(0)), (ELoFTR, W.eloftr_state_dict(0))):
        assert "runtime_match_threshold" not in cls.default_conf
        m = cls({"match_threshold": 0.3, "state_dict": sd})
        m._impl = spy = _Spy()
        m.forward_batched(img, img)
        m.conf["match_threshold"] = 0.05  # what run_matching does to a cached matcher
        m.conf["precision"] = "fp16"  # (EfficientLoFTR picks its arithmetic in _init as well)
        m.forward_batched(img, img)
        assert [a[3] for a in spy.seen] == [0.3, 0.3], cls.__name__
        if cls is ELoFTR:
            assert [a[5] for a in spy.seen] == [0, 0]
        m.conf["runtime_match_threshold"] = True
        m.forward_batched(img, img)
        assert spy.seen[-1][3] == 0.05
    assert "runtime_match_threshold" not in SuperGlue.default_conf
    sg = SuperGlue({"match_threshold": 0.3, "sinkhorn_iterations": 7, "state_dict": W.superglue_state_dict(0)})
    sg._impl = spy = _Spy()
    z = (torch.zeros(1, 4, 2), torch.zeros(1, 4, 2), torch.zeros(1, 4), torch.zeros(1, 4), torch.zeros(1, 4, 256), torch.zeros(1, 4, 256),
         torch.zeros(1, dtype=torch.int32), torch.zeros(1, dtype=torch.int32), (64, 48), (64, 48))  # fmt: skip
    sg.forward_batched(*z)
    sg.conf["match_threshold"], sg.conf["sinkhorn_iterations"] = 0.05, 100
    sg.forward_batched(*z)
    assert [a[-2:] for a in spy.seen] == [(7, 0.3), (7, 0.3)]
    sg.conf["runtime_match_threshold"] = True
    sg.forward_batched(*z)
    assert spy.seen[-1][-2:] == (100, 0.05)
