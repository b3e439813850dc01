"""Weight resolution for the HIP plugins.

The reference downloads checkpoints from the HF hub in `_init`
(imcui/hloc/utils/base_model.py:37-43, repo `Realcat/imcui_checkpoints`).  Resolution order here:
conf["state_dict"] (an in-memory dict, used by the parity tests) -> conf["weights_path"] /
conf["weights"] (a local .pth) -> hf_hub_download(repo, "<subdir>/<model_name>").
"""
from __future__ import annotations

import os

import torch

MODEL_REPO_ID = "Realcat/imcui_checkpoints"  # imcui/hloc/__init__.py:66


def resolve_state_dict(conf: dict, subdir: str) -> dict:
    sd = conf.get("state_dict")
    if sd is not None:
        return sd
    path = conf.get("weights_path") or conf.get("weights")
    if not path or not os.path.exists(str(path)):
        try:
            from huggingface_hub import hf_hub_download

            path = hf_hub_download(repo_type="model", repo_id=MODEL_REPO_ID, filename=f"{subdir}/{conf['model_name']}")
        except Exception as e:  # noqa: BLE001
            raise RuntimeError(
                f"no weights for {subdir}/{conf.get('model_name')}: pass conf['state_dict'] or conf['weights_path'] "
                f"(hub download failed: {e})"
            ) from e
    sd = torch.load(str(path), map_location="cpu")
    if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]
    return sd
