"""Weight resolution for the HIP plugins.

The reference downloads checkpoints from the HF hub in `_init`
(imcui/hloc/utils/base_model.py:37-43, repo `Realcat/imcui_checkpoints`) and hands the file to the upstream
constructor, which unwraps whatever container the authors saved.  Resolution order here:

1. conf["state_dict"]: an in-memory dict (the parity tests; a maintainer who already holds the tensors);
2. conf["weights_path"]: a local checkpoint file (a path that does not exist is an error, not a reason to go on-line);
   conf["weights"] when it names an existing file (the reference's LightGlue wrapper stores its path there,
   imcui/hloc/matchers/lightglue.py:48);
3. `fallback()` when the caller has one (LoFTR: kornia's own download, see matchers/loftr.py);
4. hf_hub_download(MODEL_REPO_ID, "<subdir>/<model_name>"): the reference's route.

Container formats (`unwrap_checkpoint`): a bare state dict (SuperPoint, SuperGlue, LightGlue `.pth`); a Lightning
checkpoint `{"state_dict": ..., "epoch": ...}` (LoFTR / EfficientLoFTR `.ckpt`: `torch.load(...)["state_dict"]`,
imcui/hloc/matchers/loftr.py:33, eloftr.py:54-56); `{"model": ..., "args": ...}` (DUSt3R / MASt3R `.pth`:
`AsymmetricCroCo3DStereo.from_pretrained` reads `ckpt["model"]`).  Tested without a GPU: tests/test_weights_cpu.py.
"""
from __future__ import annotations

import os
import pickle

import torch

MODEL_REPO_ID = "Realcat/imcui_checkpoints"  # imcui/hloc/__init__.py:66


def unwrap_checkpoint(obj):
    """The tensor dictionary inside whatever `torch.load` returned (see the module docstring); refuses anything else."""
    for _ in range(2):  # a container in a container is the deepest the upstream projects go
        if not isinstance(obj, dict):
            break
        inner = next((obj[k] for k in ("state_dict", "model") if isinstance(obj.get(k), dict)), None)
        if inner is None or any(torch.is_tensor(v) for v in obj.values()):
            break  # tensors at this level: this IS the state dict (a parameter may be called "model...." but is not a dict)
        obj = inner
    if not isinstance(obj, dict) or not obj or not all(torch.is_tensor(v) for v in obj.values()):
        kind = type(obj).__name__ if not isinstance(obj, dict) else f"dict with keys {list(obj)[:4]}"
        raise TypeError(f"checkpoint holds no state dict (got {kind}); expected tensors, or a 'state_dict' / 'model' container of tensors")
    return obj


def load_checkpoint_file(path: str) -> dict:
    """`torch.load` as the reference's wrappers call it, container unwrapped.  Tensor-only files load under `weights_only=True`;
    a Lightning / DUSt3R container that pickles other objects (hyper-parameters, an argparse namespace) needs the full unpickler,
    which the reference uses for exactly these files (`weights_only=False`, imcui/hloc/matchers/eloftr.py:54) -- a LOCAL file the
    user chose to load."""
    path = str(path)
    if not os.path.isfile(path):
        raise FileNotFoundError(f"checkpoint file not found: {path}")
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except (pickle.UnpicklingError, RuntimeError, AttributeError):
        obj = torch.load(path, map_location="cpu", weights_only=False)
    return unwrap_checkpoint(obj)


def resolve_state_dict(conf: dict, subdir: str, fallback=None) -> dict:
    sd = conf.get("state_dict")
    if sd is not None:
        return unwrap_checkpoint(sd)
    path = conf.get("weights_path")
    if path:
        return load_checkpoint_file(path)  # a wrong explicit path must not silently become a download of something else
    path = conf.get("weights")
    if path and os.path.isfile(str(path)):
        return load_checkpoint_file(path)
    errors = []
    if fallback is not None:
        try:
            got = fallback()
            if got is not None:
                return unwrap_checkpoint(got)
        except Exception as e:  # noqa: BLE001
            errors.append(f"{getattr(fallback, '__name__', 'fallback')}: {e}")
    try:
        from huggingface_hub import hf_hub_download

        path = hf_hub_download(repo_type="model", repo_id=MODEL_REPO_ID, filename=f"{subdir}/{conf['model_name']}")
    except Exception as e:  # noqa: BLE001
        errors.append(f"hub download of {MODEL_REPO_ID}:{subdir}/{conf.get('model_name')}: {type(e).__name__}: {e}")
        raise RuntimeError(
            f"no weights for {subdir}/{conf.get('model_name')}: pass conf['weights_path'] (a local checkpoint file) or conf['state_dict'] "
            f"({'; '.join(errors)})"
        ) from e
    return load_checkpoint_file(path)
