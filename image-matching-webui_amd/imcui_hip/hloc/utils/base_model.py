"""Plugin base class of the drop-in seam.

Interface contract = imcui/hloc/utils/base_model.py:9-55: the constructor merges the given conf
over the class `default_conf` and calls `_init(conf)`; calling the module asserts that every
`required_inputs` key is present and dispatches to `_forward(data)`; `dynamic_load(root, name)`
imports `<root>.<name>` and returns the single plugin class defined there.

When the reference package is importable its own classes are re-exported, so the HIP plugins
ARE reference plugins and `dynamic_load`, the UI model cache and `.eval().to(DEVICE)` work on
them unchanged.  Without imcui (e.g. on the GPU test box) the stand-in below is used.
"""
from __future__ import annotations

import importlib

from torch import nn

try:  # pragma: no cover - only where the reference is installed
    from imcui.hloc.utils.base_model import BaseModel, dynamic_load  # type: ignore  # noqa: F401

    HAVE_REFERENCE = True
except Exception:  # noqa: BLE001
    HAVE_REFERENCE = False

    class BaseModel(nn.Module):
        default_conf: dict = {}
        required_inputs: list = []

        def __init__(self, conf):
            super().__init__()
            merged = dict(self.default_conf)
            merged.update(conf)
            self.conf = merged
            self.required_inputs = list(self.required_inputs)
            self._init(merged)

        def forward(self, data):
            missing = [k for k in self.required_inputs if k not in data]
            assert not missing, "Missing key {} in data".format(missing[0] if missing else "")
            return self._forward(data)

        def _init(self, conf):  # to be provided by the plugin
            raise NotImplementedError

        def _forward(self, data):  # to be provided by the plugin
            raise NotImplementedError

    def dynamic_load(root, model):
        """Return the one BaseModel subclass that module `<root>.<model>` itself defines."""
        path = f"{root.__name__}.{model}"
        mod = importlib.import_module(path)
        found = [
            obj
            for obj in vars(mod).values()
            if isinstance(obj, type) and obj.__module__ == path and issubclass(obj, BaseModel)
        ]
        assert len(found) == 1, found
        return found[0]
