"""Seeded synthetic weights for the oracle side: the generators live in imcui_hip/synth_weights.py (pure data
generation, shared with bench.py / smoke so that checker and HIP path load identical tensors)."""
from imcui_hip.synth_weights import (  # noqa: F401
    SP_LAYERS,
    lightglue_state_dict,
    loftr_state_dict,
    superglue_state_dict,
    superpoint_state_dict,
)
