"""imcui_hip: MI355X (gfx950) native extract/match backend behind the imcui.hloc plugin API.

Host side only: ctypes bindings over the C ABI in include/imcui_hip.h, weight packing and the
BaseModel plugins (imcui_hip.hloc.*).  PyTorch is used for device memory and streams.
"""
from .lib_loader import LIB_PATH, ImcuiHipError, load_library  # noqa: F401

__version__ = "0.1.0"
