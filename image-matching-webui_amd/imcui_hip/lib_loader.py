"""ctypes binding of libimcui_hip.so (the C ABI declared in include/imcui_hip.h).

The product path fails loudly when the library is missing: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from .build import LIB_PATH

_lock = threading.Lock()
_lib = None


class ImcuiHipError(RuntimeError):
    pass


c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)

# name -> (restype, argtypes); must list every symbol include/imcui_hip.h declares
SIGNATURES = {
    "imcui_hip_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "imcui_hip_destroy": (None, [C.c_void_p]),
    "imcui_hip_last_error": (C.c_char_p, [C.c_void_p]),
    "imcui_hip_version": (C.c_int, []),
    "imcui_hip_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "imcui_hip_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "imcui_hip_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "imcui_hip_get_precision": (C.c_int, [C.c_void_p]),
    "imcui_hip_conv3x3_pack_split": (C.c_float, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "imcui_hip_rgb_to_gray_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "imcui_hip_area_table": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "imcui_hip_preprocess_area_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p]),
    "imcui_hip_linear_table": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "imcui_hip_preprocess_linear_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p]),
    "imcui_hip_aa_table": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "imcui_hip_resize_aa_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 3 + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "imcui_hip_linear_pack_split": (C.c_float, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "imcui_hip_linear_split_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_void_p]),
    "imcui_hip_set_range_check": (C.c_int, [C.c_void_p, C.c_int]),
    "imcui_hip_get_range_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "imcui_hip_qkv_split_f32": (C.c_int, [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_float, C.c_int] + [C.c_void_p] * 4),
    "imcui_hip_ffn_pack_w2": (C.c_float, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "imcui_hip_ffn_set_debug": (C.c_int, [C.c_void_p, C.c_void_p]),
    "imcui_hip_ffn_split_f32": (C.c_int, [C.c_void_p] * 14 + [C.c_int, C.c_int, C.c_void_p]),
    "imcui_hip_conv3x3_split_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 7 + [C.c_void_p]),
    "imcui_hip_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "imcui_hip_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "imcui_hip_superpoint_packed_floats": (C.c_size_t, []),
    "imcui_hip_superpoint_pack_weights": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    "imcui_hip_superpoint_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "imcui_hip_superpoint_max_keypoints_bound": (C.c_int, [C.c_int] * 3),
    "imcui_hip_superpoint_forward": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        + [C.c_void_p] * 6
        + [C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "imcui_hip_superpoint_status": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "imcui_hip_simple_nms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "imcui_hip_lightglue_packed_floats": (C.c_size_t, []),
    "imcui_hip_lightglue_num_tensors": (C.c_int, []),
    "imcui_hip_lightglue_tensor_name": (C.c_char_p, [C.c_int]),
    "imcui_hip_lightglue_pack_weights": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p]),
    "imcui_hip_lightglue_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "imcui_hip_lightglue_pack_input": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "imcui_hip_lightglue_forward": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        + [C.c_void_p] * 10
        + [C.c_float] * 4
        + [C.c_double, C.c_double, C.c_int, C.c_double]
        + [C.c_void_p] * 7
        + [C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "imcui_hip_lightglue_set_layer_dump": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "imcui_hip_superglue_packed_floats": (C.c_size_t, []),
    "imcui_hip_superglue_num_tensors": (C.c_int, []),
    "imcui_hip_superglue_tensor_name": (C.c_char_p, [C.c_int]),
    "imcui_hip_superglue_pack_weights": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p]),
    "imcui_hip_superglue_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "imcui_hip_superglue_forward": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        + [C.c_void_p] * 8
        + [C.c_float] * 4
        + [C.c_int, C.c_double]
        + [C.c_void_p] * 4
        + [C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "imcui_hip_loftr_packed_floats": (C.c_size_t, []),
    "imcui_hip_loftr_num_layers": (C.c_int, []),
    "imcui_hip_loftr_layer_shape": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "imcui_hip_loftr_num_norms": (C.c_int, []),
    "imcui_hip_loftr_norm_dim": (C.c_int, [C.c_int]),
    "imcui_hip_loftr_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    "imcui_hip_loftr_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "imcui_hip_loftr_forward": (
        C.c_int,
        [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_double, C.c_int] + [C.c_void_p] * 5 + [C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "imcui_hip_loftr_debug_offset": (C.c_size_t, [C.c_int] * 6),
    "imcui_hip_loftr_last_fine_mode": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "imcui_hip_eloftr_packed_floats": (C.c_size_t, []),
    "imcui_hip_eloftr_num_layers": (C.c_int, []),
    "imcui_hip_eloftr_layer_shape": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "imcui_hip_eloftr_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                 C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "imcui_hip_eloftr_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "imcui_hip_eloftr_forward": (
        C.c_int,
        [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_double] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "imcui_hip_eloftr_forward_ex": (
        C.c_int,
        [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_double, C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "imcui_hip_eloftr_debug_offset": (C.c_size_t, [C.c_int] * 6),
    "imcui_hip_dust3r_packed_floats": (C.c_size_t, [C.c_int] * 5),
    "imcui_hip_dust3r_num_layers": (C.c_int, [C.c_int] * 5),
    "imcui_hip_dust3r_num_vectors": (C.c_int, [C.c_int] * 5),
    "imcui_hip_dust3r_layer_shape": (C.c_int, [C.c_int] * 6 + [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "imcui_hip_dust3r_vector_len": (C.c_int, [C.c_int] * 6),
    "imcui_hip_dust3r_layer_offsets": (C.c_int, [C.c_int] * 6 + [C.POINTER(C.c_size_t)] * 4 + [C.POINTER(C.c_int)]),
    "imcui_hip_dust3r_pack_weights": (C.c_int, [C.c_int] * 5 + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    "imcui_hip_dust3r_workspace_bytes": (C.c_size_t, [C.c_int] * 9),
    "imcui_hip_dust3r_dump_floats": (C.c_size_t, [C.c_int] * 9),
    "imcui_hip_dust3r_format_version": (C.c_int, []),
    "imcui_hip_dust3r_check_packed": (C.c_int, [C.c_int] * 5 + [C.c_void_p, C.c_size_t]),
    "imcui_hip_dust3r_forward": (
        C.c_int,
        [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "imcui_hip_dust3r_workspace_bytes_sizes": (C.c_size_t, [C.c_int] * 6 + [C.POINTER(C.c_int), C.c_int]),
    "imcui_hip_dust3r_token_dump_floats": (C.c_size_t, [C.c_int] * 6 + [C.POINTER(C.c_int), C.c_int]),
    "imcui_hip_dust3r_forward_sizes": (
        C.c_int,
        [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "imcui_hip_conv_gemm_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 8 + [C.c_void_p]),
    "imcui_hip_nn_argmax_workspace_bytes": (C.c_size_t, [C.c_int] * 2),
    "imcui_hip_nn_argmax_split_workspace_bytes": (C.c_size_t, [C.c_int] * 2),
    "imcui_hip_nn_argmax_split_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "imcui_hip_nn_argmax_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "imcui_hip_jpeg_info": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "imcui_hip_jpeg_coef_count": (C.c_size_t, [C.POINTER(C.c_int)]),
    "imcui_hip_jpeg_entropy_decode": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "imcui_hip_jpeg_entropy_decode_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "imcui_hip_orient_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "imcui_hip_jpeg_workspace_bytes": (C.c_size_t, [C.POINTER(C.c_int), C.c_int]),
    "imcui_hip_jpeg_workspace_bytes_batch": (C.c_size_t, [C.POINTER(C.c_int), C.c_int, C.c_int]),
    "imcui_hip_jpeg_reconstruct_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                   C.c_size_t, C.c_void_p]),
    "imcui_hip_jpeg_reconstruct": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "imcui_hip_png_info": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "imcui_hip_png_raw_bytes": (C.c_size_t, [C.POINTER(C.c_int)]),
    "imcui_hip_png_inflate": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "imcui_hip_png_inflate_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "imcui_hip_png_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "imcui_hip_png_reconstruct_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "imcui_hip_ransac_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "imcui_hip_ransac": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_ulonglong,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "imcui_hip_mutual_nn_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "imcui_hip_mutual_nn_workspace_bytes_for": (C.c_size_t, [C.c_void_p] + [C.c_int] * 3),
    "imcui_hip_mutual_nn_workspace_bytes_d": (C.c_size_t, [C.c_void_p] + [C.c_int] * 4),
    "imcui_hip_simred_chunks": (C.c_int, [C.c_int] * 3),
    "imcui_hip_simred_debug_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "imcui_hip_simred_debug": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 2 + [C.c_float, C.c_int] + [C.c_void_p] * 13 + [C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "imcui_hip_mutual_nn": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "imcui_hip_mutual_nn_dn_workspace_bytes_for": (C.c_size_t, [C.c_void_p] + [C.c_int] * 4),
    "imcui_hip_mutual_nn_dn": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "imcui_hip_dual_softmax_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "imcui_hip_dual_softmax": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "imcui_hip_linear_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    "imcui_hip_conv3x3_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "imcui_hip_conv3x3_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_void_p]),
    "imcui_hip_attention_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p]),
    "imcui_hip_attention_mx_scratch_bytes": (C.c_size_t, [C.c_int] * 3),
    "imcui_hip_attention_mx_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p, C.c_size_t, C.c_void_p]),
}


def load_library(path: str | None = None):
    """dlopen the HIP backend; raises ImcuiHipError when it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = path or os.environ.get("IMCUI_HIP_LIB", LIB_PATH)
        if not os.path.exists(path):
            raise ImcuiHipError(
                f"{path} not found: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        try:
            lib = C.CDLL(path)
        except OSError as e:  # pragma: no cover
            raise ImcuiHipError(f"cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise ImcuiHipError(f"{path} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib
