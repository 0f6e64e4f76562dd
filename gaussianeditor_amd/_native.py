"""ctypes binding of libgsr_hip.so (the C ABI of include/gsr.h).

There is NO fallback: if the library is missing or fails to load, importing the
rasterizer raises.  ctypes releases the GIL for the duration of every native call
(the reference's pybind functions hold it, ext.cpp:16-19).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_uint, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
#: GSR_LIBRARY_PATH (development only: A/B timing of differently built libraries, tools/ab_variants.py) replaces the
#: in-tree library; it must export the same ABI and is loaded under the same "no fallback" rule.
LIB_PATH = os.environ.get("GSR_LIBRARY_PATH") or os.path.join(_HERE, "libgsr_hip.so")
GSR_ABI_VERSION = 6

_P = c_void_p
#: floats per row of the blend backward's accumulator table and its columns (include/gsr.h: GSR_ACC_*)
ACC_ROW, ACC_MEAN2D, ACC_OPACITY, ACC_CONIC, ACC_COLOR = 16, 0, 3, 4, 8


class AdamTensor(ctypes.Structure):
    """gsr_adam_tensor (include/gsr.h)."""
    _fields_ = [("param", _P), ("grad", _P), ("exp_avg", _P), ("exp_avg_sq", _P), ("anchor", _P), ("numel", c_int64),
                ("row_len", ctypes.c_int32), ("masked", ctypes.c_int32), ("lr", ctypes.c_double), ("anchor_scale", c_float)]


class DenseGrads(ctypes.Structure):
    """gsr_dense_grads (include/gsr.h)."""
    _fields_ = [("means3D", _P), ("scales", _P), ("rotations", _P), ("means2D", _P), ("opacities", _P), ("sh", _P)]


class AppendTensor(ctypes.Structure):
    """gsr_append_tensor (include/gsr.h)."""
    _fields_ = [("src", _P), ("ext", _P), ("dst", _P), ("row_bytes", c_int64)]


class CompactTensor(ctypes.Structure):
    """gsr_compact_tensor (include/gsr.h)."""
    _fields_ = [("src", _P), ("dst", _P), ("row_bytes", c_int64)]


#: every symbol include/gsr.h declares, with (restype, argtypes)
SIGNATURES = {
    "gsr_abi_version": (c_int, []),
    "gsr_status_string": (ctypes.c_char_p, [c_int]),
    "gsr_last_hip_error": (c_int, []),
    "gsr_scratch_sizes": (c_int, [c_int, c_int64, c_int64, c_int, c_int, POINTER(c_size_t)]),
    "gsr_sort_key_bits": (c_int, [c_int, c_int]),
    "gsr_preprocess": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int,
                               c_float, c_float, c_int, c_int, c_uint, _P, _P, POINTER(c_int64)]),
    "gsr_preprocess_begin": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int,
                                     c_float, c_float, c_int, c_int, c_uint, _P, _P, POINTER(_P)]),
    "gsr_preprocess_end": (c_int, [_P, c_int, c_int, c_int, _P, _P, POINTER(c_int64)]),
    "gsr_arrays_equal": (c_int, [_P, c_int, POINTER(_P), POINTER(_P), POINTER(c_size_t), POINTER(c_int)]),
    "gsr_bin": (c_int, [_P, c_int, c_int64, c_int64, c_int, c_int, _P, _P, _P]),
    "gsr_blend_forward": (c_int, [_P, c_int, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P, c_uint]),
    "gsr_blend_forward_aux": (c_int, [_P, c_int, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_uint]),
    "gsr_backward": (c_int, [_P, c_int, c_int, c_int, c_int64, c_int, c_int, _P, _P, _P, _P, _P, c_float, _P, _P, _P,
                             _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_uint]),
    # (stream, P, R, W, H, bg, geom, binning, image, dL_dpix, acc, touched, flags)
    "gsr_blend_backward": (c_int, [_P, c_int, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_uint]),
    # (... radii, geom, acc, dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drots, flags)
    "gsr_preprocess_backward": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_float, _P, _P, _P, _P, _P,
                                        c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_uint]),
    # (... radii, geom, acc, dL_dmeans2D, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_drgb, dL_dscales, dL_drots)
    "gsr_preprocess_backward_rgb": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_float, _P, _P, _P, _P, _P,
                                            c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_uint]),
    # (... radii, geom, acc, dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_drgb, dL_dscales, dL_drots, row_state)
    "gsr_preprocess_backward_rows": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_float, _P, _P, _P, _P, _P,
                                            c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gsr_sh_grad_compose": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "gsr_view_message_words": (c_int, [c_int64, c_int64, POINTER(c_int64)]),
    "gsr_view_message_plan": (c_int, [_P, c_int64, POINTER(DenseGrads), _P, _P, _P, POINTER(c_int64)]),
    "gsr_view_message_plan_blend": (c_int, [_P, c_int64, _P, _P]),
    "gsr_view_message_pack": (c_int, [_P, c_int64, POINTER(DenseGrads), _P, _P, _P, _P, c_int64, _P]),
    "gsr_view_messages_accumulate": (c_int, [_P, c_int64, c_int, c_int, c_int, _P, c_int64, c_int64, _P, POINTER(DenseGrads)]),
    "gsr_view_messages_accumulate_rows": (c_int, [_P, c_int64, c_int, c_int, c_int, _P, c_int64, c_int64, _P, POINTER(DenseGrads), _P]),
    "gsr_knn_workspace_size": (c_int, [c_int, POINTER(c_size_t)]),
    "gsr_knn_mean_dist2": (c_int, [_P, c_int, _P, _P, _P]),
    "gsr_near_workspace_size": (c_int, [c_int, POINTER(c_size_t)]),
    "gsr_near_points": (c_int, [_P, c_int, _P, c_int, _P, c_float, _P, _P, _P]),
    "gsr_compact_workspace_size": (c_int, [c_int64, POINTER(c_size_t)]),
    "gsr_compact_plan": (c_int, [_P, c_int64, _P, _P, POINTER(c_int64)]),
    "gsr_compact_apply": (c_int, [_P, c_int64, _P, _P, c_int, POINTER(CompactTensor)]),
    "gsr_append_rows": (c_int, [_P, c_int64, c_int64, c_int, POINTER(AppendTensor)]),
    "gsr_adam_step": (c_int, [_P, c_int, POINTER(AdamTensor), c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_double, _P,
                            _P]),
    "gsr_adam_step_rows": (c_int, [_P, c_int, POINTER(AdamTensor), c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_double, _P,
                            _P, _P]),
    "gsr_mark_visible": (c_int, [_P, c_int, _P, _P, _P, _P]),
    "gsr_trace_weights": (c_int, [_P, c_int, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_uint]),
    "gsr_debug_export_geom": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "gsr_debug_cov3d": (c_int, [_P, c_int, _P, c_float, _P, _P]),
    "gsr_debug_export_binning": (c_int, [_P, c_int, c_int64, c_int, c_int, _P, _P, _P, _P]),
    "gsr_debug_blend_backward_profile": (c_int, [_P, c_int, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P, _P,
                                                 c_int64, POINTER(c_int64)]),
    "gsr_debug_blend_forward_profile": (c_int, [_P, c_int, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int64,
                                                POINTER(c_int64)]),
    "gsr_debug_export_image": (c_int, [_P, c_int, c_int, _P, _P, _P, _P]),
}

_lib = None


class GsrError(RuntimeError):
    """A native call returned a non-zero gsr_status."""

    def __init__(self, fn: str, status: int, message: str, hip_error: int = 0):
        self.status = status
        self.hip_error = hip_error
        extra = f" (hipError_t {hip_error})" if hip_error else ""
        super().__init__(f"{fn} failed: {message} [status {status}]{extra}")


def lib() -> ctypes.CDLL:
    """Load (once) and type the native library.  Raises if it is absent: build it with
    `python -c 'import __graft_entry__ as g; g.build()'` or `make -C gaussianeditor_amd/csrc`."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU or PyTorch fallback; "
                "run `make -C gaussianeditor_amd/csrc` (hipcc --offload-arch=gfx950).")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        if L.gsr_abi_version() != GSR_ABI_VERSION:
            raise ImportError(f"{LIB_PATH}: ABI version {L.gsr_abi_version()} != expected {GSR_ABI_VERSION}")
        _lib = L
    return _lib


def check(fn: str, status: int) -> None:
    if status != 0:
        L = lib()
        msg = L.gsr_status_string(status).decode()
        raise GsrError(fn, status, msg, L.gsr_last_hip_error() if status == -4 else 0)


def scratch_sizes(P: int, R: int, W: int, H: int, G: int = 0):
    """(geometry, binning, image) scratch bytes; R, G = the two counts gsr_preprocess returns (0, 0 before they are known)."""
    sizes = (c_size_t * 3)()
    check("gsr_scratch_sizes", lib().gsr_scratch_sizes(P, R, G, W, H, sizes))
    return int(sizes[0]), int(sizes[1]), int(sizes[2])
