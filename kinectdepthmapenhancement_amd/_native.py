"""ctypes binding of libkde_hip.so (include/kde_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, this module
raises.  Nothing here imports, links or calls the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libkde_hip.so")
CSRC = os.path.join(_PKG, "csrc")

KDE_OK, KDE_ERR_INVALID, KDE_ERR_HIP, KDE_ERR_NOMEM, KDE_ERR_UNSUPPORTED = range(5)


class KdeError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libkde_hip error {code}: {message}")
        self.code = code


class JbfParams(C.Structure):
    """kde_jbf_params (defaults = JointBilateralFilter.cpp:3-6 and the call at JointBilateralFilter.cu:285)."""
    _fields_ = [("window_size", C.c_int), ("spatial_sigma", C.c_float), ("color_sigma", C.c_float),
                ("depth_sigma", C.c_float), ("presmooth", C.c_int), ("presmooth_kernel_size", C.c_int),
                ("presmooth_sigma_color", C.c_float), ("presmooth_sigma_spatial", C.c_float)]


def build(force: bool = False) -> str:
    """Compile libkde_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean", "-s"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    return LIB_PATH


_lib = None

# name -> (restype, argtypes); every symbol include/kde_hip.h declares
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_pp = C.POINTER(C.c_void_p)
SIGNATURES = {
    "kde_abi_version": (_i, []),
    "kde_last_error_string": (C.c_char_p, []),
    "kde_device_count": (_i, [C.POINTER(_i)]),
    "kde_set_device": (_i, [_i]),
    "kde_device_info": (_i, [C.c_char_p, _sz, C.POINTER(_i)]),
    "kde_device_pci_bus_id": (_i, [C.c_char_p, _sz]),
    "kde_jbf_default_params": (_i, [C.POINTER(JbfParams)]),
    "kde_jbf_create": (_i, [_pp, _i, _i, _i, C.POINTER(JbfParams)]),
    "kde_jbf_destroy": (_i, [_vp]),
    "kde_jbf_process": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "kde_jbf_process_batch": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "kde_jbf_presmooth_batch": (_i, [_vp, _i, _vp, _vp, _vp]),
    "kde_jbf_filter_batch": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "kde_jbf_filtered_device": (_i, [_vp, _pp]),
    "kde_jbf_filtered_host": (_i, [_vp, _vp, _pp]),
    "kde_jbf_smooth_device": (_i, [_vp, _pp]),
    "kde_jbf_spatial_table": (_i, [_vp, _vp, _i]),
    "kde_jbf_set_variant": (_i, [_vp, _i]),
    "kde_jbf_active_variant": (_i, [_vp, C.POINTER(_i)]),
    "kde_jbf_variant_count": (_i, []),
    "kde_jbf_variant_name": (C.c_char_p, [_i]),
    "kde_mrf_create": (_i, [_pp, _i, _i, _i, _i, _f, _f]),
    "kde_mrf_destroy": (_i, [_vp]),
    "kde_mrf_process_batch": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "kde_mrf_filtered_device": (_i, [_vp, _pp]),
    "kde_mrf_filtered_host": (_i, [_vp, _vp, _pp]),
    "kde_dimconv_create": (_i, [_pp]),
    "kde_dimconv_destroy": (_i, [_vp]),
    "kde_dimconv_set_camera": (_i, [_vp, _vp, _i, _i]),
    "kde_dimconv_projective_to_real_depth": (_i, [_vp, _i, _vp, _vp, _vp]),
    "kde_dimconv_projective_to_real_points": (_i, [_vp, _i, _vp, _vp, _vp]),
    "kde_dimconv_projective_to_real_interp": (_i, [_vp, _i, _vp, _vp, _vp]),
    "kde_dimconv_real_to_projective": (_i, [_vp, _i, _vp, _vp, _vp]),
    "kde_buffer2d_create": (_i, [_pp, _i, _i]),
    "kde_buffer2d_destroy": (_i, [_vp]),
    "kde_buffer2d_insert_depth": (_i, [_vp, _vp, _vp]),
    "kde_buffer2d_insert_float2": (_i, [_vp, _vp, _vp]),
    "kde_buffer2d_insert_weighted": (_i, [_vp, _vp, _vp]),
    "kde_buffer2d_get_depth_map": (_i, [_vp, _vp, _vp]),
    "kde_buffer2d_get_weight_map": (_i, [_vp, _vp, _vp]),
    "kde_buffer2d_update": (_i, [_vp, _vp, _vp]),
    "kde_buffer2d_update_sequence": (_i, [_vp, _i, _vp, _vp]),
    "kde_buffer2d_raw_pointer": (_i, [_vp, _pp]),
    "kde_dasp_create": (_i, [_pp, _i, _i]),
    "kde_dasp_destroy": (_i, [_vp]),
    "kde_dasp_set_parameters": (_i, [_vp, _i, _i, _vp]),
    "kde_dasp_segmentation": (_i, [_vp, _vp, _vp, _f, _f, _f, _i, _vp]),
    "kde_dasp_labels_device": (_i, [_vp, _pp]),
    "kde_dasp_mean_device": (_i, [_vp, _pp]),
    "kde_dasp_centers_device": (_i, [_vp, _pp]),
    "kde_dasp_ld_device": (_i, [_vp, _pp]),
    "kde_dasp_mean_host": (_i, [_vp, _vp, _pp, C.POINTER(_i)]),
    "kde_dasp_labels_host": (_i, [_vp, _vp, _pp]),
    "kde_ers_create": (_i, [_pp, _i, _i]),
    "kde_ers_destroy": (_i, [_vp]),
    "kde_ers_edge_refining": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "kde_ers_set_variant": (_i, [_vp, _i]),
    "kde_ers_stage_edge_depth_device": (_i, [_vp, _pp]),
    "kde_ers_refined_labels_device": (_i, [_vp, _pp]),
    "kde_ers_refined_depth_device": (_i, [_vp, _pp]),
    "kde_ers_refined_labels_host": (_i, [_vp, _vp, _pp]),
    "kde_ers_refined_depth_host": (_i, [_vp, _vp, _pp]),
    "kde_rgbf_create": (_i, [_pp, _i, _i]),
    "kde_rgbf_destroy": (_i, [_vp]),
    "kde_rgbf_set_parameters": (_i, [_vp, _i, _i, _vp]),
    "kde_rgbf_process": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "kde_rgbf_create_batch": (_i, [_pp, _i, _i, _i]),
    "kde_rgbf_process_batch": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "kde_rgbf_refined_depth_device": (_i, [_vp, _pp]),
    "kde_rgbf_refined_depth_host": (_i, [_vp, _vp, _pp]),
    "kde_rgbf_refined_labels_device": (_i, [_vp, _pp]),
    "kde_rgbf_sp_labels_device": (_i, [_vp, _pp]),
    "kde_rgbf_dasp_labels_device": (_i, [_vp, _pp]),
    "kde_spdsr_create": (_i, [_pp, _i, _i]),
    "kde_spdsr_destroy": (_i, [_vp]),
    "kde_spdsr_set_parameters": (_i, [_vp, _i, _i, _vp]),
    "kde_spdsr_process": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "kde_spdsr_create_batch": (_i, [_pp, _i, _i, _i]),
    "kde_spdsr_process_batch": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "kde_spdsr_refined_depth_device": (_i, [_vp, _pp]),
    "kde_spdsr_refined_depth_host": (_i, [_vp, _vp, _pp]),
    "kde_spdsr_refined_labels_device": (_i, [_vp, _pp]),
    "kde_spdsr_edge_enhanced_points_device": (_i, [_vp, _pp]),
    "kde_spdsr_optimized_points_device": (_i, [_vp, _pp]),
    "kde_spdsr_optimized_points_host": (_i, [_vp, _vp, _pp]),
    "kde_spdsr_plane_fitted_points_device": (_i, [_vp, _pp]),
    "kde_spdsr_cluster_nd_device": (_i, [_vp, _pp]),
}


def use_library(path: str) -> None:
    """Bind to another build of the same sources instead of libkde_hip.so -- the measurement builds under tools/hooks/
    (libkde_hip_ab.so: A/B switches).  Must be called before the first lib(); the product never calls it."""
    global LIB_PATH
    if _lib is not None and os.path.abspath(path) != os.path.abspath(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is already loaded")
    LIB_PATH = os.path.abspath(path)


def lib() -> C.CDLL:
    """Load libkde_hip.so; raises if it is not built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
                "There is no CPU fallback for the product path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)   # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != KDE_OK:
        raise KdeError(rc, lib().kde_last_error_string().decode("utf-8", "replace"))
