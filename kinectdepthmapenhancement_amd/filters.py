"""Python mirror of the reference's class surface for the hot path, over the C ABI.

Same class and member names as the reference (JointBilateralFilter::Process, getFiltered_Device,
RegionGrowingBilateralFilter::SetParametor [sic], ...); cv::gpu::GpuMat becomes a CUDA uint8 tensor
[H,W,3] (packed BGR), float*/float3*/int* device pointers become CUDA tensors, cv::Mat_<double>
becomes a 3x3 array.  torch is used for device memory and streams only; every computation goes
through libkde_hip.so and raises if that fails.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _native
from ._native import JbfParams, check, lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, shape, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name}: expected a CUDA tensor")
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous {dtype} {tuple(shape)}, got {t.dtype} {tuple(t.shape)}")
    return t


class _DeviceView:
    """Zero-copy torch view of an object-owned device buffer (valid while `owner` lives)."""

    def __init__(self, ptr: int, shape, typestr: str, owner):
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}


def _view(ptr: int, shape, dtype: torch.dtype, owner) -> torch.Tensor:
    typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
    holder = _DeviceView(ptr, shape, typestr, owner)
    t = torch.as_tensor(holder, device="cuda")
    t._kde_owner = holder   # keep the handle alive as long as the view
    return t


def _K9(K) -> np.ndarray:
    k = np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))
    return k


class _Handle:
    _destroy = None

    def __init__(self):
        self._h = C.c_void_p()
        self._owner_lib = lib()     # a handle is destroyed by the library that created it

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            getattr(self._owner_lib, self._destroy)(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class JointBilateralFilter(_Handle):
    """JointBilateralFilter/JointBilateralFilter.h:9-36."""
    _destroy = "kde_jbf_destroy"

    def __init__(self, width: int, height: int, params: Optional[JbfParams] = None, max_batch: int = 1):
        super().__init__()
        self.Width, self.Height, self.max_batch = width, height, max_batch
        self.params = params if params is not None else self.default_params()
        check(lib().kde_jbf_create(C.byref(self._h), width, height, max_batch, C.byref(self.params)))

    @staticmethod
    def default_params() -> JbfParams:
        p = JbfParams()
        check(lib().kde_jbf_default_params(C.byref(p)))
        return p

    # void Process(float* depth_device, cv::gpu::GpuMat color_image)
    def Process(self, depth_device: torch.Tensor, color_image: torch.Tensor) -> None:
        _req(depth_device, torch.float32, (self.Height, self.Width), "depth_device")
        _req(color_image, torch.uint8, (self.Height, self.Width, 3), "color_image")
        check(lib().kde_jbf_process(self._h, depth_device.data_ptr(), color_image.data_ptr(),
                                    self.Width * 3, _stream()))

    def process_batch(self, depth: torch.Tensor, color: torch.Tensor, out: Optional[torch.Tensor] = None):
        n = depth.shape[0]
        _req(depth, torch.float32, (n, self.Height, self.Width), "depth")
        _req(color, torch.uint8, (n, self.Height, self.Width, 3), "color")
        if out is not None:
            _req(out, torch.float32, (n, self.Height, self.Width), "out")
        check(lib().kde_jbf_process_batch(self._h, n, depth.data_ptr(), color.data_ptr(), _ptr(out), _stream()))
        return out if out is not None else self.getFiltered_Device(n)

    def presmooth_batch(self, color: torch.Tensor, out: torch.Tensor):
        n = color.shape[0]
        _req(color, torch.uint8, (n, self.Height, self.Width, 3), "color")
        _req(out, torch.uint8, (n, self.Height, self.Width, 3), "out")
        check(lib().kde_jbf_presmooth_batch(self._h, n, color.data_ptr(), out.data_ptr(), _stream()))
        return out

    def filter_batch(self, depth: torch.Tensor, guide: torch.Tensor, out: torch.Tensor):
        n = depth.shape[0]
        _req(depth, torch.float32, (n, self.Height, self.Width), "depth")
        _req(guide, torch.uint8, (n, self.Height, self.Width, 3), "guide")
        _req(out, torch.float32, (n, self.Height, self.Width), "out")
        check(lib().kde_jbf_filter_batch(self._h, n, depth.data_ptr(), guide.data_ptr(), out.data_ptr(), _stream()))
        return out

    def getFiltered_Device(self, n: int = 1) -> torch.Tensor:
        p = C.c_void_p()
        check(lib().kde_jbf_filtered_device(self._h, C.byref(p)))
        shape = (self.Height, self.Width) if n == 1 else (n, self.Height, self.Width)
        return _view(p.value, shape, torch.float32, self)

    def getFiltered_Host(self) -> np.ndarray:
        p = C.c_void_p()
        check(lib().kde_jbf_filtered_host(self._h, _stream(), C.byref(p)))
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(self.Height, self.Width))
        return arr.copy()

    def getSmoothImage_Device(self, n: int = 1) -> torch.Tensor:
        p = C.c_void_p()
        check(lib().kde_jbf_smooth_device(self._h, C.byref(p)))
        shape = (self.Height, self.Width, 3) if n == 1 else (n, self.Height, self.Width, 3)
        return _view(p.value, shape, torch.uint8, self)

    def spatial_table(self) -> np.ndarray:
        w = self.params.window_size
        t = np.empty((w, w), np.float32)
        check(lib().kde_jbf_spatial_table(self._h, t.ctypes.data, w * w))
        return t

    def set_variant(self, v: int) -> None:
        check(lib().kde_jbf_set_variant(self._h, v))

    def active_variant(self) -> str:
        """name of the kernel this handle's parameters and variant setting select ("generic-32x8-1px" when no tuned one applies)"""
        v = C.c_int(-1)
        check(lib().kde_jbf_active_variant(self._h, C.byref(v)))
        return lib().kde_jbf_variant_name(v.value).decode()

    @staticmethod
    def variants():
        n = lib().kde_jbf_variant_count()
        return [lib().kde_jbf_variant_name(i).decode() for i in range(n)]


class MarkovRandomField(_Handle):
    """MarkovRandomField/MarkovRandomField.h (sibling filter, same signature as JBF)."""
    _destroy = "kde_mrf_destroy"

    def __init__(self, width: int, height: int, max_batch: int = 1, window: int = 0,
                 color_sigma: float = -1.0, smooth_sigma: float = -1.0):
        super().__init__()
        self.Width, self.Height, self.max_batch = width, height, max_batch
        check(lib().kde_mrf_create(C.byref(self._h), width, height, max_batch, window, color_sigma, smooth_sigma))

    def Process(self, depth_device: torch.Tensor, color_image: torch.Tensor) -> None:
        _req(depth_device, torch.float32, (self.Height, self.Width), "depth_device")
        _req(color_image, torch.uint8, (self.Height, self.Width, 3), "color_image")
        check(lib().kde_mrf_process_batch(self._h, 1, depth_device.data_ptr(), color_image.data_ptr(), None, _stream()))

    def process_batch(self, depth, color, out):
        n = depth.shape[0]
        check(lib().kde_mrf_process_batch(self._h, n, depth.data_ptr(), color.data_ptr(), out.data_ptr(), _stream()))
        return out

    def getFiltered_Device(self) -> torch.Tensor:
        p = C.c_void_p()
        check(lib().kde_mrf_filtered_device(self._h, C.byref(p)))
        return _view(p.value, (self.Height, self.Width), torch.float32, self)

    def getFiltered_Host(self) -> np.ndarray:
        """MarkovRandomField.h:16"""
        p = C.c_void_p()
        check(lib().kde_mrf_filtered_host(self._h, _stream(), C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(self.Height, self.Width)).copy()


class DimensionConvertor(_Handle):
    """DimensionConvertor/DimensionConvertor.h:152-171.  float3 buffers are float32 tensors [..., 3]."""
    _destroy = "kde_dimconv_destroy"

    def __init__(self):
        super().__init__()
        check(lib().kde_dimconv_create(C.byref(self._h)))
        self.Width = self.Height = 0

    def setCameraParameters(self, intrinsic, width: int, height: int) -> None:
        k = _K9(intrinsic)
        check(lib().kde_dimconv_set_camera(self._h, k.ctypes.data, width, height))
        self.Width, self.Height = width, height

    def _n(self, t: torch.Tensor, trailing) -> int:
        lead = t.shape[:t.dim() - len(trailing) - 2]
        n = int(np.prod(lead)) if len(lead) else 1
        exp = tuple(lead) + (self.Height, self.Width) + tuple(trailing)
        if tuple(t.shape) != exp or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError(f"expected contiguous CUDA float32 {exp}, got {t.dtype} {tuple(t.shape)}")
        return n

    def projectiveToReal(self, data: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """both overloads: float* depth -> float3* (DimensionConvertor.cu:3-23) or float3* -> float3* (:25-33)."""
        if data.shape[-1] == 3 and data.dim() >= 3 and tuple(data.shape[-3:-1]) == (self.Height, self.Width):
            n = self._n(data, (3,))
            self._n(out, (3,))
            check(lib().kde_dimconv_projective_to_real_points(self._h, n, data.data_ptr(), out.data_ptr(), _stream()))
        else:
            n = self._n(data, ())
            self._n(out, (3,))
            check(lib().kde_dimconv_projective_to_real_depth(self._h, n, data.data_ptr(), out.data_ptr(), _stream()))
        return out

    def projectiveToRealInterp(self, data: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        n = self._n(data, ())
        self._n(out, (3,))
        check(lib().kde_dimconv_projective_to_real_interp(self._h, n, data.data_ptr(), out.data_ptr(), _stream()))
        return out

    def realToProjective(self, data: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        n = self._n(data, (3,))
        self._n(out, (3,))
        check(lib().kde_dimconv_real_to_projective(self._h, n, data.data_ptr(), out.data_ptr(), _stream()))
        return out


class Buffer2D(_Handle):
    """ArrayBuffer/Buffer2D.h:9-37 (the OpenNI DepthMetaData overload is out of scope)."""
    _destroy = "kde_buffer2d_destroy"

    def __init__(self, width: int, height: int):
        super().__init__()
        self.width, self.height = width, height
        check(lib().kde_buffer2d_create(C.byref(self._h), width, height))

    def insertData(self, data: torch.Tensor) -> None:
        """float* [H,W]; float2* [H,W,2] (d = .x, w = row index [sic]); weighted_d* is insertWeighted."""
        if data.dim() == 3 and data.shape[-1] == 2:
            _req(data, torch.float32, (self.height, self.width, 2), "data")
            check(lib().kde_buffer2d_insert_float2(self._h, data.data_ptr(), _stream()))
        else:
            _req(data, torch.float32, (self.height, self.width), "data")
            check(lib().kde_buffer2d_insert_depth(self._h, data.data_ptr(), _stream()))

    def insertWeighted(self, data: torch.Tensor) -> None:
        _req(data, torch.float32, (self.height, self.width, 2), "data")
        check(lib().kde_buffer2d_insert_weighted(self._h, data.data_ptr(), _stream()))

    def getDepthMap(self, out: torch.Tensor) -> torch.Tensor:
        _req(out, torch.float32, (self.height, self.width), "out")
        check(lib().kde_buffer2d_get_depth_map(self._h, out.data_ptr(), _stream()))
        return out

    def getWeightMap(self, out: torch.Tensor) -> torch.Tensor:
        _req(out, torch.float32, (self.height, self.width), "out")
        check(lib().kde_buffer2d_get_weight_map(self._h, out.data_ptr(), _stream()))
        return out

    def updateData(self, data: torch.Tensor) -> None:
        if data.dim() == 3:   # a sequence of frames [F,H,W], fused into one pass
            _req(data, torch.float32, (data.shape[0], self.height, self.width), "data")
            check(lib().kde_buffer2d_update_sequence(self._h, data.shape[0], data.data_ptr(), _stream()))
        else:
            _req(data, torch.float32, (self.height, self.width), "data")
            check(lib().kde_buffer2d_update(self._h, data.data_ptr(), _stream()))

    def getRawPointer(self) -> torch.Tensor:
        p = C.c_void_p()
        check(lib().kde_buffer2d_raw_pointer(self._h, C.byref(p)))
        return _view(p.value, (self.height, self.width, 2), torch.float32, self)


class DepthAdaptiveSuperpixel(_Handle):
    """SuperpixelSegmentation/DepthAdaptiveSuperpixel.h:15-28."""
    _destroy = "kde_dasp_destroy"

    def __init__(self, width: int, height: int):
        super().__init__()
        self.width, self.height = width, height
        self.rows = self.cols = 0
        check(lib().kde_dasp_create(C.byref(self._h), width, height))

    def SetParametor(self, rows: int, cols: int, intrinsic) -> None:
        k = _K9(intrinsic)
        check(lib().kde_dasp_set_parameters(self._h, rows, cols, k.ctypes.data))
        self.rows, self.cols = rows, cols

    def Segmentation(self, color_image: torch.Tensor, points3d_device: torch.Tensor, color_sigma: float,
                     spatial_sigma: float, depth_sigma: float, iteration: int) -> None:
        _req(color_image, torch.uint8, (self.height, self.width, 3), "color_image")
        _req(points3d_device, torch.float32, (self.height, self.width, 3), "points3d_device")
        check(lib().kde_dasp_segmentation(self._h, color_image.data_ptr(), points3d_device.data_ptr(),
                                          color_sigma, spatial_sigma, depth_sigma, iteration, _stream()))

    def _get(self, fn, shape, dtype):
        p = C.c_void_p()
        check(getattr(lib(), fn)(self._h, C.byref(p)))
        return _view(p.value, shape, dtype, self)

    def getLabelDevice(self) -> torch.Tensor:
        return self._get("kde_dasp_labels_device", (self.height, self.width), torch.int32)

    def getMeanDataDevice(self) -> torch.Tensor:
        """superpixel records as raw bytes [rows*cols, 16] (r,g,b,pad, x:int32, y:int32, size:int32)."""
        return self._get("kde_dasp_mean_device", (self.rows * self.cols, 16), torch.uint8)

    def getCentersDevice(self) -> torch.Tensor:
        return self._get("kde_dasp_centers_device", (self.rows * self.cols, 3), torch.float32)

    def getLDDevice(self) -> torch.Tensor:
        """label_distance records as raw bytes [H,W,8] (d:float32, l:int32)."""
        return self._get("kde_dasp_ld_device", (self.height, self.width, 8), torch.uint8)


class EdgeRefinedSuperpixel(_Handle):
    """EdgeRefinedSuperpixel/EdgeRefinedSuperpixel.h:14-45."""
    _destroy = "kde_ers_destroy"

    def __init__(self, width: int, height: int):
        super().__init__()
        self.Width, self.Height = width, height
        check(lib().kde_ers_create(C.byref(self._h), width, height))

    def EdgeRefining(self, color_label_device, depth_label_device, depth_device, color_image) -> None:
        hw = (self.Height, self.Width)
        _req(color_label_device, torch.int32, hw, "color_label_device")
        _req(depth_label_device, torch.int32, hw, "depth_label_device")
        _req(depth_device, torch.float32, hw, "depth_device")
        _req(color_image, torch.uint8, hw + (3,), "color_image")
        check(lib().kde_ers_edge_refining(self._h, color_label_device.data_ptr(), depth_label_device.data_ptr(),
                                          depth_device.data_ptr(), color_image.data_ptr(), _stream()))

    def set_variant(self, v: int) -> None:
        """0 = built-in choice, 1 = packed-pair, 2 = scalar tuned, 3 = generic depthmap_enhancement kernel"""
        check(lib().kde_ers_set_variant(self._h, int(v)))

    def _get(self, fn, dtype):
        p = C.c_void_p()
        check(getattr(lib(), fn)(self._h, C.byref(p)))
        return _view(p.value, (self.Height, self.Width), dtype, self)

    def getRefinedLabels_Device(self):
        return self._get("kde_ers_refined_labels_device", torch.int32)

    def getRefinedDepth_Device(self):
        return self._get("kde_ers_refined_depth_device", torch.float32)

    def getEdgeStageDepth_Device(self):
        """depth after edge_refining, before depthmap_enhancement (for per-kernel parity tests)."""
        return self._get("kde_ers_stage_edge_depth_device", torch.float32)

    def getRefinedLabels_Host(self) -> np.ndarray:
        p = C.c_void_p()
        check(lib().kde_ers_refined_labels_host(self._h, _stream(), C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(self.Height, self.Width)).copy()

    def getRefinedDepth_Host(self) -> np.ndarray:
        p = C.c_void_p()
        check(lib().kde_ers_refined_depth_host(self._h, _stream(), C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(self.Height, self.Width)).copy()


class _PipelineBase(_Handle):
    _prefix = ""

    def __init__(self, width: int, height: int, max_batch: int = 1):
        super().__init__()
        self.Width, self.Height, self.max_batch = width, height, max_batch
        if max_batch == 1:
            check(getattr(lib(), f"{self._prefix}_create")(C.byref(self._h), width, height))
        else:
            check(getattr(lib(), f"{self._prefix}_create_batch")(C.byref(self._h), width, height, max_batch))
        self._n = 1

    def process_batch(self, depth: torch.Tensor, points: torch.Tensor, color: torch.Tensor) -> None:
        """n independent frames back to back ([n,H,W], [n,H,W,3], [n,H,W,3]); the getters then return n frames"""
        n = depth.shape[0]
        hw = (self.Height, self.Width)
        _req(depth, torch.float32, (n,) + hw, "depth")
        _req(points, torch.float32, (n,) + hw + (3,), "points")
        _req(color, torch.uint8, (n,) + hw + (3,), "color")
        check(getattr(lib(), f"{self._prefix}_process_batch")(self._h, n, depth.data_ptr(), points.data_ptr(), color.data_ptr(), _stream()))
        self._n = n

    def _lead(self):
        return () if self._n == 1 else (self._n,)

    def SetParametor(self, rows: int, cols: int, intrinsic) -> None:
        k = _K9(intrinsic)
        check(getattr(lib(), f"{self._prefix}_set_parameters")(self._h, rows, cols, k.ctypes.data))
        self.sp_rows, self.sp_cols = rows, cols

    def Process(self, depth_device: torch.Tensor, points_device: torch.Tensor, color_device: torch.Tensor) -> None:
        hw = (self.Height, self.Width)
        _req(depth_device, torch.float32, hw, "depth_device")
        _req(points_device, torch.float32, hw + (3,), "points_device")
        _req(color_device, torch.uint8, hw + (3,), "color_device")
        check(getattr(lib(), f"{self._prefix}_process")(self._h, depth_device.data_ptr(), points_device.data_ptr(),
                                                        color_device.data_ptr(), _stream()))
        self._n = 1

    def _get(self, fn, shape, dtype):
        p = C.c_void_p()
        check(getattr(lib(), f"{self._prefix}_{fn}")(self._h, C.byref(p)))
        return _view(p.value, shape, dtype, self)

    def getRefinedDepth_Device(self) -> torch.Tensor:
        return self._get("refined_depth_device", self._lead() + (self.Height, self.Width), torch.float32)

    def getRefinedLabels_Device(self) -> torch.Tensor:
        return self._get("refined_labels_device", self._lead() + (self.Height, self.Width), torch.int32)

    def getRefinedDepth_Host(self) -> np.ndarray:
        p = C.c_void_p()
        check(getattr(lib(), f"{self._prefix}_refined_depth_host")(self._h, _stream(), C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=self._lead() + (self.Height, self.Width)).copy()


class RegionGrowingBilateralFilter(_PipelineBase):
    """RegionGrowingBilateralFilter.h:11-27."""
    _destroy = "kde_rgbf_destroy"
    _prefix = "kde_rgbf"

    def getSPLabels_Device(self):
        return self._get("sp_labels_device", self._lead() + (self.Height, self.Width), torch.int32)

    def getDASPLabels_Device(self):
        return self._get("dasp_labels_device", self._lead() + (self.Height, self.Width), torch.int32)


class SPDepthSuperResolution(_PipelineBase):
    """SPDepthSuperResolution.h:17-46."""
    _destroy = "kde_spdsr_destroy"
    _prefix = "kde_spdsr"

    def getEdgeEnhanced3DPoints_Device(self):
        return self._get("edge_enhanced_points_device", self._lead() + (self.Height, self.Width, 3), torch.float32)

    def getOptimizedPoints_Device(self):
        return self._get("optimized_points_device", self._lead() + (self.Height, self.Width, 3), torch.float32)

    def getOptimizedPoints_Host(self) -> np.ndarray:
        p = C.c_void_p()
        check(lib().kde_spdsr_optimized_points_host(self._h, _stream(), C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=self._lead() + (self.Height, self.Width, 3)).copy()

    def getPlaneFitted3D_Device(self):
        return self._get("plane_fitted_points_device", self._lead() + (self.Height, self.Width, 3), torch.float32)

    def getClusterND_Device(self):
        return self._get("cluster_nd_device", self._lead() + (self.sp_rows * self.sp_cols, 4), torch.float32)
