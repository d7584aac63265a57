"""ctypes loader of tools/hooks/libkde_hooks.so (include/kde_test_hooks.h): the float4 copy used as the empirical HBM
ceiling by bench.py and the device-function probe used by tests.  Not product code."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libkde_hooks.so")
_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        l = C.CDLL(_PATH)
        l.kde_bench_copy.restype = C.c_int
        l.kde_bench_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        l.kde_test_sqrt_int24.restype = C.c_int
        l.kde_test_sqrt_int24.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        l.kde_bench_bgr3_copy.restype = C.c_int
        l.kde_bench_bgr3_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        l.kde_test_fastdiv24.restype = C.c_int
        l.kde_test_fastdiv24.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = l
    return _lib


def hbm_copy(src, dst, stream=0) -> None:
    """float4 streaming copy of a torch tensor (empirical HBM ceiling)"""
    rc = lib().kde_bench_copy(src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size(), stream)
    if rc:
        raise RuntimeError(f"kde_bench_copy failed ({rc})")
