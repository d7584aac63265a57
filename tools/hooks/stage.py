"""Loader of tools/hooks/libkde_hip_stage.so (include/kde_test_hooks.h): the product library's own sources compiled with
-DKDE_STAGE_HOOKS -- same ABI, plus kde_stage_set().  Test infrastructure only.

    with stage_library() as ctl:          # inside the block kinectdepthmapenhancement_amd.filters talks to the stage build
        ctl.set(jbf_avg=tensor)           # K1 dumps its first-pass average there
        ...run filters.JointBilateralFilter as usual...

Handles created inside the block must be closed inside it (a handle belongs to the library that made it).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "libkde_hip_stage.so")
_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-j8", "libkde_hip_stage.so"])     # no-op when up to date
        from kinectdepthmapenhancement_amd import _native
        l = C.CDLL(PATH)
        for name, (res, args) in _native.SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        l.kde_stage_set.restype = C.c_int
        l.kde_stage_set.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib = l
    return _lib


class StageCtl:
    def __init__(self, l):
        self._l = l

    def set(self, jbf_avg=None, ers_avg=None, ers_dev=None, counters=None, force_full_rules=False):
        """torch CUDA tensors (float32 sinks, int32[8] counters) or None"""
        p = lambda t: None if t is None else t.data_ptr()
        rc = self._l.kde_stage_set(p(jbf_avg), p(ers_avg), p(ers_dev), p(counters), 1 if force_full_rules else 0)
        if rc:
            raise RuntimeError(f"kde_stage_set failed ({rc})")

    def clear(self):
        self._l.kde_stage_set(None, None, None, None, 0)


@contextlib.contextmanager
def stage_library():
    from kinectdepthmapenhancement_amd import _native
    l = lib()
    prev = _native._lib
    _native._lib = l
    ctl = StageCtl(l)
    try:
        yield ctl
    finally:
        ctl.clear()
        _native._lib = prev


def bits_equal(a, b) -> bool:
    """float32 arrays equal to the bit (NaN payloads included)"""
    import numpy as np
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def jbf_stage_run(params, depth, guide, variant=-1, force_full_rules=False):
    """K1 on the stage build: depth [n,H,W] f32, guide [n,H,W,3] u8 (numpy) -> (out, avg, counters) numpy.
    `guide` is the image K1 is guided by (the smoothed one when K0 runs in the product call)."""
    import numpy as np
    import torch
    from kinectdepthmapenhancement_amd import filters
    n, h, w = depth.shape
    d = torch.from_numpy(np.ascontiguousarray(depth, np.float32)).cuda()
    g = torch.from_numpy(np.ascontiguousarray(guide, np.uint8)).cuda()
    out = torch.empty((n, h, w), dtype=torch.float32, device="cuda")
    avg = torch.full((n, h, w), -1.0, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(8, dtype=torch.int32, device="cuda")
    with stage_library() as ctl:
        jbf = filters.JointBilateralFilter(w, h, params, max_batch=n)
        try:
            jbf.set_variant(variant)
            ctl.set(jbf_avg=avg, counters=cnt, force_full_rules=force_full_rules)
            jbf.filter_batch(d, g, out)
            torch.cuda.synchronize()
        finally:
            jbf.close()
    return out.cpu().numpy(), avg.cpu().numpy(), cnt.cpu().numpy()


def ers_stage_run(color_labels, depth_labels, depth, bgr, variant=0, force_full_rules=False):
    """EdgeRefinedSuperpixel::EdgeRefining on the stage build -> dict(labels, edge_depth, depth, avg, dev, counters)"""
    import numpy as np
    import torch
    from kinectdepthmapenhancement_amd import filters
    h, w = depth.shape
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).cuda()
    avg = torch.full((h, w), -1.0, dtype=torch.float32, device="cuda")
    dev = torch.full((h, w), -1.0, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(8, dtype=torch.int32, device="cuda")
    with stage_library() as ctl:
        ers = filters.EdgeRefinedSuperpixel(w, h)
        try:
            ers.set_variant(variant)
            ctl.set(ers_avg=avg, ers_dev=dev, counters=cnt, force_full_rules=force_full_rules)
            ers.EdgeRefining(t(color_labels, np.int32), t(depth_labels, np.int32), t(depth, np.float32), t(bgr, np.uint8))
            torch.cuda.synchronize()
            res = {"labels": ers.getRefinedLabels_Device().cpu().numpy(), "edge_depth": ers.getEdgeStageDepth_Device().cpu().numpy(),
                   "depth": ers.getRefinedDepth_Device().cpu().numpy()}
        finally:
            ers.close()
    res.update(avg=avg.cpu().numpy(), dev=dev.cpu().numpy(), counters=cnt.cpu().numpy())
    return res
