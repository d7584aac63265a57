"""kinectdepthmapenhancement_amd — MI355X (gfx950) native depth-enhancement filter stage.

Drop-in for the hot path of stevesuyao/KinectDepthMapEnhancement
(DimensionConvertor / Buffer2D -> JointBilateralFilter -> RegionGrowingBilateralFilter) behind the
C ABI of include/kde_hip.h.  `filters` (which needs torch) is imported lazily.
"""
from . import _native, sharding, synth  # noqa: F401
from ._native import JbfParams, KdeError  # noqa: F401

__all__ = ["_native", "sharding", "synth", "filters", "JbfParams", "KdeError"]


def __getattr__(name):
    if name == "filters":
        import importlib
        return importlib.import_module(".filters", __name__)
    raise AttributeError(name)
