"""tools/hooks/libkde_hip_ab.so: the product library's sources compiled with -DKDE_AB_SWITCHES -- the build that reads the
KDE_* environment switches and carries the kernels that were measured slower and dropped.  A/B tools and the bit-identity
tests of those kernels bind to it with use_ab_library() BEFORE the first library call; the product library has no switch."""
import os

AB_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libkde_hip_ab.so")
SWITCHES = ("KDE_K0_2X2", "KDE_K0_BAND_WALK", "KDE_K8_NO_BAND_WALK", "KDE_K10_MASK_PRODUCT", "KDE_SWEEP_NO_BAND_WALK", "KDE_SPDSR_TWO_SWEEPS",
            "KDE_SPDSR_RESIDENT", "KDE_K8_ROWS")


def use_ab_library() -> str:
    from kinectdepthmapenhancement_amd import _native
    if not os.path.exists(AB_LIB):
        raise FileNotFoundError(f"{AB_LIB} is missing: make -C tools/hooks")
    _native.use_library(AB_LIB)
    return AB_LIB


def use_ab_library_if_switched() -> bool:
    """for tools that run either way: bind to the A/B build iff one of its switches is set in the environment"""
    if any(os.environ.get(k) is not None for k in SWITCHES):
        try:
            use_ab_library()
        except RuntimeError as e:       # imported into a process that already runs on the product library: no switch there
            import sys
            print(f"tools/hooks/ab.py: A/B switches ignored ({e})", file=sys.stderr)
            return False
        return True
    return False
