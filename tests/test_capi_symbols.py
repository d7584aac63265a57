"""The C-ABI library loads and exports every symbol include/kde_hip.h declares (no compute calls:
this runs without a GPU)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "kde_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kde_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def native():
    from kinectdepthmapenhancement_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        _native.build()
    return _native


def test_header_declares_a_reasonable_surface():
    names = declared_functions()
    assert len(names) >= 70
    for must in ("kde_jbf_process", "kde_rgbf_process", "kde_spdsr_process", "kde_dimconv_projective_to_real_depth",
                 "kde_buffer2d_update", "kde_dasp_segmentation", "kde_ers_edge_refining"):
        assert must in names


def test_library_exports_every_declared_symbol(native):
    lib = ctypes.CDLL(native.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in kde_hip.h but not exported: {missing}"


def test_hooks_are_not_part_of_the_product_abi(native):
    """test / measurement hooks live in tools/hooks/libkde_hooks.so (include/kde_test_hooks.h), not in libkde_hip.so"""
    out = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert not {n for n in exported if n.startswith(("kde_test_", "kde_bench_"))}
    # and everything the library exports with a kde_ prefix is declared in the public header
    assert {n for n in exported if n.startswith("kde_")} == set(declared_functions())


def test_stage_build_is_the_same_abi_plus_one_hook(native):
    """tools/hooks/libkde_hip_stage.so = the product sources with -DKDE_STAGE_HOOKS: every product symbol plus
    kde_stage_set (include/kde_test_hooks.h), which the product library must not contain"""
    path = os.path.join(ROOT, "tools", "hooks", "libkde_hip_stage.so")
    subprocess.check_call(["make", "-C", os.path.dirname(path), "-s", "-j8", "libkde_hip_stage.so"])     # no-op when up to date
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("kde_")}
    assert exported == set(declared_functions()) | {"kde_stage_set"}
    hooks_header = open(os.path.join(ROOT, "include", "kde_test_hooks.h")).read()
    assert "kde_stage_set" in hooks_header and "kde_stage_set" not in open(HEADER).read()
    prod = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True).stdout
    assert "kde_stage_set" not in prod and "g_stage" not in prod


def test_python_binding_covers_the_header(native):
    names = set(declared_functions())
    assert names == set(native.SIGNATURES), (names ^ set(native.SIGNATURES))
    native.lib()    # binds restype/argtypes for every symbol; AttributeError if one is gone


def test_abi_version_and_error_paths_without_gpu(native):
    lib = native.lib()
    assert lib.kde_abi_version() == 1
    # argument validation happens before any HIP call, so these are safe on a CPU-only host
    h = ctypes.c_void_p()
    p = native.JbfParams()
    assert lib.kde_jbf_default_params(ctypes.byref(p)) == 0
    assert (p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma) == (5, 70.0, 50.0, 20.0)
    assert (p.presmooth, p.presmooth_kernel_size, p.presmooth_sigma_color, p.presmooth_sigma_spatial) == (1, 5, 30.0, 30.0)
    p.window_size = 4
    rc = lib.kde_jbf_create(ctypes.byref(h), 640, 480, 1, ctypes.byref(p))
    assert rc == native.KDE_ERR_INVALID and b"odd" in lib.kde_last_error_string()
    assert lib.kde_jbf_create(ctypes.byref(h), 0, 480, 1, None) == native.KDE_ERR_INVALID
    assert lib.kde_jbf_process(None, None, None, 0, None) == native.KDE_ERR_INVALID
    assert lib.kde_dimconv_set_camera(None, None, 1, 1) == native.KDE_ERR_INVALID
    with pytest.raises(native.KdeError):
        native.check(lib.kde_rgbf_process(None, None, None, None, None))


def test_product_does_not_reference_the_oracle():
    """the shipped path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "kinectdepthmapenhancement_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "kde_oracle" not in txt and "okde_" not in txt and "from oracle" not in txt, os.path.join(dp, f)
    from kinectdepthmapenhancement_amd import _native
    out = subprocess.run(["ldd", _native.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_product_library_reads_no_environment_variable():
    """the A/B switches (and the kernels that were measured slower and dropped) live in the measurement
    build tools/hooks/libkde_hip_ab.so (-DKDE_AB_SWITCHES) only -- the product library does not even import getenv, and
    carries none of the rejected kernels"""
    from kinectdepthmapenhancement_amd import _native
    syms = subprocess.run(["nm", "-D", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms
    blob = open(_native.LIB_PATH, "rb").read()
    for name in (b"presmooth22_kernel", b"mrf_sweep2_kernel", b"KDE_K0_2X2", b"KDE_K0_BAND_WALK", b"KDE_K10_MASK_PRODUCT",
                 b"KDE_K8_NO_BAND_WALK", b"KDE_SWEEP_NO_BAND_WALK", b"KDE_SPDSR_TWO_SWEEPS"):
        assert name not in blob, name
    for f in ("jbf_kernels.hip", "jbf_fast.hip", "ers_kernels.hip", "dasp_kernels.hip", "spdsr_kernels.hip", "stream_kernels.hip", "kde_api.cpp"):
        txt = open(os.path.join(_native.CSRC, f)).read()
        assert "getenv(" not in txt.replace("::getenv(name)", ""), f       # only through KDE_AB_ENV (kde_internal.h)
    ab = os.path.join(ROOT, "tools", "hooks", "libkde_hip_ab.so")
    if os.path.exists(ab):
        assert "getenv" in subprocess.run(["nm", "-D", ab], capture_output=True, text=True, check=True).stdout


def test_every_entry_point_rejects_null_arguments_without_a_gpu():
    """Every int-returning entry point that takes a handle or an out-pointer first is called with all-zero arguments in a
    child process: the answer must be KDE_ERR_INVALID (validation precedes any HIP call) — `*_destroy(NULL)` is a no-op like
    `free(NULL)` — and the child must not crash."""
    code = r"""
import ctypes as C, json, sys
sys.path.insert(0, %r)
from kinectdepthmapenhancement_amd import _native as N
lib = N.lib()
out = {}
for name, (res, args) in N.SIGNATURES.items():
    if res is not C.c_int or not args or args[0] not in (C.c_void_p, C.POINTER(C.c_void_p)):
        continue
    zeros = []
    for a in args:
        if a in (C.c_int, C.c_size_t):
            zeros.append(0)
        elif a is C.c_float:
            zeros.append(0.0)
        else:
            zeros.append(None)
    out[name] = getattr(lib, name)(*zeros)
print(json.dumps(out))
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    import json
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(res) >= 60
    from kinectdepthmapenhancement_amd import _native as N
    bad = {k: v for k, v in res.items() if v != (N.KDE_OK if k.endswith("_destroy") else N.KDE_ERR_INVALID)}
    assert not bad, bad
