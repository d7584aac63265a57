"""Shared fixtures.  `-m "not gpu"` covers the CPU oracle, host logic and ABI export checks;
`-m gpu` tests are the parity tests proper and call the HIP path through the C ABI."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    O.set_threads(min(8, os.cpu_count() or 1))
    return O


@pytest.fixture(scope="session")
def synth():
    from kinectdepthmapenhancement_amd import synth as S
    return S


_frames = {}


@pytest.fixture(scope="session")
def frame(synth):
    """frame(seed, w, h) -> (bgr, depth), cached for the session."""
    def get(seed=1, w=640, h=480):
        key = (seed, w, h)
        if key not in _frames:
            _frames[key] = synth.make_frame(seed, w, h)
        return _frames[key]
    return get


@pytest.fixture(scope="session")
def color_fixture():
    """raw BGR decode of the reference's input/color.jpg (committed as a lossless PNG)."""
    from PIL import Image
    rgb = np.asarray(Image.open(os.path.join(GOLDEN, "color_640x480.png")).convert("RGB"))
    return np.ascontiguousarray(rgb[..., ::-1])


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU: the HIP path has no CPU fallback")
    return torch


def assert_depth_close(got, ref, rtol=1e-4, ill=None, max_bad=0, what="depth"):
    """HIP vs oracle bar for float depth maps: identical zero / non-zero mask and <= rtol relative
    error on non-zero outputs (NaNs must coincide).  Pixels the oracle flags ill-conditioned
    (every surviving weight denormal-scale) are excluded and counted."""
    got = np.asarray(got, np.float32)
    ref = np.asarray(ref, np.float32)
    assert got.shape == ref.shape
    keep = np.ones(ref.shape, bool) if ill is None else ~np.asarray(ill, bool)
    nan_ref, nan_got = np.isnan(ref), np.isnan(got)
    bad_nan = (nan_ref != nan_got) & keep
    zero_mismatch = ((ref == 0) != (got == 0)) & keep & ~nan_ref & ~nan_got
    fin = keep & ~nan_ref & ~nan_got & (ref != 0) & (got != 0)
    rel = np.zeros(ref.shape, np.float64)
    rel[fin] = np.abs(got[fin].astype(np.float64) - ref[fin]) / np.abs(ref[fin])
    bad = bad_nan | zero_mismatch | (rel > rtol)
    nbad = int(bad.sum())
    if nbad > max_bad:
        idx = np.argwhere(bad)[:10]
        detail = [(tuple(i), float(got[tuple(i)]), float(ref[tuple(i)])) for i in idx]
        raise AssertionError(f"{what}: {nbad} pixels off (nan {int(bad_nan.sum())}, zero-mask "
                             f"{int(zero_mismatch.sum())}, max rel {rel.max():.3e}); first: {detail}")
    return float(rel.max())


def assert_mrf_close(got, ref, what="MRF", rtol=1e-4):
    """MRF bar: 1e-4 relative; outputs below 1e-30 mm (an invalid centre plus taps whose weight is a float denormal,
    e.g. exp(-100) at the reference's ColorSigma = 50) are quantisation noise in ANY float32 evaluation -- the
    oracle's own sums round to the denormal grid -- and only have to be that small as well."""
    got, ref = np.asarray(got, np.float32), np.asarray(ref, np.float32)
    tiny = np.abs(ref) < 1e-30
    assert np.all(np.abs(got[tiny]) < 1e-30), what
    assert_depth_close(np.where(tiny, 0.0, got), np.where(tiny, 0.0, ref), rtol, what=what)
