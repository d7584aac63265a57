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


PARITY_LOG = []     # one line per assert_depth_close call with an envelope; printed in the terminal summary


def pytest_terminal_summary(terminalreporter):
    if PARITY_LOG:
        terminalreporter.write_sep("-", "parity classes (no pixel is excluded; flagged pixels are held to the oracle's envelope)")
        for line in PARITY_LOG:
            terminalreporter.write_line(line)


def assert_depth_close(got, ref, rtol=1e-4, ill=None, max_bad=0, what="depth", max_flagged=None):
    """HIP vs oracle bar for float depth maps.  EVERY pixel is checked:
      * pixels the oracle's envelope does not flag (and all pixels when no envelope is given): identical zero / NaN
        mask and <= rtol relative error against the float32 restatement;
      * flagged pixels (oracle.Env: a tap on a Q1 decision, or the pixel amplifies the rounding of its own first-pass
        average beyond 5e-5): the value must lie inside the envelope [lo, hi] of binary64 evaluations of the same
        formula (widened by rtol), or be 0 / NaN where the envelope admits that.
    Returns the max relative error on unflagged pixels; per-class counts and the max error on flagged pixels are
    logged (PARITY_LOG) and `max_flagged` bounds the flagged fraction."""
    got = np.asarray(got, np.float32)
    ref = np.asarray(ref, np.float32)
    if ill is not None and not hasattr(ill, "flagged"):
        raise TypeError("assert_depth_close needs an oracle.Env (flags + envelope), not a bare flag map")
    from oracle.oracle import parity_check
    r = parity_check(got, ref, ill, rtol)
    if ill is not None:
        frac = r["flagged"] / max(1, r["n"])
        PARITY_LOG.append(f"{what}: {r['n']} px, flagged {r['flagged']} ({frac:.2e}: band {r['band']}, cond {r['cond']}), "
                          f"max rel err unflagged {r['max_rel_unflagged']:.2e}, flagged vs float32 value "
                          f"{r['max_rel_flagged']:.2e}, outside envelope {r['outside_envelope']}")
        if max_flagged is not None:
            assert frac <= max_flagged, f"{what}: {frac:.3e} of the pixels flagged (> {max_flagged})"
    nbad = int(r["bad"].sum())
    if nbad > max_bad:
        idx = np.argwhere(r["bad"])[:10]
        detail = [(tuple(i), float(got[tuple(i)]), float(ref[tuple(i)])) for i in idx]
        raise AssertionError(f"{what}: {nbad} pixels off (nan {r['bad_nan']}, zero-mask {r['bad_zero']}, rel {r['bad_rel']} "
                             f"max {r['max_rel_unflagged']:.3e}, outside envelope {r['outside_envelope']}); first: {detail}")
    return r["max_rel_unflagged"]


def assert_mrf_close(got, ref, what="MRF", rtol=1e-4):
    """MRF bar: 1e-4 relative; outputs below 1e-30 mm (an invalid centre plus taps whose weight is a float denormal,
    e.g. exp(-100) at the reference's ColorSigma = 50) are quantisation noise in ANY float32 evaluation -- the
    oracle's own sums round to the denormal grid -- and only have to be that small as well."""
    got, ref = np.asarray(got, np.float32), np.asarray(ref, np.float32)
    tiny = np.abs(ref) < 1e-30
    assert np.all(np.abs(got[tiny]) < 1e-30), what
    assert_depth_close(np.where(tiny, 0.0, got), np.where(tiny, 0.0, ref), rtol, what=what)
