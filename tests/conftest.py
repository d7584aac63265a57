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
        terminalreporter.write_sep("-", "parity classes (no pixel is excluded; [stage-wise] lines: BAND pixels are the only ones held to an interval)")
        for line in PARITY_LOG:
            terminalreporter.write_line(line)


CENSUS_LOG = {}     # what -> deviation census of the last comparison of that name (oracle.deviation_census)
CENSUS_ALL = []     # every census taken in this process (tools/stress_parity.py totals them)


def assert_depth_close(got, ref, rtol=1e-4, ill=None, max_bad=0, what="depth", max_flagged=None, max_far=None, max_zero_diff=None,
                       grid=None):
    """HIP vs oracle bar for float depth maps.  EVERY pixel is checked:
      * pixels the oracle's envelope does not flag (and all pixels when no envelope is given): identical zero / NaN
        mask and <= rtol relative error against the float32 restatement;
      * flagged pixels (oracle.Env: a tap on a Q1 decision, or the pixel amplifies the rounding of its own first-pass
        average beyond 5e-5): the value must lie inside the envelope [lo, hi] of binary64 evaluations of the same
        formula (widened by rtol), or be 0 / NaN where the envelope admits that.
    Returns the max relative error on unflagged pixels; per-class counts and the max error on flagged pixels are
    logged (PARITY_LOG) and `max_flagged` bounds the flagged fraction.
    The END-TO-END distance from the float32 value is counted over all pixels (oracle.deviation_census) and bounded:
    `max_far` = ceiling on the fraction of pixels more than rtol away from the float32 value, `max_zero_diff` = ceiling on the
    fraction whose zero mask differs; `grid` (Stage.GRID map) adds the same counts for the denormal-grid class."""
    got = np.asarray(got, np.float32)
    ref = np.asarray(ref, np.float32)
    if ill is not None and not hasattr(ill, "flagged"):
        raise TypeError("assert_depth_close needs an oracle.Env (flags + envelope), not a bare flag map")
    from oracle.oracle import parity_check
    r = parity_check(got, ref, ill, rtol, grid=grid)
    c = r["census"]
    CENSUS_LOG[what] = c
    CENSUS_ALL.append(c)
    if ill is not None or max_far is not None or max_zero_diff is not None:
        PARITY_LOG.append(f"{what} [census vs float32]: > {rtol:g} away {c['n_rel_gt_rtol']} ({c['frac_rel_gt_rtol']:.2e}; unflagged "
                          f"{c['n_rel_gt_rtol_unflagged']}), zero mask differs {c['n_zero_mask_differs']} ({c['frac_zero_mask_differs']:.2e}: "
                          f"gained {c['n_gained_zero']}, lost {c['n_lost_zero']}), NaN mask differs {c['n_nan_mask_differs']}, max rel "
                          f"{c['max_rel']:.2e}, p99 over flagged {c['rel_p99_flagged']:.2e}"
                          + (f"; GRID {c['grid_pixels']} px: > rtol {c['grid_rel_gt_rtol']}, zero mask {c['grid_zero_mask_differs']}" if grid is not None else ""))
    if max_far is not None:
        assert c["frac_rel_gt_rtol"] <= max_far, f"{what}: {c['frac_rel_gt_rtol']:.3e} of the pixels are more than {rtol:g} from the float32 value (> {max_far})"
    if max_zero_diff is not None:
        assert c["frac_zero_mask_differs"] <= max_zero_diff, f"{what}: zero mask differs on {c['frac_zero_mask_differs']:.3e} of the pixels (> {max_zero_diff})"
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


def _stage_report(r, what, band_max, ignore=None, decision_max=None):
    PARITY_LOG.append(f"{what} [stage-wise]: {r['n']} px, BAND {r['band']} ({r['band_frac']:.2e}: decision {r['band_decision']} = {r['band_decision_frac']:.2e}, grid {r['grid']}; width p50 "
                      f"{r['band_width_p50']:.1e} p99 {r['band_width_p99']:.1e} max {r['band_width_max']:.1e}), strict max rel "
                      f"{r['max_rel_strict']:.2e}, average within {r['avg_bound_frac_max']:.2f} of its bound "
                      f"(bound p50 {r['avg_tol_p50']:.1e} max {r['avg_tol_max']:.1e}), deviation within {r['dev_bound_frac_max']:.2f}")
    bad = r["bad"] if ignore is None else (r["bad"] & ~ignore)
    nbad = int(bad.sum())
    if nbad:
        idx = [tuple(int(v) for v in i) for i in np.argwhere(bad)[:8]]
        raise AssertionError(f"{what}: {nbad} pixels fail the stage-wise check (avg {r['bad_avg']}, dev {r['bad_dev']}, nan "
                             f"{r['bad_nan']}, zero-mask {r['bad_zero']}, rel {r['bad_rel']} max {r['max_rel_strict']:.3e}, outside "
                             f"band interval {r['outside_band']}, mismatch {r['mismatch']}); first: {idx}")
    assert r["band_frac"] <= band_max, f"{what}: {r['band_frac']:.3e} of the pixels are BAND (> {band_max})"
    if decision_max is not None:
        assert r["band_decision_frac"] <= decision_max, \
            f"{what}: {r['band_decision_frac']:.3e} of the pixels are interval-checked for a tap on a decision (> {decision_max})"
    # every strict pixel that has a first-pass average had it compared with binary64 (no silent exclusion)
    unchecked = int((r["avg_unchecked"] if ignore is None else (r["avg_unchecked"] & ~ignore)).sum())
    assert unchecked <= 1e-4 * r["n"], f"{what}: the first-pass average of {unchecked} strict pixels was not compared with binary64"


def assert_k1_stagewise(params, depth, guide, got, variant=-1, what="K1", band_max=0.02, rtol=1e-4, ignore=None, decision_max=None):
    """The K1 bar (round 3).  `got` is the PRODUCT library's output for (depth, guide, params, variant), [H,W]:
      1. the same call on tools/hooks/libkde_hip_stage.so (same sources + dumps) must reproduce `got` to the bit, so the
         first-pass average it dumps is the one the product kernel used;
      2. that average against the binary64 average within the float32 first-order bound (oracle avg_bound);
      3. every pixel with no tap on a Q1 decision AT THAT AVERAGE: <= rtol against pass 2 evaluated in binary64 from it,
         identical zero mask; the few BAND pixels: inside the interval of both outcomes of the open decision.
    `guide` is the image K1 is guided by (the K0 output when Process ran).  Returns the statistics."""
    import ctypes
    from kinectdepthmapenhancement_amd._native import JbfParams
    from oracle import oracle as O
    from tools.hooks import stage
    p = JbfParams()
    ctypes.memmove(ctypes.byref(p), ctypes.byref(params), ctypes.sizeof(JbfParams))
    p.presmooth = 0
    depth = np.ascontiguousarray(depth, np.float32)
    guide = np.ascontiguousarray(guide, np.uint8)
    out, avg, _ = stage.jbf_stage_run(p, depth[None], guide[None], variant)
    same = stage.bits_equal(out[0], got) if ignore is None else stage.bits_equal(np.where(ignore, 0, out[0]), np.where(ignore, 0, got))
    assert same, f"{what}: the stage build's output differs from the product library's"
    st = O.jbf_stage(depth, guide, p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma, avg_in=avg[0])
    r = O.stage_check(got, st, rtol)
    _stage_report(r, what, band_max, ignore, decision_max)
    return r


def assert_k10_stagewise(color_labels, depth_labels, depth, bgr, got, variant=0, what="K10", band_max=0.02, rtol=1e-4,
                         ignore=None, decision_max=None):
    """The K10 bar (round 3), as assert_k1_stagewise: EdgeRefinedSuperpixel::EdgeRefining is re-run on the stage build
    (final depth bit-identical to `got`), which dumps the label-restricted average and the mean absolute deviation per
    pixel; both are checked against binary64, and the last pass is evaluated in binary64 from them."""
    from oracle import oracle as O
    from tools.hooks import stage
    s = stage.ers_stage_run(color_labels, depth_labels, depth, bgr, variant)
    eq = stage.bits_equal(s["depth"], got) if ignore is None else stage.bits_equal(np.where(ignore, 0, s["depth"]), np.where(ignore, 0, got))
    assert eq, f"{what}: the stage build's output differs from the product library's"
    st = O.ers_stage(s["edge_depth"], bgr, s["labels"], avg_in=s["avg"], dev_in=s["dev"])
    r = O.stage_check(got, st, rtol)
    _stage_report(r, what, band_max, ignore, decision_max)
    return r


def assert_mrf_close(got, ref, what="MRF", rtol=1e-4):
    """MRF bar: 1e-4 relative; outputs below 1e-30 mm (an invalid centre plus taps whose weight is a float denormal,
    e.g. exp(-100) at the reference's ColorSigma = 50) are quantisation noise in ANY float32 evaluation -- the
    oracle's own sums round to the denormal grid -- and only have to be that small as well."""
    got, ref = np.asarray(got, np.float32), np.asarray(ref, np.float32)
    tiny = np.abs(ref) < 1e-30
    assert np.all(np.abs(got[tiny]) < 1e-30), what
    assert_depth_close(np.where(tiny, 0.0, got), np.where(tiny, 0.0, ref), rtol, what=what)
