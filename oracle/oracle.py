"""ctypes/numpy front-end of the CPU oracle (oracle/kde_oracle.c).

TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (see kde_oracle.h).  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkde_oracle.so")

FLOAT3 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
WEIGHTED_D = np.dtype([("d", "<f4"), ("w", "<f4")])
SUPERPIXEL = np.dtype([("r", "u1"), ("g", "u1"), ("b", "u1"), ("pad", "u1"),
                       ("x", "<i4"), ("y", "<i4"), ("size", "<i4")])
LABEL_DISTANCE = np.dtype([("d", "<f4"), ("l", "<i4")])
assert FLOAT3.itemsize == 12 and SUPERPIXEL.itemsize == 16 and LABEL_DISTANCE.itemsize == 8


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "kde_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.okde_mean_3d_error.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class _CEnv(C.Structure):
    _fields_ = [("flags", C.c_void_p), ("lo", C.c_void_p), ("hi", C.c_void_p)]


class Env:
    """The parity envelope of kde_oracle.h (okde_env): per pixel, flags and the interval [lo, hi] of values a faithful
    evaluation may return.  BAND / COND pixels (`flagged`) are compared with the interval, all others with the float32
    value; ZERO_OK / NAN_OK say whether 0 / NaN are admissible as well."""
    BAND, COND, ZERO_OK, NAN_OK = 2, 4, 8, 16

    def __init__(self, shape=None, flags=None, lo=None, hi=None):
        if flags is None:
            flags, lo, hi = np.zeros(shape, np.uint8), np.zeros(shape, np.float64), np.zeros(shape, np.float64)
        self.flags = np.ascontiguousarray(flags, np.uint8)
        self.lo = np.ascontiguousarray(lo, np.float64)
        self.hi = np.ascontiguousarray(hi, np.float64)
        self._c = _CEnv(self.flags.ctypes.data, self.lo.ctypes.data, self.hi.ctypes.data)

    def ref(self):
        return C.byref(self._c)

    @property
    def flagged(self):
        return (self.flags & (self.BAND | self.COND)) != 0

    def crop(self, ys, xs):
        return Env(flags=self.flags[ys, xs], lo=self.lo[ys, xs], hi=self.hi[ys, xs])

    def to_dict(self, prefix):
        return {prefix + "_flags": self.flags, prefix + "_lo": self.lo, prefix + "_hi": self.hi}

    @staticmethod
    def from_dict(g, prefix):
        return Env(flags=g[prefix + "_flags"], lo=g[prefix + "_lo"], hi=g[prefix + "_hi"])


def deviation_census(got, ref, flagged=None, grid=None, rtol=1e-4):
    """How far an output is from the float32 restatement END TO END, pixel by pixel, whatever class the pixel is in (the
    number north_star's "within 1e-4 of the reference path" asks for):
      n_rel_gt_rtol                 pixels where both hold a number and differ by more than rtol (relative to the float32 value)
      n_rel_gt_rtol_flagged / _unflagged   the same, split by `flagged` (the envelope's BAND / COND classes); unflagged must be 0
      n_zero_mask_differs           pixels where exactly one of the two is 0 (= n_gained_zero + n_lost_zero: the output is 0 where
                                    the restatement holds a value / holds a value where the restatement is 0), and its flagged split
      n_nan_mask_differs            pixels where exactly one of the two is NaN
      max_rel, rel_p99_flagged      the largest deviation, and the 99th percentile over the flagged pixels
      grid_*                        the same counts over `grid` (oracle.Stage.GRID: results on the float32 denormal grid)."""
    got = np.asarray(got, np.float32)
    ref = np.asarray(ref, np.float32).reshape(got.shape)
    fl = np.zeros(got.shape, bool) if flagged is None else np.asarray(flagged, bool).reshape(got.shape)
    nan_ref, nan_got = np.isnan(ref), np.isnan(got)
    num = ~nan_ref & ~nan_got
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.where(num & (ref != 0), np.abs(got.astype(np.float64) - ref.astype(np.float64)) / np.abs(ref.astype(np.float64)), 0.0)
    far = num & (ref != 0) & (got != 0) & (rel > rtol)
    gained = num & (ref != 0) & (got == 0)
    lost = num & (ref == 0) & (got != 0)
    zdiff = gained | lost
    out = {"n": int(got.size), "rtol": rtol,
           "n_rel_gt_rtol": int(far.sum()), "n_rel_gt_rtol_flagged": int((far & fl).sum()), "n_rel_gt_rtol_unflagged": int((far & ~fl).sum()),
           "n_zero_mask_differs": int(zdiff.sum()), "n_gained_zero": int(gained.sum()), "n_lost_zero": int(lost.sum()),
           "n_zero_mask_differs_flagged": int((zdiff & fl).sum()), "n_zero_mask_differs_unflagged": int((zdiff & ~fl).sum()),
           "n_nan_mask_differs": int((nan_ref != nan_got).sum()),
           "frac_rel_gt_rtol": float(far.sum() / max(1, got.size)), "frac_zero_mask_differs": float(zdiff.sum() / max(1, got.size)),
           "max_rel": float(rel[far].max()) if far.any() else float(rel.max()) if rel.size else 0.0,
           "rel_p99_flagged": float(np.percentile(rel[fl & num & (got != 0)], 99)) if (fl & num & (got != 0)).any() else 0.0}
    if grid is not None:
        g = np.asarray(grid, bool).reshape(got.shape)
        out.update(grid_pixels=int(g.sum()), grid_rel_gt_rtol=int((far & g).sum()), grid_zero_mask_differs=int((zdiff & g).sum()),
                   grid_max_rel=float(rel[far & g].max()) if (far & g).any() else 0.0)
    return out


def parity_check(got, ref, env=None, rtol=1e-4, grid=None):
    """The parity bar of this repo, every pixel checked (used by tests/conftest.py, smoke() and bench.py's `verified`):
      * unflagged pixels (all pixels when env is None): identical zero / NaN mask, <= rtol relative to the float32 value;
      * flagged pixels (env.flagged): inside [lo, hi] widened by rtol, or 0 / NaN where the envelope admits that.
    -> dict with the boolean map `bad`, the statistics the tests print, and `census`: deviation_census(got, ref, flagged, grid),
    the end-to-end distance from the float32 value counted over ALL pixels."""
    got = np.asarray(got, np.float32)
    ref = np.asarray(ref, np.float32)
    assert got.shape == ref.shape
    flagged = np.zeros(ref.shape, bool) if env is None else env.flagged.reshape(ref.shape)
    nan_ref, nan_got = np.isnan(ref), np.isnan(got)
    g64, r64 = got.astype(np.float64), ref.astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.where((ref != 0) & ~nan_ref & ~nan_got, np.abs(g64 - r64) / np.abs(r64), 0.0)
    strict = ~flagged
    # ZERO_OK without BAND / COND: the envelope's binary64 evaluation found NO weight where the float32 restatement still
    # holds sub-unit survivors (exact weights below 2^-150 that round up to one denormal unit: a hole whose taps all lie the
    # depth-factor underflow distance away) -- 0 is admissible there next to the float32 value, nothing else is
    # (tools/stress_parity.py seed 603 case 17338: window 3, sigma 1 / 2 / 70; tests/golden/k1_zero_ok_without_band.npz)
    zero_adm = np.zeros(ref.shape, bool) if env is None else (((env.flags.reshape(ref.shape) & Env.ZERO_OK) != 0) & (got == 0) & ~nan_got)
    bad_nan = strict & (nan_ref != nan_got)
    bad_zero = strict & ~nan_ref & ~nan_got & ((ref == 0) != (got == 0)) & ~zero_adm
    bad_rel = strict & (rel > rtol) & ~zero_adm
    bad_env = np.zeros(ref.shape, bool)
    band = cond = 0
    if env is not None and flagged.any():
        fl = env.flags.reshape(ref.shape)
        lo, hi = env.lo.reshape(ref.shape), env.hi.reshape(ref.shape)
        zero_ok = ((fl & Env.ZERO_OK) != 0) | (ref == 0)
        nan_ok = ((fl & Env.NAN_OK) != 0) | nan_ref
        inside = (hi > 0) & (g64 >= lo * (1 - rtol)) & (g64 <= hi * (1 + rtol))
        ok = np.where(nan_got, nan_ok, np.where(got == 0, zero_ok, inside))
        bad_env = flagged & ~ok
        band, cond = int(((fl & Env.BAND) != 0).sum()), int(((fl & Env.COND) != 0).sum())
    cen = deviation_census(got, ref, flagged, grid, rtol)
    return {"bad": bad_nan | bad_zero | bad_rel | bad_env, "rel": rel, "census": cen,
            **{k: cen[k] for k in ("n_rel_gt_rtol", "n_rel_gt_rtol_flagged", "n_zero_mask_differs", "n_nan_mask_differs")},
            **{k: cen[k] for k in ("grid_pixels", "grid_rel_gt_rtol", "grid_zero_mask_differs") if k in cen},
            "n": int(ref.size), "flagged": int(flagged.sum()), "band": band, "cond": cond,
            "bad_nan": int(bad_nan.sum()), "bad_zero": int(bad_zero.sum()), "bad_rel": int(bad_rel.sum()),
            "outside_envelope": int(bad_env.sum()),
            "max_rel_unflagged": float(rel[strict].max()) if strict.any() else 0.0,
            "max_rel_flagged": float(rel[flagged].max()) if flagged.any() else 0.0}


# ---- stage-wise parity (kde_oracle.h: okde_stage) -----------------------------------------------------------------
class _CStage(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("avg32", "dev32", "avg64", "avg_tol", "dev64", "dev_tol", "fin64", "lo", "hi", "flags")]


class Stage:
    """Outputs of okde_jbf_stage / okde_ers_stage (see kde_oracle.h)."""
    BAND, ZERO_OK, NAN_OK, GRID, NOWEIGHT, MISMATCH = 2, 8, 16, 32, 64, 128

    def __init__(self, shape, with_dev=False):
        self.shape = tuple(shape)
        self.avg32 = np.zeros(shape, np.float32)
        self.dev32 = np.zeros(shape, np.float32) if with_dev else None
        self.avg64, self.avg_tol = np.zeros(shape, np.float64), np.zeros(shape, np.float64)
        self.dev64 = np.zeros(shape, np.float64) if with_dev else None
        self.dev_tol = np.zeros(shape, np.float64) if with_dev else None
        self.fin64, self.lo, self.hi = (np.zeros(shape, np.float64) for _ in range(3))
        self.flags = np.zeros(shape, np.uint8)
        ptr = lambda a: None if a is None else a.ctypes.data
        self._c = _CStage(ptr(self.avg32), ptr(self.dev32), ptr(self.avg64), ptr(self.avg_tol), ptr(self.dev64),
                          ptr(self.dev_tol), ptr(self.fin64), ptr(self.lo), ptr(self.hi), ptr(self.flags))

    def ref(self):
        return C.byref(self._c)

    @property
    def band(self):
        return (self.flags & self.BAND) != 0


def jbf_stage(depth, guide, window=5, spatial_sigma=70.0, color_sigma=50.0, depth_sigma=20.0, avg_in=None) -> Stage:
    """Stage data of K1 on `guide` (the image K1 is guided by: the smoothed one when K0 ran).  avg_in: the first-pass
    average an implementation used (float32 [H,W], NaN = no weight); None = the float32 restatement's own."""
    depth, guide = _f32(depth), _u8(guide)
    h, w = depth.shape
    tab = spatial_table(window, spatial_sigma)
    st = Stage((h, w))
    a = None if avg_in is None else _f32(avg_in)
    assert a is None or a.shape == (h, w)
    lib().okde_jbf_stage(w, h, _p(depth), _p(guide), _p(tab), window, C.c_float(color_sigma), C.c_float(depth_sigma),
                         None if a is None else _p(a), st.ref())
    return st


def ers_stage(refined_depth, bgr, refined_labels, window=7, spatial_sigma=30.0, color_sigma=50.0, depth_sigma=70.0,
              avg_in=None, dev_in=None) -> Stage:
    """Stage data of K10 (inputs = the K9 result).  avg_in / dev_in: the pass-1 average and pass-2 deviation an
    implementation used; None = the float32 restatement's own."""
    rd, bgr, rl = _f32(refined_depth), _u8(bgr), _i32(refined_labels)
    h, w = rd.shape
    tab = spatial_table(window, spatial_sigma)
    st = Stage((h, w), with_dev=True)
    assert (avg_in is None) == (dev_in is None)
    a = None if avg_in is None else _f32(avg_in)
    d = None if dev_in is None else _f32(dev_in)
    lib().okde_ers_stage(w, h, _p(rd), _p(bgr), _p(rl), _p(tab), window, C.c_float(color_sigma), C.c_float(depth_sigma),
                         None if a is None else _p(a), None if d is None else _p(d), st.ref())
    return st


def stage_check(final, st: Stage, rtol=1e-4):
    """The stage-wise parity bar, every pixel checked (st was evaluated at the implementation's own average / deviation):
      (i)   that average (and deviation) against binary64 within the float32 first-order bound of kde_oracle.c;
      (ii)  every pixel with no tap on a Q1 decision at that average: identical zero / NaN mask and <= rtol against the
            binary64 last pass evaluated from it;
      (iii) BAND pixels: inside [lo, hi] (both outcomes of the open decision at the same average), or 0 / NaN where admitted.
    -> dict with the boolean map `bad` and the statistics the tests print."""
    got = np.asarray(final, np.float32).reshape(st.shape)
    g64 = got.astype(np.float64)
    fl = st.flags
    band = (fl & Stage.BAND) != 0
    noweight = (fl & Stage.NOWEIGHT) != 0
    mismatch = (fl & Stage.MISMATCH) != 0
    nan_got = np.isnan(got)
    nan_ref = np.isnan(st.fin64)
    have_avg = ~noweight & ~np.isnan(st.avg32)
    with np.errstate(invalid="ignore", divide="ignore"):
        # (i) pass 1
        cmp_avg = have_avg & np.isfinite(st.avg_tol) & (st.avg64 != 0) & np.isfinite(st.avg64) & np.isfinite(st.avg32)
        avg_err = np.where(cmp_avg, np.abs(st.avg32.astype(np.float64) - st.avg64) / np.abs(st.avg64), 0.0)
        avg_frac = np.where(cmp_avg, avg_err / st.avg_tol, 0.0)
        bad_avg = cmp_avg & ~(avg_err <= st.avg_tol)
        bad_dev = np.zeros(st.shape, bool)
        dev_frac = np.zeros(st.shape)
        if st.dev64 is not None:
            cmp_dev = have_avg
            d32 = st.dev32.astype(np.float64)
            exact0 = cmp_dev & (st.dev64 == 0)
            bad_dev = (exact0 & (d32 != 0)) | (cmp_dev & ~exact0 & ~(np.abs(d32 - st.dev64) <= st.dev_tol * st.dev64))
            dev_frac = np.where(cmp_dev & ~exact0, np.abs(d32 - st.dev64) / (st.dev_tol * st.dev64), 0.0)
        # (ii) strict pixels against the binary64 last pass at the given average
        strict = ~band & ~mismatch
        bad_nan = strict & (nan_ref != nan_got)
        bad_zero = strict & ~nan_ref & ~nan_got & ((st.fin64 == 0) != (got == 0))
        rel = np.where((st.fin64 != 0) & ~nan_ref & ~nan_got, np.abs(g64 - st.fin64) / np.abs(st.fin64), 0.0)
        bad_rel = strict & (rel > rtol)
        # (iii) BAND pixels against the interval
        zero_ok = (fl & Stage.ZERO_OK) != 0
        nan_ok = (fl & Stage.NAN_OK) != 0
        inside = (st.hi > 0) & (g64 >= st.lo * (1 - rtol)) & (g64 <= st.hi * (1 + rtol))
        ok_band = np.where(nan_got, nan_ok, np.where(got == 0, zero_ok, inside))
        bad_band = band & ~mismatch & ~ok_band
        width = np.where(band & (st.hi > 0), (st.hi - st.lo) / np.maximum(np.abs(st.fin64), 1e-300), 0.0)
    bad = mismatch | bad_avg | bad_dev | bad_nan | bad_zero | bad_rel | bad_band
    nb = int(band.sum())
    grid = (fl & Stage.GRID) != 0
    ndec = int((band & ~grid).sum())            # interval-checked because a tap is ON a decision (not the denormal-grid class)
    expect_avg = have_avg & ~band & ~mismatch   # every strict pixel that has an average must have had it compared
    q = lambda a, m, pc: float(np.percentile(a[m], pc)) if m.any() else 0.0
    return {"bad": bad, "rel": rel, "avg_unchecked": expect_avg & ~cmp_avg, "grid_map": grid, "band_map": band,
            "n": int(got.size), "band": nb, "grid": int(grid.sum()), "band_decision": ndec,
            "band_frac": nb / max(1, got.size), "band_decision_frac": ndec / max(1, got.size), "grid_frac": int(grid.sum()) / max(1, got.size),
            "avg_checked_of_strict": float((cmp_avg & expect_avg).sum() / max(1, int(expect_avg.sum()))),
            "mismatch": int(mismatch.sum()),
            "bad_avg": int(bad_avg.sum()), "bad_dev": int(bad_dev.sum()), "bad_nan": int(bad_nan.sum()),
            "bad_zero": int(bad_zero.sum()), "bad_rel": int(bad_rel.sum()), "outside_band": int(bad_band.sum()),
            "max_rel_strict": float(rel[strict].max()) if strict.any() else 0.0,
            "avg_checked": int(cmp_avg.sum()),
            "avg_bound_frac_p50": q(avg_frac, cmp_avg, 50), "avg_bound_frac_p99": q(avg_frac, cmp_avg, 99),
            "avg_bound_frac_max": float(avg_frac.max()) if cmp_avg.any() else 0.0,
            "avg_tol_p50": q(st.avg_tol, cmp_avg, 50), "avg_tol_p99": q(st.avg_tol, cmp_avg, 99),
            "avg_tol_max": float(st.avg_tol[cmp_avg].max()) if cmp_avg.any() else 0.0,
            "dev_bound_frac_max": float(dev_frac.max()),
            "band_width_p50": q(width, band, 50), "band_width_p99": q(width, band, 99),
            "band_width_max": float(width.max()) if nb else 0.0}


def set_threads(n: int) -> int:
    return lib().okde_set_threads(int(n))


def max_threads() -> int:
    return lib().okde_max_threads()


def spatial_table(window: int, sigma: float) -> np.ndarray:
    t = np.empty((window, window), np.float32)
    lib().okde_spatial_table(window, C.c_float(sigma), _p(t))
    return t


def cv_bilateral(bgr: np.ndarray, ksize: int = 5, sigma_color: float = 30.0,
                 sigma_spatial: float = 30.0) -> np.ndarray:
    bgr = _u8(bgr)
    h, w, _ = bgr.shape
    out = np.empty_like(bgr)
    lib().okde_cv_bilateral_8uc3(_p(bgr), w, h, C.c_size_t(w * 3), ksize, C.c_float(sigma_color),
                                 C.c_float(sigma_spatial), _p(out), C.c_size_t(w * 3))
    return out


def jbf_kernel(depth, guide, window=5, spatial_sigma=70.0, color_sigma=50.0, depth_sigma=20.0,
               return_ill=False):
    depth, guide = _f32(depth), _u8(guide)
    h, w = depth.shape
    tab = spatial_table(window, spatial_sigma)
    out = np.empty((h, w), np.float32)
    env = Env((h, w)) if return_ill else None          # the envelope probe only runs on request
    lib().okde_jbf_kernel(w, h, _p(depth), _p(guide), _p(tab), window, C.c_float(color_sigma),
                          C.c_float(depth_sigma), _p(out), env.ref() if return_ill else None)
    return (out, env) if return_ill else out


def jbf_process(depth, bgr, window=5, spatial_sigma=70.0, color_sigma=50.0, depth_sigma=20.0,
                presmooth=(5, 30.0, 30.0), return_all=False):
    """JointBilateralFilter::Process. presmooth=None disables K0."""
    depth, bgr = _f32(depth), _u8(bgr)
    h, w = depth.shape
    out = np.empty((h, w), np.float32)
    smooth = np.empty((h, w, 3), np.uint8)
    env = Env((h, w)) if return_all else None    # the envelope probe only runs on request
    ks, sc, ss = presmooth if presmooth is not None else (-100000, 0.0, 0.0)
    lib().okde_jbf_process(w, h, _p(depth), _p(bgr), window, C.c_float(spatial_sigma),
                           C.c_float(color_sigma), C.c_float(depth_sigma), ks, C.c_float(sc),
                           C.c_float(ss), _p(smooth), _p(out), env.ref() if return_all else None)
    return (out, smooth, env) if return_all else out


def mrf_kernel(depth, bgr, window=5, color_sigma=50.0, smooth_sigma=150.0):
    depth, bgr = _f32(depth), _u8(bgr)
    h, w = depth.shape
    out = np.empty((h, w), np.float32)
    lib().okde_mrf_kernel(w, h, _p(depth), _p(bgr), window, C.c_float(color_sigma),
                          C.c_float(smooth_sigma), _p(out))
    return out


def camera_from_K(K):
    """DimensionConvertor::setCameraParameters (DimensionConvertor.cpp:3-13)."""
    K = np.asarray(K, np.float64).reshape(3, 3)
    return float(np.float32(K[0, 0])), float(np.float32(K[1, 1])), int(K[0, 2]), int(K[1, 2])


def p2r_depth(depth, K):
    depth = _f32(depth)
    h, w = depth.shape
    fx, fy, cx, cy = camera_from_K(K)
    out = np.empty((h, w), FLOAT3)
    lib().okde_p2r_depth(w, h, C.c_float(fx), C.c_float(fy), cx, cy, _p(depth), _p(out))
    return out


def _pts(points):
    a = np.ascontiguousarray(points)
    if a.dtype != FLOAT3:
        a = np.ascontiguousarray(a, np.float32)
        assert a.shape[-1] == 3
        a = a.view(FLOAT3).reshape(a.shape[:-1])
    return a


def p2r_points(points, K):
    pts = _pts(points)
    h, w = pts.shape
    fx, fy, cx, cy = camera_from_K(K)
    out = np.empty((h, w), FLOAT3)
    lib().okde_p2r_points(w, h, C.c_float(fx), C.c_float(fy), cx, cy, _p(pts), _p(out))
    return out


def p2r_interp(depth, K):
    depth = _f32(depth)
    h, w = depth.shape
    fx, fy, cx, cy = camera_from_K(K)
    out = np.empty((h, w), FLOAT3)
    lib().okde_p2r_interp(w, h, C.c_float(fx), C.c_float(fy), cx, cy, _p(depth), _p(out))
    return out


def r2p(points, K):
    pts = _pts(points)
    h, w = pts.shape
    fx, fy, cx, cy = camera_from_K(K)
    out = np.empty((h, w), FLOAT3)
    lib().okde_r2p(w, h, C.c_float(fx), C.c_float(fy), cx, cy, _p(pts), _p(out))
    return out


class Buffer2D:
    """Mirror of ArrayBuffer/Buffer2D on the CPU."""

    def __init__(self, width, height):
        self.w, self.h = width, height
        self.buf = np.empty((height, width), WEIGHTED_D)
        lib().okde_buf_init(width * height, _p(self.buf))

    def insert_depth(self, data):
        lib().okde_buf_insert_depth(self.w * self.h, _p(self.buf), _p(_f32(data)))

    def insert_float2(self, data_xy):
        d = _f32(data_xy)
        assert d.shape == (self.h, self.w, 2)
        lib().okde_buf_insert_float2(self.w, self.h, _p(self.buf), _p(d))

    def insert_weighted(self, other):
        self.buf[...] = np.asarray(other).view(WEIGHTED_D).reshape(self.h, self.w)

    def update(self, data):
        lib().okde_buf_update(self.w * self.h, _p(self.buf), _p(_f32(data)))

    def depth_map(self):
        out = np.empty((self.h, self.w), np.float32)
        lib().okde_buf_get_depth(self.w * self.h, _p(self.buf), _p(out))
        return out

    def weight_map(self):
        out = np.empty((self.h, self.w), np.float32)
        lib().okde_buf_get_weight(self.w * self.h, _p(self.buf), _p(out))
        return out


def intr9(K):
    return np.ascontiguousarray(np.asarray(K, np.float64).reshape(9).astype(np.float32))


def dasp_segmentation(bgr, points, rows, cols, K, color_sigma, spatial_sigma, depth_sigma, iteration):
    bgr, pts = _u8(bgr), _pts(points)
    h, w = pts.shape
    labels = np.empty((h, w), np.int32)
    ld = np.empty((h, w), LABEL_DISTANCE)
    mean = np.zeros(rows * cols, SUPERPIXEL)
    centers = np.zeros(rows * cols, FLOAT3)
    k = intr9(K)
    rc = lib().okde_dasp_segmentation(w, h, rows, cols, _p(k), _p(bgr), _p(pts),
                                      C.c_float(color_sigma), C.c_float(spatial_sigma),
                                      C.c_float(depth_sigma), iteration,
                                      _p(labels), _p(ld), _p(mean), _p(centers))
    if rc:
        raise ValueError("DASP geometry rejected")
    return labels, ld, mean, centers


def dasp_steps(bgr, points, rows, cols):
    """init_LD + sampleInitialClusters only (for unit tests of the individual kernels)."""
    bgr, pts = _u8(bgr), _pts(points)
    h, w = pts.shape
    ld = np.empty((h, w), LABEL_DISTANCE)
    mean = np.zeros(rows * cols, SUPERPIXEL)
    centers = np.zeros(rows * cols, FLOAT3)
    lib().okde_dasp_init_ld(w, h, rows, cols, _p(ld))
    lib().okde_dasp_sample_clusters(w, h, rows, cols, _p(bgr), _p(pts), _p(mean), _p(centers))
    return ld, mean, centers


def dasp_calculate_ld(bgr, points, rows, cols, ld, mean, centers, color_sigma, spatial_sigma, depth_sigma):
    bgr, pts = _u8(bgr), _pts(points)
    h, w = pts.shape
    ld = np.ascontiguousarray(ld).copy()
    labels = np.empty((h, w), np.int32)
    lib().okde_dasp_calculate_ld(w, h, rows, cols, _p(bgr), _p(pts), _p(ld), _p(np.ascontiguousarray(mean)),
                                 _p(np.ascontiguousarray(centers)), _p(labels), C.c_float(color_sigma),
                                 C.c_float(spatial_sigma), C.c_float(depth_sigma))
    return labels, ld


def dasp_analyze_clusters(bgr, points, rows, cols, ld, mean, centers, K):
    bgr, pts = _u8(bgr), _pts(points)
    h, w = pts.shape
    mean = np.ascontiguousarray(mean).copy()
    centers = np.ascontiguousarray(centers).copy()
    k = intr9(K)
    lib().okde_dasp_analyze_clusters(w, h, rows, cols, _p(bgr), _p(pts), _p(np.ascontiguousarray(ld)),
                                     _p(mean), _p(centers), _p(k))
    return mean, centers


def ers_edge_refining(color_labels, depth_labels, depth, window=7):
    cl, labels, d = _i32(color_labels), _i32(depth_labels).copy(), _f32(depth).copy()
    h, w = d.shape
    lib().okde_ers_edge_refining(w, h, _p(cl), _p(labels), _p(d), window)
    return labels, d


class ers_flags:
    """context manager: collects okde_ers_enhance's parity envelope (Env) for the call made inside it"""

    def __init__(self, shape):
        self.env = Env(shape)

    def __enter__(self):
        lib().okde_ers_set_env_sink(self.env.ref())
        return self.env

    def __exit__(self, *exc):
        lib().okde_ers_set_env_sink(None)
        return False


def ers_enhance(refined_depth, bgr, refined_labels, window=7, spatial_sigma=30.0, color_sigma=50.0,
                depth_sigma=70.0):
    rd, bgr, rl = _f32(refined_depth), _u8(bgr), _i32(refined_labels)
    h, w = rd.shape
    tab = spatial_table(window, spatial_sigma)
    out = np.empty((h, w), np.float32)
    lib().okde_ers_enhance(w, h, _p(rd), _p(bgr), _p(rl), _p(tab), window, C.c_float(color_sigma),
                           C.c_float(depth_sigma), _p(out))
    return out


def ers_process(color_labels, depth_labels, depth, bgr):
    cl, dl, d, bgr = _i32(color_labels), _i32(depth_labels), _f32(depth), _u8(bgr)
    h, w = d.shape
    rl = np.empty((h, w), np.int32)
    rd = np.empty((h, w), np.float32)
    lib().okde_ers_process(w, h, _p(cl), _p(dl), _p(d), _p(bgr), _p(rl), _p(rd))
    return rl, rd


def rgbf_process(depth, points, bgr, rows, cols, K):
    d, pts, bgr = _f32(depth), _pts(points), _u8(bgr)
    h, w = d.shape
    sp = np.empty((h, w), np.int32)
    da = np.empty((h, w), np.int32)
    rl = np.empty((h, w), np.int32)
    rd = np.empty((h, w), np.float32)
    k = intr9(K)
    rc = lib().okde_rgbf_process(w, h, rows, cols, _p(k), _p(d), _p(pts), _p(bgr),
                                 _p(sp), _p(da), _p(rl), _p(rd))
    if rc:
        raise ValueError("DASP geometry rejected")
    return {"sp_labels": sp, "dasp_labels": da, "refined_labels": rl, "refined_depth": rd}


def spdsr_head(depth, points, bgr, rows, cols, K):
    d, pts, bgr = _f32(depth), _pts(points), _u8(bgr)
    h, w = d.shape
    rl = np.empty((h, w), np.int32)
    rd = np.empty((h, w), np.float32)
    rp = np.empty((h, w), FLOAT3)
    k = np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))
    rc = lib().okde_spdsr_head(w, h, rows, cols, _p(k), _p(d), _p(pts), _p(bgr), _p(rl), _p(rd), _p(rp))
    if rc:
        raise ValueError("DASP geometry rejected")
    return rl, rd, rp


def spdsr_cluster_planes(labels, points, nclusters, nd_prev=None):
    """ClusterND (float4 per cluster: normal.xyz, plane distance) as SPDepthSuperResolution.cpp:65-170 computes it"""
    lab, pts = _i32(labels), _pts(points)
    h, w = lab.shape
    nd = np.zeros((nclusters, 4), np.float32) if nd_prev is None else np.ascontiguousarray(nd_prev, np.float32).copy()
    lib().okde_spdsr_cluster_planes(w, h, nclusters, _p(lab), _p(pts), _p(nd))
    return nd


def projection_plane(nd, labels, points, K, sweeps=20):
    """Projection_GPU::PlaneProjection(nd, labels, points): (plane_fitted, optimized) float3 images"""
    lab, pts = _i32(labels), _pts(points)
    h, w = lab.shape
    nd = np.ascontiguousarray(nd, np.float32)
    fx, fy, cx, cy = camera_from_K(K)
    pf = np.empty((h, w), FLOAT3)
    opt = np.empty((h, w), FLOAT3)
    lib().okde_projection_plane(w, h, C.c_float(fx), C.c_float(fy), cx, cy, _p(nd), nd.shape[0], _p(lab), _p(pts),
                                _p(pf), _p(opt), sweeps)
    return pf, opt


def mean_3d_error(points, truth):
    a, b = _pts(points), _pts(truth)
    cnt = C.c_int(0)
    e = lib().okde_mean_3d_error(a.size, _p(a), _p(b), C.byref(cnt))
    return float(e), cnt.value
