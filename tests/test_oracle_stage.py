"""The stage-wise parity machinery itself, on the CPU: the float32 restatement's OWN evaluation must pass the check the
GPU kernels are held to (oracle.stage_check on okde_jbf_stage / okde_ers_stage, kde_oracle.h okde_stage) -- its first-pass
average within the float32 first-order bound of the binary64 one, its final value within 1e-4 of the last pass evaluated
in binary64 from that average -- and the check must reject perturbed averages / finals."""
import numpy as np
import pytest


@pytest.mark.parametrize("cfg", [(5, 70.0, 50.0, 20.0), (11, 3.0, 7.65, 20.0), (19, 3.0, 7.65, 20.0), (7, 5.0, 20.0, 4.0),
                                 (3, 1.0, 0.0, 20.0), (5, 70.0, 50.0, 0.0)])
def test_k1_restatement_passes_its_own_stage_check(oracle, frame, cfg):
    bgr, depth = frame(2, 160, 120)
    w, ss, cs, ds = cfg
    out = oracle.jbf_kernel(depth, bgr, w, ss, cs, ds)
    st = oracle.jbf_stage(depth, bgr, w, ss, cs, ds)                   # avg_in=None: the restatement's own average
    r = oracle.stage_check(out, st)
    assert not r["bad"].any(), {k: v for k, v in r.items() if k not in ("bad", "rel", "avg_unchecked")}
    assert r["mismatch"] == 0 and r["max_rel_strict"] < 2e-5 and r["avg_bound_frac_max"] < 1.0
    assert r["band_frac"] < 0.02
    # the check has teeth: an average moved by 30 of its bounds, or a final value moved by 3e-4, is rejected
    avg = st.avg32.copy()
    ok = np.isfinite(avg) & np.isfinite(st.avg_tol)
    avg[ok] = (avg[ok].astype(np.float64) * (1.0 + 30.0 * st.avg_tol[ok])).astype(np.float32)
    r2 = oracle.stage_check(out, oracle.jbf_stage(depth, bgr, w, ss, cs, ds, avg_in=avg))
    assert r2["bad_avg"] > 0.9 * ok.sum()
    r3 = oracle.stage_check(out * np.float32(1.0003), st)
    assert r3["bad_rel"] > 0.9 * (out != 0).sum() - r["band"]


def test_k1_stage_flags_band_pixels_and_brackets_both_outcomes(oracle):
    """a tap exactly on the depth-factor underflow distance (288.41 mm at sigma_d 20): BAND, and [lo, hi] holds the value
    with the tap skipped (full weight S*cf) and with it kept (weight ~1e-45)"""
    h, w = 9, 9
    depth = np.full((h, w), 1000.0, np.float32)
    bgr = np.full((h, w, 3), 50, np.uint8)
    st0 = oracle.jbf_stage(depth, bgr, 5, 70.0, 50.0, 20.0)
    assert not st0.band.any()
    avg = float(st0.avg32[4, 4])
    thr = float(np.sqrt(150.0 * np.log(2.0) * 800.0))                  # 288.4053...: (d - avg)^2 / (2 sigma_d^2) == 150 ln 2
    depth[4, 5] = np.float32(avg + thr)                                # the nearest float: inside the 2e-6 band (stage mode) in x
    st = oracle.jbf_stage(depth, bgr, 5, 70.0, 50.0, 20.0, avg_in=np.full((h, w), avg, np.float32))
    assert st.band[4, 4]
    assert st.hi[4, 4] > st.lo[4, 4] * 1.001                            # skipped vs kept differ by about 288 / 25
    near = depth.copy()
    near[4, 5] = np.float32(avg + thr * (1.0 + 1e-5))                   # 2e-5 in x beyond the point: r03's 1.5e-4 band held it, this one does not
    st1 = oracle.jbf_stage(near, bgr, 5, 70.0, 50.0, 20.0, avg_in=np.full((h, w), avg, np.float32))
    assert not st1.band[4, 4] and abs(st1.fin64[4, 4] - st.hi[4, 4]) < 0.5      # decided: skipped (full weight)
    far = depth.copy()
    far[4, 5] = np.float32(avg + 295.0)                                 # clearly beyond: skipped, no band
    st2 = oracle.jbf_stage(far, bgr, 5, 70.0, 50.0, 20.0, avg_in=np.full((h, w), avg, np.float32))
    assert not st2.band[4, 4] and abs(st2.fin64[4, 4] - st.hi[4, 4]) < 0.5


def test_k1_stage_reports_a_missing_or_spurious_average(oracle, frame):
    bgr, depth = frame(3, 64, 48)
    st = oracle.jbf_stage(depth, bgr)
    avg = st.avg32.copy()
    avg[10, 10] = np.nan                                                # weights exist but the implementation claims none
    r = oracle.stage_check(oracle.jbf_kernel(depth, bgr), oracle.jbf_stage(depth, bgr, avg_in=avg))
    assert r["mismatch"] == 1 and r["bad"][10, 10]
    none = np.zeros_like(depth)                                         # no valid depth anywhere: NOWEIGHT, average must be NaN
    st0 = oracle.jbf_stage(none, bgr)
    assert ((st0.flags & oracle.Stage.NOWEIGHT) != 0).all() and np.isnan(st0.avg32).all()
    assert not oracle.stage_check(np.zeros_like(depth), st0)["bad"].any()
    assert oracle.stage_check(np.ones_like(depth), st0)["bad"].all()


@pytest.mark.parametrize("case", ["synthetic", "flat-nan-quirk", "speckle"])
def test_k10_restatement_passes_its_own_stage_check(oracle, synth, frame, case):
    if case == "synthetic":
        bgr, depth = frame(12, 160, 120)
        K = synth.intrinsics(160, 120)
        pts = oracle.p2r_depth(depth, K)
        sp = oracle.dasp_segmentation(bgr, pts, 5, 6, K, 200.0, 40.0, 0.0, 1)[0]
        da = oracle.dasp_segmentation(bgr, pts, 5, 6, K, 100.0, 20.0, 200.0, 1)[0]
        rl, rd9 = oracle.ers_edge_refining(sp, da, depth)
    elif case == "flat-nan-quirk":
        rd9 = np.full((24, 40), 1024.0, np.float32)
        rd9[5:9, 7:30] = 0
        bgr = np.full((24, 40, 3), 9, np.uint8)
        rl = np.zeros((24, 40), np.int32)
    else:
        rng = np.random.default_rng(17)
        H, W = 51, 97
        rl = ((np.arange(W)[None, :] + 2) // 7 + 13 * ((np.arange(H)[:, None] + 3) // 5)).astype(np.int32)
        rd9 = (800 + 40 * rl + rng.normal(0, 30, (H, W))).astype(np.float32)
        rd9[rng.random((H, W)) < 0.25] = 0
        bgr = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        bgr[:, 40:60] = 77
    out = oracle.ers_enhance(rd9, bgr, rl)
    st = oracle.ers_stage(rd9, bgr, rl)
    r = oracle.stage_check(out, st)
    assert not r["bad"].any(), {k: v for k, v in r.items() if k not in ("bad", "rel", "avg_unchecked")}
    assert r["max_rel_strict"] < 2e-5 and r["avg_bound_frac_max"] < 1.0 and r["dev_bound_frac_max"] < 1.0
    if case == "flat-nan-quirk":
        assert np.isnan(out).sum() > 0 and np.array_equal(np.isnan(out), np.isnan(st.fin64))
        # the NaN class follows the deviation: with a deviation of one rounding error instead of exactly 0 there is no 0/0
        dev = st.dev32.copy()
        dev[np.isnan(out)] = 1e-4
        st2 = oracle.ers_stage(rd9, bgr, rl, avg_in=st.avg32, dev_in=dev)
        assert not np.isnan(st2.fin64[np.isnan(out)]).any()
        assert oracle.stage_check(out, st2)["bad_dev"] >= np.isnan(out).sum()      # ... and that deviation is rejected


def test_k1_stage_grid_pixels_allow_the_quantised_numerator(oracle):
    """Found by tools/stress_parity.py (seed 204, case 192; tests/golden/k1_grid_numerator.npz is a crop of it): a hole whose
    neighbours all differ from it by cd ~ 8e4 at sigma_c 20 has weights below ONE unit of the float32 denormal grid; the
    float32 code holds the single surviving weight as 1 unit and the product d * weight as round(d) units, so it returns
    round(d) -- 1849.0 where the binary64 value is 1848.53.  Such pixels are GRID: the interval brackets the quantisation
    of the weights (each tap independently within two units of its exact weight, see grid_extremes() in kde_oracle.c) and of
    the numerator's terms (half a unit each)."""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "k1_grid_numerator.npz"))
    win, ss, cs, ds = z["params"]
    out = oracle.jbf_kernel(z["depth"], z["bgr"], int(win), float(ss), float(cs), float(ds))
    st = oracle.jbf_stage(z["depth"], z["bgr"], int(win), float(ss), float(cs), float(ds))
    assert out[7, 0] == 1849.0 and out[10, 0] == 1853.0                      # whole millimetres: the quantised numerator
    for y in (7, 10):
        assert st.flags[y, 0] & oracle.Stage.GRID and st.lo[y, 0] < out[y, 0] < st.hi[y, 0]
        assert st.hi[y, 0] - st.lo[y, 0] < 12.0                              # ... the span of the few taps that may survive
    assert not oracle.stage_check(out, st)["bad"].any()


def test_k1_stage_sub_unit_weights_that_survive_by_double_rounding(oracle):
    """tools/stress_parity.py seed 208 case 4866 (crop: tests/golden/k1_subunit_double_rounding.npz): a hole at window 3 whose
    three valid neighbours have exact weights of 0.25 / 0.41 / 0.25 denormal units -- all below the float32 underflow point
    2^-150.  The float32 code still fills it: exp(-x) = 0.67 u rounds to 1 u, times S = 0.61 is 0.61 u and rounds to 1 u
    again, while the diagonal neighbours (S = 0.37) vanish; the output is round(d) of that one tap.  The stage check must
    not call this "no weight": such pixels are GRID, 0 and every value the surviving taps can produce are admissible."""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "k1_subunit_double_rounding.npz"))
    win, ss, cs, ds = z["params"]
    out = oracle.jbf_kernel(z["depth"], z["bgr"], int(win), float(ss), float(cs), float(ds))
    st = oracle.jbf_stage(z["depth"], z["bgr"], int(win), float(ss), float(cs), float(ds))
    assert out[4, 5] == 2691.0 and z["depth"][4, 5] == 0.0
    f = st.flags[4, 5]
    assert f & oracle.Stage.GRID and f & oracle.Stage.ZERO_OK and not f & (oracle.Stage.NOWEIGHT | oracle.Stage.MISMATCH)
    assert st.lo[4, 5] <= 2691.0 <= st.hi[4, 5]
    assert not oracle.stage_check(out, st)["bad"].any()
    # an implementation that flushes those weights (the tuned kernels' 2^24 scale: exact weight < 2^-150 -> 0) reports no
    # weight and writes 0: admissible as well
    avg = st.avg32.copy()
    avg[4, 5] = np.nan
    zero = out.copy()
    zero[4, 5] = 0.0
    st2 = oracle.jbf_stage(z["depth"], z["bgr"], int(win), float(ss), float(cs), float(ds), avg_in=avg)
    assert not oracle.stage_check(zero, st2)["bad"].any()


def test_k1_stage_open_decisions_on_both_sides_of_the_mean(oracle):
    """tools/stress_parity.py seed 501 case 38542 (crop: tests/golden/k1_mixed_open_decisions.npz; window 31, sigma_s 0.5, colour
    term off, sigma_d 70).  The spatial table underflows beyond r^2 = 52, so far taps enter with weight 1 (Q1) and the average
    (2489 mm) sits between a near surface (1435 mm) and a far one (3500 mm), 1009.4 mm = the depth-factor underflow distance
    from BOTH: one tap at 1479.95 mm lies 5e-9 (relative, in x) from the underflow point and one at 3498.83 mm 7.8e-5 beyond it.
    Under r03's 1.5e-4 decision band both were open and the float32 value (2136.88: near tap multiplied in, far tap skipped)
    lay outside the all-or-nothing interval [2133.30, 2135.16] -- the extremes are the MIXED decisions, which is why the
    interval is formed per tap.  Under the 2e-6 band of r04 only the near tap is open: the far one is decided (skipped, as the
    float32 code does) and the interval shrinks to the two outcomes of ONE tap."""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "k1_mixed_open_decisions.npz"))
    win, ss, cs, ds = z["params"]
    y, x = (int(v) for v in z["pixel"])
    out = oracle.jbf_kernel(z["depth"], z["bgr"], int(win), float(ss), float(cs), float(ds))
    st = oracle.jbf_stage(z["depth"], z["bgr"], int(win), float(ss), float(cs), float(ds))
    # (only the crop's centre pixel has its whole window inside the crop; there the HIP kernel had returned the same bits)
    assert abs(float(out[y, x]) - 2136.8796) < 1e-3 and out[y, x] == z["got"][y, x]
    assert st.flags[y, x] & oracle.Stage.BAND
    assert 2135.0 < st.lo[y, x] < 2135.3 and 2136.87 < st.hi[y, x] < 2136.89     # near tap skipped / multiplied in; far tap decided
    assert 2135.0 < st.fin64[y, x] < 2135.3               # binary64's own decision for the tap 5e-9 from the point
    assert not oracle.stage_check(out, st)["bad"].any()
    # the bracket is not a blank cheque: a value one more tap away on either side is rejected -- including 2133.3, which
    # r03's band admitted
    for delta in (+2.0, -3.5, -7.5):
        bad = out.copy()
        bad[y, x] += np.float32(delta)
        assert oracle.stage_check(bad, st)["bad"][y, x]


def test_k1_stage_mixed_open_decisions_bracket(oracle):
    """two taps ON the depth-factor underflow distance (to the nearest float: within 2e-6 in x), one above and one below the
    given average: the admissible values are those of all four decision combinations, and the extremes are the MIXED ones
    (upper tap skipped = full weight with lower tap multiplied in = weight ~ 0, and the reverse), not all-or-nothing."""
    h, w = 9, 9
    depth = np.full((h, w), 2000.0, np.float32)
    bgr = np.full((h, w, 3), 50, np.uint8)
    thr = float(np.sqrt(150.0 * np.log(2.0) * 2.0 * 70.0 * 70.0))      # 1009.4 mm at sigma_d 70
    avg = 2000.0
    depth[4, 5] = np.float32(avg + thr)
    depth[4, 3] = np.float32(avg - thr)
    st = oracle.jbf_stage(depth, bgr, 5, 70.0, 50.0, 70.0, avg_in=np.full((h, w), avg, np.float32))
    assert st.flags[4, 4] & oracle.Stage.BAND
    # 23 taps at 2000 with weight ~1 each, the two open taps with weight ~1 (skipped) or ~1e-45 (multiplied in)
    up_only, down_only = (23 * 2000.0 + depth[4, 5]) / 24.0, (23 * 2000.0 + depth[4, 3]) / 24.0
    assert st.hi[4, 4] > up_only - 0.5 and st.lo[4, 4] < down_only + 0.5
    assert abs(st.hi[4, 4] - up_only) < 1.0 and abs(st.lo[4, 4] - down_only) < 1.0      # spatial weights 0.9992 .. 1
    for v, ok in ((up_only, True), (down_only, True), (2000.0, True), (up_only + 3.0, False), (down_only - 3.0, False)):
        got = np.full((h, w), 2000.0, np.float32)
        got[4, 4] = np.float32(v)
        assert oracle.stage_check(got, st)["bad"][4, 4] == (not ok), v


def test_k1_stage_pass2_weights_below_the_underflow_point_that_survive(oracle):
    """tools/stress_parity.py seed 503 case 19859 (crop: tests/golden/k1_pass2_subunit_survivors.npz; window 5, sigma_s 0.5, colour
    term off, sigma_d 70): a hole whose left neighbours lie at 3000 mm and right neighbours at 1000 mm.  Its average (2000.2)
    is 1000 mm from every tap, x = 102 — every pass-2 weight S exp(-x) is 0.3 .. 0.5 denormal units, below the float32
    underflow point, yet exp(-x) alone is 4 units and its product with S = 0.135 rounds UP to one unit: the float32 code
    returns round(d) of the survivors (1000.0) where the exact weights give "no weight".  Such pixels are GRID in pass 2 as
    well: 0 and every value the surviving taps can produce are admissible."""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "k1_pass2_subunit_survivors.npz"))
    win, ss, cs, ds = z["params"]
    y, x = (int(v) for v in z["pixel"])
    out = oracle.jbf_kernel(z["depth"], z["bgr"], int(win), float(ss), float(cs), float(ds))
    st = oracle.jbf_stage(z["depth"], z["bgr"], int(win), float(ss), float(cs), float(ds))
    assert out[y, x] == 1000.0 and out[y, x] == z["got"][y, x] and z["depth"][y, x] == 0.0
    f = st.flags[y, x]
    assert f & oracle.Stage.GRID and f & oracle.Stage.ZERO_OK and not f & (oracle.Stage.NOWEIGHT | oracle.Stage.MISMATCH)
    assert st.fin64[y, x] == 0.0 and st.lo[y, x] <= 1000.0 <= st.hi[y, x]
    assert not oracle.stage_check(out, st)["bad"][y, x]
    zero = out.copy()
    zero[y, x] = 0.0                       # an implementation that flushes the sub-unit weights: admissible as well
    assert not oracle.stage_check(zero, st)["bad"][y, x]
    far = out.copy()
    far[y, x] = 5000.0                     # ... but not a value no tap of the window can produce
    assert oracle.stage_check(far, st)["bad"][y, x]


def test_envelope_cross_check_admits_zero_where_binary64_finds_no_weight(oracle):
    """tools/stress_parity.py seed 603 case 17338 (tests/golden/k1_zero_ok_without_band.npz; window 3, sigma 1 / 2 / 70, the first
    stress case that hit the tuned window-3 kernel in this regime).  A hole whose valid taps all differ in colour (colour factor
    underflowed: skipped) and lie ~1006 mm from the first-pass average: every pass-2 weight S exp(-103.3) is 0.4 .. 0.8 denormal
    units.  The float32 restatement keeps the taps whose product rounds UP to one unit and returns 1833.5; the exact weights are
    all below 2^-150, so binary64 -- and the tuned kernel, which flushes at an exact 2^-150 -- say "no weight": 0.  The stage-wise
    check has admitted both since r03 (GRID + ZERO_OK); the float32-envelope cross-check flagged the pixel ZERO_OK but, having
    neither BAND nor COND, compared it strictly.  Now 0 is admissible wherever the envelope says so -- and nothing else is."""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "k1_zero_ok_without_band.npz"))
    win, ss, cs, ds = z["params"]
    y, x = (int(v) for v in z["pixel"])
    ref, env = oracle.jbf_kernel(z["depth"], z["bgr"], int(win), float(ss), float(cs), float(ds), return_ill=True)
    assert ref[y, x] == 1833.5 and z["got"][y, x] == 0.0 and z["depth"][y, x] == 0.0
    assert env.flags[y, x] & oracle.Env.ZERO_OK and not env.flagged[y, x]
    st = oracle.jbf_stage(z["depth"], z["bgr"], int(win), float(ss), float(cs), float(ds))
    assert st.flags[y, x] & oracle.Stage.GRID and st.flags[y, x] & oracle.Stage.ZERO_OK and st.fin64[y, x] == 0.0
    assert not oracle.parity_check(z["got"], ref, env)["bad"].any()             # the GPU's output of that case: passes now
    assert not oracle.parity_check(ref, ref, env)["bad"].any()                  # ... and so does the float32 value itself
    other = ref.copy()
    other[y, x] = 1500.0                                                        # any third value is still rejected
    assert oracle.parity_check(other, ref, env)["bad"][y, x]
    st_zero = oracle.stage_check(z["got"], st)
    assert not st_zero["bad"][y, x]


def test_deviation_census_counts_every_class(oracle):
    """oracle.deviation_census (round 5): the END-TO-END distance from the float32 value, counted over all pixels whatever
    their class -- pixels beyond rtol, zero-mask differences in both directions, NaN-mask differences, and the same over a
    GRID map; parity_check carries it as `census`"""
    ref = np.array([[0.0, 1000.0, 2000.0, 3000.0, np.nan, 0.0, 500.0, 700.0]], np.float32)
    got = np.array([[5.0, 1000.05, 0.0, 3000.0, np.nan, 0.0, np.nan, 700.5]], np.float32)
    flagged = np.array([[0, 1, 1, 0, 0, 0, 0, 0]], bool)
    grid = np.array([[1, 0, 0, 0, 0, 0, 0, 1]], bool)
    c = oracle.deviation_census(got, ref, flagged=flagged, grid=grid, rtol=1e-4)
    assert c["n"] == 8
    # beyond rtol where both hold a number: 1000.05 vs 1000 (5e-5: inside), 700.5 vs 700 (7e-4: beyond, unflagged, GRID)
    assert (c["n_rel_gt_rtol"], c["n_rel_gt_rtol_flagged"], c["n_rel_gt_rtol_unflagged"]) == (1, 0, 1)
    assert (c["n_zero_mask_differs"], c["n_gained_zero"], c["n_lost_zero"]) == (2, 1, 1)      # 2000 -> 0 (flagged), 0 -> 5
    assert (c["n_zero_mask_differs_flagged"], c["n_zero_mask_differs_unflagged"]) == (1, 1)
    assert c["n_nan_mask_differs"] == 1 and abs(c["max_rel"] - 0.5 / 700) < 1e-6
    assert (c["grid_pixels"], c["grid_rel_gt_rtol"], c["grid_zero_mask_differs"]) == (2, 1, 1)
    same = oracle.deviation_census(ref, ref)
    assert same["n_rel_gt_rtol"] == 0 and same["n_zero_mask_differs"] == 0 and same["n_nan_mask_differs"] == 0 and same["max_rel"] == 0.0
    r = oracle.parity_check(got, ref, None, 1e-4, grid=grid)
    assert r["census"]["n_rel_gt_rtol"] == 1 and r["census"]["grid_pixels"] == 2
