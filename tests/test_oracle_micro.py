"""Hand-checkable micro-cases and independent pure-Python restatements that pin the CPU oracle.

The reference ships no tests or golden vectors, so the oracle is pinned by
(a) closed-form cases, (b) slow literal Python ports of the CUDA text written separately from
oracle/kde_oracle.c, on tiny inputs."""
import math
import os

import numpy as np
import pytest

F = np.float32


def test_spatial_table_values(oracle):
    t = oracle.spatial_table(5, 70.0)
    # JointBilateralFilter.cpp:31-40: exp(-(dx^2+dy^2)/(2*70^2))
    assert t[2, 2] == 1.0
    assert abs(t[0, 0] - math.exp(-8 / 9800)) < 1e-7      # corner 0.999184
    assert abs(t[0, 2] - math.exp(-4 / 9800)) < 1e-7      # edge   0.999592
    assert np.array_equal(t, t.T) and np.array_equal(t, t[::-1, ::-1])
    t7 = oracle.spatial_table(7, 30.0)
    assert abs(t7[0, 0] - math.exp(-18 / 1800)) < 1e-7


def _flat(h, w, d=1000.0, c=(10, 20, 30)):
    depth = np.full((h, w), d, F)
    bgr = np.empty((h, w, 3), np.uint8)
    bgr[...] = c
    return depth, bgr


def test_jbf_constant_depth_is_identity(oracle):
    depth, bgr = _flat(12, 16)
    out = oracle.jbf_kernel(depth, bgr)
    assert np.allclose(out, 1000.0, rtol=1e-6)


def test_jbf_all_invalid_is_zero(oracle):
    depth, bgr = _flat(9, 9, d=50.0)    # 50 is NOT valid (strict >)
    assert np.all(oracle.jbf_kernel(depth, bgr) == 0)
    depth[...] = 0
    assert np.all(oracle.jbf_kernel(depth, bgr) == 0)


def test_jbf_single_valid_pixel_fills_its_window(oracle):
    depth, bgr = _flat(11, 11, d=0.0)
    depth[5, 5] = 1234.5
    out = oracle.jbf_kernel(depth, bgr)
    exp = np.zeros_like(depth)
    exp[3:8, 3:8] = 1234.5          # the centre need not be valid: holes are filled (Q2)
    assert np.allclose(out, exp, rtol=1e-6)


def test_jbf_q1_underflow_discontinuity(oracle):
    """Q1: a depth factor that underflows to exactly 0 is NOT multiplied in, so an outlier further than
    sqrt(103.972*2*20^2) = 288.41 mm from the window average regains full weight."""
    def run(delta):
        depth, bgr = _flat(5, 5, d=1000.0)
        depth[2, 4] = 1000.0 + delta
        return oracle.jbf_kernel(depth, bgr)[2, 2]
    # 24 taps at 1000 and one at 1000+delta: the average is ~1000+delta/25
    near = run(250.0)     # |d - avg| = 240 < 288.41 -> weight exp(-72) ~ 0
    far = run(320.0)      # |d - avg| = 307 > 288.41 -> factor skipped -> weight ~1
    assert abs(near - 1000.0) < 1e-3
    inl = 24 * math.exp(-(320.0 / 25) ** 2 / 800)          # the 24 inliers keep their (small) depth factor
    assert abs(far - (1000.0 + 320.0 / (inl + 1))) < 0.05
    xT = 150 * math.log(2)
    assert 288.40 < math.sqrt(xT * 800) < 288.42


def test_jbf_sigma_zero_terms_are_skipped(oracle):
    depth, bgr = _flat(8, 8)
    depth[3, 3] = 2000.0
    bgr[3, 3] = (200, 200, 200)
    a = oracle.jbf_kernel(depth, bgr, 5, 70.0, 0.0, 0.0)     # both terms off: spatial-only average
    tab = oracle.spatial_table(5, 70.0).astype(np.float64)
    win = depth[1:6, 1:6].astype(np.float64)
    assert abs(a[3, 3] - (win * tab).sum() / tab.sum()) < 1e-2


def _jbf_python(depth, guide, w, ss, cs, ds):
    """literal port of JointBilateralFilter.cu:4-83 in float32 scalar Python (tiny inputs only)."""
    H, W = depth.shape
    r = w // 2
    S = np.empty((w, w), F)
    for i in range(w):
        for j in range(w):
            S[i, j] = F(math.exp(-F(F((j - r) ** 2) + F((i - r) ** 2)) / F(F(2.0) * F(ss * ss))))
    out = np.zeros((H, W), F)

    def ex(x):
        return F(math.exp(float(x)))   # double exp rounded once: within 1 ulp of expf

    for y in range(H):
        for x in range(W):
            taps = []
            for i in range(-r, r + 1):
                for j in range(-r, r + 1):
                    xj, yi = x + j, y + i
                    if 0 <= xj < W and 0 <= yi < H and depth[yi, xj] > 50.0:
                        cd = F(sum((float(guide[y, x, c]) - float(guide[yi, xj, c])) ** 2 for c in range(3)))
                        cf = ex(-cd / F(2 * F(cs * cs))) if cs != 0 else F(0)
                        f = F(1.0)
                        if S[i + r, j + r] != 0:
                            f = F(f * S[i + r, j + r])
                        if cf != 0:
                            f = F(f * cf)
                        taps.append((depth[yi, xj], f))
            wa, wt = F(0), F(0)
            for d, f in taps:
                wa = F(wa + F(d * f))
                wt = F(wt + f)
            if not wt > 0:
                continue
            wa = F(wa / wt)
            num, den = F(0), F(0)
            for d, f in taps:
                dd = F(d - wa)
                df = ex(-F(dd * dd) / F(F(2.0) * F(ds * ds))) if ds != 0 else F(0)
                g = f if df == 0 else F(f * df)
                num = F(num + F(d * g))
                den = F(den + g)
            out[y, x] = 0 if den == 0 else F(num / den)
    return out


@pytest.mark.parametrize("params", [(5, 70.0, 50.0, 20.0), (3, 2.0, 7.65, 20.0), (7, 3.0, 25.0, 0.0)])
def test_jbf_matches_independent_python_port(oracle, frame, params):
    bgr, depth = frame(3, 160, 120)
    cb = np.ascontiguousarray(bgr[40:58, 60:84])
    cd = np.ascontiguousarray(depth[40:58, 60:84])
    w, ss, cs, ds = params
    ref = _jbf_python(cd, cb, w, F(ss), F(cs), F(ds))
    got = oracle.jbf_kernel(cd, cb, w, ss, cs, ds)
    assert np.array_equal(ref == 0, got == 0)
    nz = ref != 0
    assert np.max(np.abs(got[nz] - ref[nz]) / ref[nz]) < 2e-6


def test_cv_bilateral_constant_and_rounding(oracle):
    _, bgr = _flat(10, 12, c=(7, 99, 250))
    assert np.array_equal(oracle.cv_bilateral(bgr), bgr)
    # an isolated bright pixel is pulled towards its neighbours but colour weights keep it mostly
    img = np.zeros((9, 9, 3), np.uint8)
    img[4, 4] = (30, 30, 30)
    out = oracle.cv_bilateral(img, 5, 30.0, 30.0)
    # centre weight 1; 12 neighbours (4 each at space2 = 1, 2, 4) of value 0 with L1 colour distance 90
    wsum = 1.0 + sum(4 * math.exp(s2 * (-0.5 / 900) + 90 ** 2 * (-0.5 / 900)) for s2 in (1, 2, 4))
    assert out[4, 4, 0] == round(30 / wsum) == 26
    assert np.all(out[0, 0] == 0)


def test_cv_bilateral_reflect101_border(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (8, 9, 3), dtype=np.uint8)
    pad = np.pad(img, ((2, 2), (2, 2), (0, 0)), mode="reflect")     # numpy 'reflect' == BORDER_REFLECT_101
    ref = oracle.cv_bilateral(pad)[2:-2, 2:-2]
    assert np.array_equal(oracle.cv_bilateral(img), ref)


def test_projective_to_real_formula(oracle, synth):
    K = synth.intrinsics(64, 48)
    K[0, 2], K[1, 2] = 31.9, 24.7            # Cx, Cy are TRUNCATED to int (DimensionConvertor.cpp:8-9)
    depth = np.linspace(0, 4000, 64 * 48, dtype=F).reshape(48, 64)
    p = oracle.p2r_depth(depth, K)
    fx = F(K[0, 0])
    y, x = np.mgrid[0:48, 0:64]
    ex = ((x.astype(F) - F(31)) / fx) * depth       # subtract, divide, multiply
    ey = ((F(24) - y.astype(F)) / fx) * depth
    assert np.array_equal(p["x"], ex.astype(F)) and np.array_equal(p["y"], ey.astype(F))
    assert np.array_equal(p["z"], depth)
    back = oracle.r2p(p, K)
    m = depth >= 1
    assert np.allclose(back["x"][m], x[m], atol=1e-2) and np.allclose(back["y"][m], y[m], atol=1e-2)
    assert np.all(back["x"][~m] == -1) and np.all(back["y"][~m] == -1)
    # the float3 overload applies the same functor to (x,y,z) triples
    trip = np.stack([x.astype(F), y.astype(F), depth], -1)
    assert np.array_equal(oracle.p2r_points(trip, K).view(F), p.view(F))


def test_buffer2d_update_rule(oracle):
    b = oracle.Buffer2D(4, 2)
    assert np.all(b.depth_map() == 0) and np.all(b.weight_map() == 0)
    f1 = np.array([[1000, 40, 2000, 0], [500, 500, 500, 500]], F)
    b.update(f1)
    assert np.array_equal(b.depth_map(), np.array([[1000, 0, 2000, 0], [500] * 4], F))
    assert np.array_equal(b.weight_map(), np.array([[1, 0, 1, 0], [1] * 4], F))
    f2 = np.array([[1004, 1000, 2100, 60], [504, 505.5, 495, 100]], F)
    b.update(f2)
    # gate: (float)abs((int)ref - (int)d) < d*0.01 ; update ((ref*(w+1)) + d*w)/(2w+1), w++
    d, w = b.depth_map(), b.weight_map()
    assert d[0, 0] == F((F(1000 * 2.0) + F(1004 * 1.0)) / F(3.0)) and w[0, 0] == 2
    assert d[0, 1] == 1000 and w[0, 1] == 1          # first valid sample seeds the cell
    assert d[0, 2] == 2000 and w[0, 2] == 1          # |2000-2100| = 100 >= 21 -> rejected
    assert d[0, 3] == 60 and w[0, 3] == 1
    # row 1: |500-504| = 4 < 5.04 ok; |(int)500-(int)505.5| = 5 < 5.055 ok (int truncation); 5 < 4.95 no; 400 < 1 no
    assert w[1, 0] == 2 and w[1, 1] == 2 and w[1, 2] == 1 and w[1, 3] == 1
    b.insert_depth(f1)
    assert np.array_equal(b.depth_map(), f1) and np.all(b.weight_map() == 1)
    b.insert_float2(np.stack([f2, f2 * 0], -1))
    assert np.array_equal(b.weight_map(), np.array([[0] * 4, [1] * 4], F))   # w = row index (sic)


# ---- DASP: independent literal port of calculateLD on a tiny image -----------------------------
def _tree_argmin(dist, lab):
    dist, lab = list(dist), list(lab)
    for step in (8, 4, 2, 1):
        for t in range(step):
            if dist[t] > dist[t + step]:
                dist[t], lab[t] = dist[t + step], lab[t + step]
    return dist[0], lab[0]


def test_tree_argmin_tie_break_is_not_lowest_index():
    d = [5.0] * 16
    d[1] = d[4] = 1.0
    assert _tree_argmin(d, list(range(16)))[1] == 4         # Q4
    d = [5.0] * 16
    d[3] = d[9] = 1.0
    assert _tree_argmin(d, list(range(16)))[1] == 9          # +8 partner of slot 1 wins, then 1 beats 3? no: 9 folds into slot 1
    nan = float("nan")
    d = [nan] + [2.0] * 15
    assert math.isnan(_tree_argmin(d, list(range(16)))[0])   # a NaN in the low slot is never replaced


def test_dasp_calculate_ld_matches_python_port(oracle, frame, synth):
    bgr, depth = frame(5, 64, 48)
    K = synth.intrinsics(64, 48)
    pts = oracle.p2r_depth(depth, K)
    rows, cols = 3, 4
    ld0, mean, centers = oracle.dasp_steps(bgr, pts, rows, cols)
    wx, wy = 64 // cols, 48 // rows
    assert np.array_equal(ld0["l"], (np.arange(48)[:, None] // wy) * cols + np.arange(64)[None, :] // wx)
    cs, ss, ds = F(100.0), F(20.0), F(200.0)
    labels, ld = oracle.dasp_calculate_ld(bgr, pts, rows, cols, ld0, mean, centers, cs, ss, ds)
    sumS = F(F(ss + cs) + ds)
    kc, ks, kd = F(F(cs / sumS) ** 2), F(F(ss / sumS) ** 2), F(F(ds / sumS) ** 2)
    win2 = F(F(F(wx + wy) / F(2.0)) ** 2)
    for (y, x) in [(0, 0), (5, 17), (23, 31), (47, 63), (24, 16), (30, 40), (11, 50)]:
        l0 = int(ld0["l"][y, x])
        ccx, ccy = l0 % cols, l0 // cols
        dist, lab = [], []
        for ty in range(4):
            for tx in range(4):
                rx, ry = ccx - 2 + tx, ccy - 2 + ty
                if 0 <= rx < cols and 0 <= ry < rows:
                    m = mean[ry * cols + rx]
                    c = bgr[y, x].astype(F)
                    cdist = F(F(F((c[0] - F(m["r"])) ** 2) + F((c[1] - F(m["g"])) ** 2)) + F((c[2] - F(m["b"])) ** 2))
                    sd = F(np.sqrt(F(F(F(x - m["x"]) ** 2) + F(F(y - m["y"]) ** 2))) * win2)
                    z, cz = pts["z"][y, x], centers["z"][ry * cols + rx]
                    dd = F(abs(F(z - cz))) if (z > 50 and cz > 50) else F(0)
                    dist.append(F(F(F(cdist * kc) + F(sd * ks)) + F(dd * kd)))
                    lab.append(ry * cols + rx)
                else:
                    dist.append(ld0["d"][y, x])
                    lab.append(l0)
        d, l = _tree_argmin(dist, lab)
        if pts["z"][y, x] < 50:
            d, l = F(0), -1
        assert labels[y, x] == l and ld["l"][y, x] == l and ld["d"][y, x] == d


def test_dasp_sample_clusters_b_channel_quirk(oracle, frame, synth):
    bgr, depth = frame(5, 64, 48)
    pts = oracle.p2r_depth(depth, synth.intrinsics(64, 48))
    _, mean, centers = oracle.dasp_steps(bgr, pts, 3, 4)
    for k in range(12):
        x, y = int(mean["x"][k]), int(mean["y"][k])
        cx, cy = (k % 4) * 16 + 8, (k // 4) * 16 + 8
        assert cx - 2 <= x <= cx + 1 and cy - 2 <= y <= cy + 1
        assert mean["r"][k] == bgr[y, x, 0] and mean["g"][k] == bgr[y, x, 1]
        assert mean["b"][k] == (int(bgr[y, x, 0]) + 2) % 256       # sic: DepthAdaptiveSuperpixel.cu:159
        assert centers[k] == pts[y, x]


def test_dasp_geometry_guard(oracle, frame, synth):
    bgr, depth = frame(5, 64, 48)
    pts = oracle.p2r_depth(depth, synth.intrinsics(64, 48))
    with pytest.raises(ValueError):
        oracle.dasp_segmentation(bgr, pts, 15, 20, synth.intrinsics(64, 48), 1.0, 1.0, 1.0, 1)   # window 3x3 < 4


# ---- K9: the literal in-place raster-order port agrees with snapshot semantics on isolated edges ----
def _edge_refining_inplace(cl, labels, depth, window=7):
    """sequential raster-order execution of EdgeRefinedSuperpixel.cu:4-102 (one legal schedule of the
    racy kernel when no two sources touch the same cells)."""
    H, W = labels.shape
    L, D = labels.copy(), depth.copy()
    for dirn in (0, 1):
        ln = W if dirn == 0 else H
        for y in range(H):
            for x in range(W):
                pos = x if dirn == 0 else y
                def at(a, k):
                    return a[y, k] if dirn == 0 else a[k, x]
                def put(a, k, v):
                    if dirn == 0:
                        a[y, k] = v
                    else:
                        a[k, x] = v
                if not pos + 1 < ln or at(L, pos) == at(L, pos + 1):
                    continue
                cur = at(cl, pos)
                tgt, dist = cur, 0
                while (pos - dist >= 0 or pos + dist < ln) and tgt == cur and dist <= window // 2:
                    tp = pos - dist
                    if tp >= 0:
                        tgt = at(cl, tp)
                    if tgt != cur:
                        lab = at(L, pos + 1)
                        for i in range(tp + 1, pos + 1):
                            put(L, i, lab)
                            if abs(F(at(D, i) - at(D, i + 1))) > F(at(D, i) * F(0.1)):
                                put(D, i, F(0))
                        break
                    tp = pos + dist
                    if tp < ln:
                        tgt = at(cl, tp)
                    if tgt != cur:
                        lab = at(L, pos)
                        for i in range(pos + 1, tp):
                            put(L, i, lab)
                            if abs(F(at(D, i) - at(D, i - 1))) > F(at(D, i) * F(0.1)):
                                put(D, i, F(0))
                        break
                    dist += 1
    return L, D


def test_edge_refining_isolated_edges_match_inplace_port(oracle):
    H, W = 12, 40
    cl = np.zeros((H, W), np.int32)
    dl = np.zeros((H, W), np.int32)
    depth = np.full((H, W), 1000, F)
    cl[:, 12:] = 1                       # colour edge at x=12
    dl[:, 10:] = 1                       # depth edge at x=10: 2 px left of the colour edge -> right branch
    depth[:, 10:] = 1500
    cl[:, 30:] = 2                       # colour edge at x=30
    dl[:, 32:] = 2                       # depth edge 2 px right of it -> left branch
    depth[:, 32:] = 800
    ref_l, ref_d = _edge_refining_inplace(cl, dl, depth)
    got_l, got_d = oracle.ers_edge_refining(cl, dl, depth)
    assert np.array_equal(got_l, ref_l) and np.array_equal(got_d, ref_d)
    # the depth boundary snapped onto the colour boundary
    assert np.all(got_l[:, 10:12] == 0) and np.all(got_l[:, 30:32] == 2)
    # relabelled pixels whose depth jumps > 10 % are zeroed; the right branch cascades
    assert np.all(got_d[:, 10:12] == 0)
    assert np.all(got_d[:, 31] == 0) and np.all(got_d[:, 30] == 1500)   # left branch does not cascade


def test_edge_refining_no_colour_edge_nearby_is_noop(oracle):
    cl = np.zeros((8, 20), np.int32)
    dl = np.zeros((8, 20), np.int32)
    dl[:, 10:] = 1
    depth = np.full((8, 20), 900, F)
    got_l, got_d = oracle.ers_edge_refining(cl, dl, depth)
    assert np.array_equal(got_l, dl) and np.array_equal(got_d, depth)


def test_enhance_mutating_sigma_and_flat_patch(oracle):
    """Q6: on an exactly flat patch dev == 0, the colour sigma decays 0.3x per valid tap until
    2*sigma^2 underflows: cd == 0 taps then give 0/0 = NaN (49 valid taps -> NaN), windows with
    fewer than 47 valid taps stay finite."""
    depth, bgr = _flat(12, 12, d=1024.0)     # power of two: the weighted average is exact
    labels = np.zeros((12, 12), np.int32)
    out = oracle.ers_enhance(depth, bgr, labels)
    inner = out[3:9, 3:9]
    assert np.all(np.isnan(inner))
    assert np.all(out[0, :] == 1024.0) and np.all(out[:, 0] == 1024.0)
    # with deviation the adaptive sigma floors the decay and everything stays finite
    rng = np.random.default_rng(1)
    noisy = (depth + rng.normal(0, 3, depth.shape)).astype(F)
    assert np.all(np.isfinite(oracle.ers_enhance(noisy, bgr, labels)))


def test_mean_3d_error(oracle):
    a = np.zeros((2, 3, 3), F)
    b = np.zeros((2, 3, 3), F)
    a[..., 2] = 1000
    b[..., 2] = 1003
    b[0, 0, 2] = 10        # invalid in truth -> excluded
    a[1, 1] = (4, 0, 1000)
    b[1, 1] = (0, 3, 1000)
    e, n = oracle.mean_3d_error(a, b)
    assert n == 5 and abs(e - (3 * 4 + 5) / 5) < 1e-6


# ---- the parity envelope (oracle.Env / okde_env): what a faithful float32 evaluation may return ---------------------

def _denormal_weight_scene():
    """a hole whose valid neighbours all differ from it by (60,60,60): at sigma_c = 7.65 every colour factor is
    exp(-10800/117.045) = 2^-133, a float32 denormal -- the reference still averages with these weights"""
    h, w = 9, 9
    depth = (1000.0 + 10.0 * np.arange(h * w, dtype=F).reshape(h, w)).astype(F)
    bgr = np.full((h, w, 3), 60, np.uint8)
    depth[4, 4] = 0.0
    bgr[4, 4] = 0
    return depth, bgr


def test_envelope_contains_the_float32_value_everywhere(oracle, frame):
    bgr, depth = frame(2, 96, 64)
    for cfg in ((5, 70.0, 50.0, 20.0), (11, 3.0, 7.65, 20.0), (7, 5.0, 20.0, 4.0)):
        ref, env = oracle.jbf_kernel(depth, bgr, *cfg, return_ill=True)
        nz = ref != 0
        assert np.all(ref[nz] >= env.lo[nz] * (1 - 1e-12)) and np.all(ref[nz] <= env.hi[nz] * (1 + 1e-12))
        assert np.all((env.flags[~nz] & oracle.Env.ZERO_OK) != 0)
        assert np.all(env.lo <= env.hi)
        # unflagged pixels are the ones the strict 1e-4 test is meaningful for: their envelope is narrow
        un = ~env.flagged & nz
        assert np.all((env.hi - env.lo)[un] <= 5.0001e-5 * np.abs(ref[un]))
        assert (env.flags & 1).sum() == 0                      # round 1's "denominator < 1e-30" class is retired


def test_envelope_flags_a_tap_on_the_q1_jump(oracle):
    """a tap whose depth term sits on the expf-underflow jump (|d - avg| = 288.41 mm at sigma_d 20): both outcomes
    (factor ~1e-45, or skipped = 1) must be inside the envelope, and the pixel must be flagged BAND"""
    depth, bgr = _flat(11, 11)
    depth[5, 5] = 1000.0 + 288.41 * 25.0 / 24.0     # avg moves by delta/25 in a 5x5 window of equal weights
    ref, env = oracle.jbf_kernel(depth, bgr, return_ill=True)
    ys, xs = np.nonzero(env.flags & oracle.Env.BAND)
    assert len(ys) > 0 and np.all(np.abs(ys - 5) <= 2) and np.all(np.abs(xs - 5) <= 2)
    p = (ys[0], xs[0])
    assert env.hi[p] - env.lo[p] > 1.0                 # skipped: the outlier pulls the result by ~ delta/25 = 12 mm
    assert abs(env.lo[p] - 1000.0) < 0.01 and env.hi[p] > 1010.0   # factor 1e-45: outlier ignored; skipped: it counts fully


def test_envelope_spans_the_whole_sawtooth_next_to_a_q1_flip(oracle):
    """Found by the stress tool (seed 777, case 700): window 19, sigma_d 5.  One tap sits at x = (d - avg)^2 / 50 = 103.957,
    i.e. 1.4e-4 below the underflow point, and the near taps' weights exp(-(d - avg)^2 / 50) move by 0.3 % per 0.02 mm of
    the average.  As the average falls from avg + eps to avg - eps the result slides from 905.3 down to ~898 and then jumps
    to ~1255 when the tap's factor underflows and is skipped.  Sampling avg - eps, avg, avg + eps with the decision
    following the average misses the lower part of the slide; the threshold band must be widened by what eps does to x
    (oracle thr_band).  The three values the GPU kernels returned (packed, scalar, generic) lie on the slide."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "k1_band_sawtooth.npz"))
    win, ss, cs, ds = z["params"]
    ref, env = oracle.jbf_kernel(z["depth"], z["bgr"], int(win), float(ss), float(cs), float(ds), return_ill=True)
    c = (9, 9)
    assert env.flags[c] & oracle.Env.BAND and env.flags[c] & oracle.Env.COND
    assert env.lo[c] < 899.0 and env.hi[c] > 1255.0
    for v in z["observed"]:
        assert env.lo[c] <= v <= env.hi[c]
    assert abs(ref[c] - 901.3955) < 1e-3


def test_envelope_allows_the_summation_error_of_every_tap_that_can_round(oracle):
    """Found by the stress tool (seeds 11 / 22 / 44): window 19, sigma_s 0.5 (most of the spatial table underflows, so
    those taps carry the bare colour factor), quantised colours, sigma_d 5.  345 valid taps, 160 of them heavy enough
    to round the running sum, but a participation ratio (sum w)^2 / sum w^2 of only 41: the float32 restatement's own
    first-pass average is 29 ulps off the exact one, beyond the 4 + 2.5 sqrt(41) = 20 the envelope allowed, and a tuned
    kernel that sums in another order landed on the other side (2377.14 vs 2375.72; exact 2376.33).  The envelope now
    moves the average by the summation bound over the taps that can round, 4 + n_sig / 2 ulps (oracle n_significant)."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "k1_sum_bound.npz"))
    win, ss, cs, ds = z["params"]
    ref, env = oracle.jbf_kernel(z["depth"], z["bgr"], int(win), float(ss), float(cs), float(ds), return_ill=True)
    c = (9, 9)
    assert env.flags[c] & oracle.Env.COND
    assert abs(ref[c] - 2375.724) < 1e-2
    assert env.lo[c] < ref[c] - 1.0 and env.hi[c] > z["observed"][0] + 1.0      # both sides of the exact value, with room
    # the bound itself: a float32 running sum of n positive terms is off by at most (n - 1) / 2 ulps
    rng = np.random.default_rng(0)
    for n in (64, 160, 361):
        w = rng.choice(np.array([1.0, 0.8148102, 0.3678794, 0.0307], F), n)
        acc = F(0.0)
        for v in w:
            acc = F(acc + v)
        exact = float(w.astype(np.float64).sum())
        assert abs(float(acc) - exact) <= 0.5 * n * exact * 2.0 ** -23


def test_envelope_keeps_denormal_range_weights(oracle):
    depth, bgr = _denormal_weight_scene()
    ref, env = oracle.jbf_kernel(depth, bgr, 5, 3.0, 7.65, 20.0, return_ill=True)
    # the float32 reference fills the hole from weights of 2^-133 * S (quantised to the denormal grid) ...
    assert 1000.0 < ref[4, 4] < 1800.0
    # ... and the binary64 evaluation of the same formula keeps them unquantised: both are in the envelope
    assert env.lo[4, 4] <= ref[4, 4] <= env.hi[4, 4] and env.lo[4, 4] > 1000.0
    assert (env.flags[4, 4] & oracle.Env.ZERO_OK) == 0         # 0 (all weights flushed) is NOT an admissible answer


def test_envelope_of_k10_admits_nan_only_next_to_the_flat_patch_quirk(oracle):
    depth, bgr = _flat(24, 40, d=1024.0, c=(9, 9, 9))
    lab = np.zeros((24, 40), np.int32)
    with oracle.ers_flags((24, 40)) as env:
        ref = oracle.ers_enhance(depth, bgr, lab)
    assert np.isnan(ref).sum() > 0
    assert np.all((env.flags[np.isnan(ref)] & oracle.Env.NAN_OK) != 0)
    rng = np.random.default_rng(3)
    d2 = (depth + rng.normal(0, 5, depth.shape)).astype(F)
    with oracle.ers_flags((24, 40)) as env2:
        ref2 = oracle.ers_enhance(d2, bgr, lab)
    assert not np.isnan(ref2).any() and ((env2.flags & oracle.Env.NAN_OK) != 0).sum() == 0
    nz = ref2 != 0
    assert np.all(ref2[nz] >= env2.lo[nz] * (1 - 1e-12)) and np.all(ref2[nz] <= env2.hi[nz] * (1 + 1e-12))


# ---- K10: a literal scalar port of depthmap_enhancement written from the CUDA text, independent of kde_oracle.c ----
def _enhance_port(rd, bgr, labels, spatial, window=7, color_sigma_in=50.0, depth_sigma=70.0):
    """EdgeRefinedSuperpixel.cu:104-205 in numpy float32 scalars, reading the phase-start depth (D3).  np.exp on a
    float32 is not libm's expf to the last ulp, so the comparison below allows 2e-6."""
    H, W = rd.shape
    hw = window // 2
    out = np.zeros((H, W), F)
    c = bgr.astype(np.float32)
    with np.errstate(all="ignore"):
        for y in range(H):
            for x in range(W):
                cs = F(color_sigma_in)
                wavg, weight = F(0), F(0)
                for i in range(-hw, hw + 1):
                    for j in range(-hw, hw + 1):
                        xj, yi = x + j, y + i
                        if 0 <= xj < W and 0 <= yi < H and rd[yi, xj] > F(50) and labels[y, x] == labels[yi, xj]:
                            d0, d1, d2 = c[y, x] - c[yi, xj]
                            cd = F(F(d0 * d0) + F(d1 * d1)) + F(d2 * d2)
                            cf = np.exp(F(-cd / F(F(2) * F(cs * cs)))) if cs != 0 else F(0)
                            f = F(1)
                            s = spatial[i + hw, j + hw]
                            if s != 0:
                                f = F(f * s)
                            if cf != 0:
                                f = F(f * cf)
                            wavg = F(wavg + F(rd[yi, xj] * f))
                            weight = F(weight + f)
                if not weight > 0:
                    continue
                wavg = F(wavg / weight)
                count, dev = 0, F(0)
                for i in range(-hw, hw + 1):
                    for j in range(-hw, hw + 1):
                        xj, yi = x + j, y + i
                        if 0 <= xj < W and 0 <= yi < H and rd[yi, xj] > F(50) and labels[y, x] == labels[yi, xj]:
                            dev = F(dev + abs(F(rd[yi, xj] - wavg)))
                            count += 1
                if count:
                    dev = F(dev / F(count))
                num, den = F(0), F(0)
                for i in range(-hw, hw + 1):
                    for j in range(-hw, hw + 1):
                        xj, yi = x + j, y + i
                        if 0 <= xj < W and 0 <= yi < H and rd[yi, xj] > F(50):
                            d0, d1, d2 = c[y, x] - c[yi, xj]
                            cd = F(F(d0 * d0) + F(d1 * d1)) + F(d2 * d2)
                            cf = F(0)
                            if cs != 0:
                                a = F(5.0 * float(dev) / float(F(wavg * wavg)))          # double arithmetic, then float
                                cs = a if a > F(cs * F(0.3)) else F(cs * F(0.3))
                                cf = np.exp(F(-cd / F(F(2) * F(cs * cs))))
                            dd = F(rd[yi, xj] - wavg)
                            df = np.exp(F(-F(dd * dd) / F(F(2) * F(F(depth_sigma) * F(depth_sigma))))) if depth_sigma != 0 else F(0)
                            f = F(1)
                            s = spatial[i + hw, j + hw]
                            if s != 0:
                                f = F(f * s)
                            if cf != 0:                     # NaN != 0 is True: the 0/0 quirk multiplies the NaN in
                                f = F(f * cf)
                            if df != 0:
                                f = F(f * df)
                            num = F(num + F(rd[yi, xj] * f))
                            den = F(den + f)
                out[y, x] = F(0) if den == 0 else F(num / den)
    return out


def test_enhance_matches_independent_python_port(oracle):
    rng = np.random.default_rng(11)
    H, W = 13, 17
    labels = ((np.arange(W)[None, :] // 6) + 3 * (np.arange(H)[:, None] // 5)).astype(np.int32)
    depth = (900 + 250 * labels + rng.normal(0, 6, (H, W))).astype(F)
    depth[rng.random((H, W)) < 0.08] = 0                      # holes: ranks differ from tap positions
    depth[2:6, 2:9] = 1536.0                                  # an exactly flat stretch inside one label
    bgr = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    bgr[:, 8:12] = (40, 80, 120)                              # equal colours: cd == 0 taps
    labels[6, 6] = 77                                         # an isolated label (single same-label tap)
    spatial = oracle.spatial_table(7, 30.0)
    want = _enhance_port(depth, bgr, labels, spatial)
    got = oracle.ers_enhance(depth, bgr, labels)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got == 0, want == 0)
    fin = np.isfinite(want) & (want != 0)
    assert fin.sum() > 150
    assert np.max(np.abs(got[fin] - want[fin]) / np.abs(want[fin])) < 2e-6


def test_fastdiv24_formula_is_exact_on_its_domain():
    """csrc/kde_device_math.h make_fastdiv24 / fastdiv24 (K7's x / wx, y / wy, label / cols; K1's workgroup -> tile):
    with l = ceil(log2 d), m = floor(2^(24+l) / d) + 1, floor(x * m / 2^(24+l)) == x // d for every 0 <= x < 2^24.
    Checked here on the multiples of d and their neighbours (where a wrong quotient would show first), on random x, and
    exhaustively for small d."""
    rng = np.random.default_rng(0)

    def magic(d):
        l = 0
        while (1 << l) < d:
            l += 1
        return ((1 << (24 + l)) // d + 1, 24 + l)

    ds = list(range(1, 300)) + [int(v) for v in rng.integers(300, 1 << 24, 300)] + [(1 << 24) - 1, 1 << 23, (1 << 23) + 1]
    for d in ds:
        m, sh = magic(d)
        assert m < (1 << 32)
        k = np.arange(0, (1 << 24) // d + 1, max(1, ((1 << 24) // d) // 4096), dtype=np.uint64)
        xs = np.concatenate([k * d, k * d + (d - 1), np.maximum(k * d, 1) - 1, rng.integers(0, 1 << 24, 4096).astype(np.uint64),
                             np.array([0, (1 << 24) - 1], np.uint64)])
        xs = xs[xs < (1 << 24)]
        assert np.array_equal((xs * np.uint64(m)) >> np.uint64(sh), xs // np.uint64(d)), d
    for d in (1, 2, 3, 5, 7, 20, 96):
        m, sh = magic(d)
        xs = np.arange(1 << 24, dtype=np.uint64)
        assert np.array_equal((xs * np.uint64(m)) >> np.uint64(sh), xs // np.uint64(d)), d
