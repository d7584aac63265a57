"""K0 / K1 / Process parity: HIP (through the C ABI) vs the CPU oracle.

Bar: exact equality for K0's u8 output; K1 is checked STAGE BY STAGE (conftest.assert_k1_stagewise,
oracle/kde_oracle.h okde_stage): the GPU's own first-pass average against binary64 within the float32 first-order
bound, then the final value at 1e-4 against pass 2 evaluated in binary64 FROM that average, identical zero mask.
No pixel is excluded; only pixels with a tap ON a Q1 decision at that average (BAND, <= 0.3 % of a frame at BASELINE's
sizes) are held to the interval of both outcomes instead."""
import os

import numpy as np
import pytest

from conftest import assert_depth_close, assert_k1_stagewise, assert_mrf_close
from gpu_util import dev, host

pytestmark = pytest.mark.gpu

RTOL = 1e-4


@pytest.fixture(scope="module")
def F(torch_cuda):
    from kinectdepthmapenhancement_amd import filters
    return filters


def params(F, w=5, ss=70.0, cs=50.0, ds=20.0, pre=1, pk=5, pc=30.0, ps=30.0):
    p = F.JointBilateralFilter.default_params()
    p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma = w, ss, cs, ds
    p.presmooth, p.presmooth_kernel_size, p.presmooth_sigma_color, p.presmooth_sigma_spatial = pre, pk, pc, ps
    return p


def test_presmooth_k0_is_bit_exact(torch_cuda, F, oracle, frame, color_fixture):
    # (66 x 34 and 98 x 50: the halo of the last interior tile ends exactly on the last pixel of the buffer, whose 4-byte
    #  read the staging code must split; 33 x 9: a single partial tile)
    for bgr in (color_fixture, frame(2)[0], frame(4, 70, 50)[0], frame(4, 33, 9)[0], frame(5, 66, 34)[0], frame(6, 98, 50)[0]):
        h, w, _ = bgr.shape
        jbf = F.JointBilateralFilter(w, h)
        out = torch_cuda.empty((1, h, w, 3), dtype=torch_cuda.uint8, device="cuda")
        jbf.presmooth_batch(dev(torch_cuda, bgr[None]), out)
        assert np.array_equal(host(out)[0], oracle.cv_bilateral(bgr, 5, 30.0, 30.0))


def test_presmooth_tile_walks_agree(torch_cuda, F, oracle, frame):
    """K0 walks its tiles in XCD bands while a launch's input fits the L2s (<= 32 MiB) and in XCD bands of RUNS of four adjacent
    tiles per workgroup beyond (r05): a batch of 40 640x480 frames (36.9 MB, run walk) must give, frame by frame, the bytes of
    the single-frame launches (band walk), and frames 0 and 39 the oracle's; a ragged size puts partial tiles, a tile count
    that is no multiple of four and a remainder after the full runs into both walks"""
    for (w, h, n) in ((640, 480, 40), (637, 479, 40)):
        base = np.stack([frame(20 + i, w, h)[0] for i in range(4)])
        bgr = np.concatenate([np.roll(base, i, axis=2) for i in range(n // 4)])      # 40 distinct frames
        assert bgr.nbytes > 32 << 20
        big = F.JointBilateralFilter(w, h, max_batch=n)
        one = F.JointBilateralFilter(w, h)
        src = dev(torch_cuda, bgr)
        out = torch_cuda.empty_like(src)
        big.presmooth_batch(src, out)
        single = torch_cuda.empty_like(src)
        for i in range(n):
            one.presmooth_batch(src[i:i + 1], single[i:i + 1])
        assert torch_cuda.equal(out, single)
        for i in (0, n - 1):
            assert np.array_equal(host(out[i]), oracle.cv_bilateral(bgr[i], 5, 30.0, 30.0))


@pytest.mark.parametrize("ksize,sc,ss", [(3, 10.0, 5.0), (7, 60.0, 2.0), (9, 25.0, 25.0), (0, 30.0, 1.7)])
def test_presmooth_other_kernel_sizes(torch_cuda, F, oracle, frame, ksize, sc, ss):
    bgr = frame(6, 96, 64)[0]
    jbf = F.JointBilateralFilter(96, 64, params(F, pk=ksize, pc=sc, ps=ss))
    out = torch_cuda.empty((1, 64, 96, 3), dtype=torch_cuda.uint8, device="cuda")
    jbf.presmooth_batch(dev(torch_cuda, bgr[None]), out)
    assert np.array_equal(host(out)[0], oracle.cv_bilateral(bgr, ksize, sc, ss))


@pytest.mark.parametrize("ksize,sc,ss", [(11, 30.0, 30.0), (13, 20.0, 4.0), (0, 30.0, 4.0), (31, 40.0, 9.0)])
def test_presmooth_radius_above_four_uses_the_generic_kernel(torch_cuda, F, oracle, frame, ksize, sc, ss):
    """cv::gpu::bilateralFilter takes any kernel size (ksize 0 -> radius round(1.5 sigma_s)); radii > 4 have no tuned
    kernel and run the generic one: same bytes as the oracle, including ragged sizes and batches"""
    for size, seed in (((96, 64), 6), ((33, 9), 4)):
        w, h = size
        bgr = np.stack([frame(seed, w, h)[0], frame(seed + 1, w, h)[0]])
        jbf = F.JointBilateralFilter(w, h, params(F, pk=ksize, pc=sc, ps=ss), max_batch=2)
        out = torch_cuda.empty((2, h, w, 3), dtype=torch_cuda.uint8, device="cuda")
        jbf.presmooth_batch(dev(torch_cuda, bgr), out)
        for i in range(2):
            assert np.array_equal(host(out)[i], oracle.cv_bilateral(bgr[i], ksize, sc, ss))


def test_filtered_host_never_reads_a_callers_buffer(torch_cuda, F, oracle, synth):
    """filter_batch(n > max_batch, out=caller buffer) followed by getFiltered_Host() used to copy n frames out
    of the caller's pointer into a pinned buffer sized for max_batch.  The host getter mirrors the object's own
    Filtered_Device only."""
    bgr, depth = synth.make_batch(60, 4, 64, 48)
    jbf = F.JointBilateralFilter(64, 48, params(F, pre=0), max_batch=1)
    jbf.filter_batch(dev(torch_cuda, depth[:1]), dev(torch_cuda, bgr[:1]), jbf.getFiltered_Device()[None])
    own = jbf.getFiltered_Host().copy()
    big = torch_cuda.empty((4, 48, 64), dtype=torch_cuda.float32, device="cuda")
    jbf.filter_batch(dev(torch_cuda, depth), dev(torch_cuda, bgr), big)           # n = 4 > max_batch = 1, caller-owned output
    del big
    torch_cuda.cuda.empty_cache()
    again = jbf.getFiltered_Host()
    assert again.shape == (48, 64) and np.array_equal(again, own)                # still the object's own frame 0
    ref = oracle.jbf_kernel(depth[0], bgr[0])
    assert_depth_close(own, ref, RTOL, what="filtered host")


@pytest.mark.parametrize("cfg", [
    dict(w=5, ss=70.0, cs=50.0, ds=20.0),      # the reference's compile-time constants
    dict(w=11, ss=3.0, cs=7.65, ds=20.0),      # BASELINE "radius=5 sigma_s=3 sigma_r=0.03"
    dict(w=19, ss=3.0, cs=7.65, ds=20.0),      # BASELINE "radius=9"
    dict(w=7, ss=30.0, cs=50.0, ds=70.0),
    dict(w=3, ss=1.0, cs=0.0, ds=20.0),        # colour term off
    dict(w=5, ss=70.0, cs=50.0, ds=0.0),       # depth term off (Q3)
    dict(w=1, ss=70.0, cs=50.0, ds=20.0),
    dict(w=31, ss=0.5, cs=30.0, ds=40.0),      # spatial table underflows to 0 at the rim -> factor skipped
])
def test_filter_k1_matches_oracle(torch_cuda, F, oracle, frame, cfg):
    bgr, depth = frame(2, 320, 240)
    jbf = F.JointBilateralFilter(320, 240, params(F, cfg["w"], cfg["ss"], cfg["cs"], cfg["ds"], pre=0))
    out = torch_cuda.empty((1, 240, 320), dtype=torch_cuda.float32, device="cuda")
    jbf.filter_batch(dev(torch_cuda, depth[None]), dev(torch_cuda, bgr[None]), out)
    assert_k1_stagewise(jbf.params, depth, bgr, host(out)[0], what=f"K1 {cfg}", band_max=0.01)
    assert np.array_equal(jbf.spatial_table(), oracle.spatial_table(cfg["w"], cfg["ss"]))


def test_process_on_reference_color_fixture(torch_cuda, F, oracle, color_fixture, synth):
    """BASELINE config 2: the single 640x480 frame (input/color.jpg decode + synthetic depth seed 1,
    depth.xml being absent), JointBilateralFilter::Process with the reference constants."""
    _, depth = synth.make_frame(1, 640, 480)
    jbf = F.JointBilateralFilter(640, 480)
    jbf.Process(dev(torch_cuda, depth), dev(torch_cuda, color_fixture))
    ref, smooth, ill = oracle.jbf_process(depth, color_fixture, return_all=True)
    assert np.array_equal(host(jbf.getSmoothImage_Device()), smooth)
    got = host(jbf.getFiltered_Device())
    assert_k1_stagewise(jbf.params, depth, smooth, got, what="Process (reference constants, colour fixture)", band_max=0.003)
    assert_depth_close(got, ref, RTOL, ill=ill, what="Process vs the float32 restatement (cross-check)", max_flagged=0.01)
    assert np.array_equal(jbf.getFiltered_Host(), got)


@pytest.mark.parametrize("size", [(70, 50), (33, 9), (7, 5), (1, 1), (640, 1), (2, 300)])
def test_ragged_and_tiny_frames(torch_cuda, F, oracle, frame, size):
    w, h = size
    bgr, depth = frame(9, max(w, 8), max(h, 8))
    bgr, depth = np.ascontiguousarray(bgr[:h, :w]), np.ascontiguousarray(depth[:h, :w])
    jbf = F.JointBilateralFilter(w, h)
    jbf.Process(dev(torch_cuda, depth), dev(torch_cuda, bgr))
    smooth = oracle.cv_bilateral(bgr, 5, 30.0, 30.0)
    assert np.array_equal(host(jbf.getSmoothImage_Device()), smooth)
    assert_k1_stagewise(jbf.params, depth, smooth, host(jbf.getFiltered_Device()), what=f"{w}x{h}", band_max=1.0 if w * h < 500 else 0.05)


def test_batch_equals_per_frame_and_is_order_independent(torch_cuda, F, oracle, synth):
    bgr, depth = synth.make_batch(20, 5, 160, 120)
    jbf = F.JointBilateralFilter(160, 120, max_batch=5)
    out = host(jbf.process_batch(dev(torch_cuda, depth), dev(torch_cuda, bgr))).copy()
    smooth = host(jbf.getSmoothImage_Device(5))
    for i in range(5):
        assert np.array_equal(smooth[i], oracle.cv_bilateral(bgr[i], 5, 30.0, 30.0))
        assert_k1_stagewise(jbf.params, depth[i], smooth[i], out[i], what=f"batch frame {i}")
    perm = [3, 0, 4, 1, 2]
    out2 = host(jbf.process_batch(dev(torch_cuda, depth[perm]), dev(torch_cuda, bgr[perm])))
    assert np.array_equal(out2, out[perm])          # frames are independent units: bitwise


def test_edge_cases_invalid_and_q1(torch_cuda, F, oracle):
    h, w = 40, 48
    bgr = np.full((h, w, 3), 77, np.uint8)
    jbf = F.JointBilateralFilter(w, h, params(F, pre=0))
    out = torch_cuda.empty((1, h, w), dtype=torch_cuda.float32, device="cuda")

    def run(depth):
        jbf.filter_batch(dev(torch_cuda, depth[None]), dev(torch_cuda, bgr[None]), out)
        return host(out)[0].copy()

    assert np.all(run(np.zeros((h, w), np.float32)) == 0)                    # all invalid
    assert np.all(run(np.full((h, w), 50.0, np.float32)) == 0)               # 50 is not > 50
    d = np.zeros((h, w), np.float32)
    d[20, 20] = 1234.5                                                       # single valid pixel fills its window
    o = run(d)
    assert np.count_nonzero(o) == 25 and np.allclose(o[18:23, 18:23], 1234.5, rtol=1e-6)
    # Q1: outliers either side of the 288.41 mm underflow threshold
    for delta in (250.0, 280.0, 295.0, 320.0, 2000.0):
        d = np.full((h, w), 1000.0, np.float32)
        d[10, 10] += delta
        assert_depth_close(run(d), oracle.jbf_kernel(d, bgr), RTOL, what=f"outlier {delta}")
    d = np.full((h, w), np.nan, np.float32)                                  # NaN depth is "not > 50": ignored
    d[5:9, 5:9] = 800.0
    assert_depth_close(run(d), oracle.jbf_kernel(d, bgr), RTOL, what="nan depth")


def test_denormal_range_weights_are_kept(torch_cuda, F, oracle):
    """Round 1's divergence: raw v_exp_f32 flushes results below 2^-126, so a pixel whose weights were ALL float32
    denormals came out 0 (a hole) where the reference writes a finite depth.  The tuned kernels now form weights at
    2^24 scale.  Scene: holes whose valid neighbours all differ by (60,60,60) -> colour factor 2^-133 at sigma_c 7.65."""
    h, w = 40, 48
    rng = np.random.default_rng(1)
    depth = (1000.0 + 10.0 * rng.random((h, w))).astype(np.float32)
    bgr = np.full((h, w, 3), 60, np.uint8)
    holes = [(y, x) for y in range(6, h - 6, 7) for x in range(6, w - 6, 7)]
    for y, x in holes:
        depth[y, x] = 0.0
        bgr[y, x] = 0
    for win in (5, 7, 11, 19):
        ref, env = oracle.jbf_kernel(depth, bgr, win, 3.0, 7.65, 20.0, return_ill=True)
        assert all(ref[y, x] > 900.0 for y, x in holes)                 # the reference fills these holes
        for v, nm, vw in [(-1, "auto", win), (0, "generic", win)] + [t for t in _variant_windows(F) if t[2] == win]:
            jbf = F.JointBilateralFilter(w, h, params(F, win, 3.0, 7.65, 20.0, pre=0))
            jbf.set_variant(v)
            out = torch_cuda.empty((1, h, w), dtype=torch_cuda.float32, device="cuda")
            jbf.filter_batch(dev(torch_cuda, depth[None]), dev(torch_cuda, bgr[None]), out)
            got = host(out)[0]
            assert all(got[y, x] > 900.0 for y, x in holes), f"variant {nm}: a denormal-weight pixel was flushed to 0"
            assert_depth_close(got, ref, RTOL, ill=env, what=f"denormal weights, variant {nm} window {win}")
            assert_k1_stagewise(jbf.params, depth, bgr, got, variant=v, what=f"denormal weights, variant {nm} window {win}", band_max=0.1)


def test_weights_of_the_last_denormal_unit_are_kept(torch_cuda, F, oracle):
    """Found by the stress tool (seed 777, case 791).  A colour factor exp(-103.68) is 0.67 units of the float32 denormal
    grid: expf rounds it to one unit, it is not "== 0", so it is multiplied in and the hole is filled.  The library
    expf() on the device returns 0 below 2^-149 (x < -103.28), which lost such weights in the reference-shaped
    (generic) kernel for 103.28 < x < 103.97; it now uses exp_denormal() (csrc/kde_device_math.h).
    Scene: holes of colour (96,96,0) in a (224,224,224) frame, sigma_c 20 -> cd / 800 = 103.68 for every neighbour."""
    h, w = 36, 44
    rng = np.random.default_rng(5)
    depth = (1800.0 + 20.0 * rng.random((h, w))).astype(np.float32)
    bgr = np.full((h, w, 3), 224, np.uint8)
    holes = [(y, x) for y in range(5, h - 5, 6) for x in range(5, w - 5, 6)]
    for y, x in holes:
        depth[y, x] = 0.0
        bgr[y, x] = (96, 96, 0)
    for win in (3, 5, 9, 11):
        ref, env = oracle.jbf_kernel(depth, bgr, win, 70.0, 20.0, 1000.0, return_ill=True)
        assert all(ref[y, x] > 1700.0 for y, x in holes)                # the reference fills these holes
        assert all((env.flags[y, x] & oracle.Env.ZERO_OK) == 0 for y, x in holes)
        for v, nm, vw in [(-1, "auto", win), (0, "generic", win)] + [t for t in _variant_windows(F) if t[2] == win]:
            jbf = F.JointBilateralFilter(w, h, params(F, win, 70.0, 20.0, 1000.0, pre=0))
            jbf.set_variant(v)
            out = torch_cuda.empty((1, h, w), dtype=torch_cuda.float32, device="cuda")
            jbf.filter_batch(dev(torch_cuda, depth[None]), dev(torch_cuda, bgr[None]), out)
            got = host(out)[0]
            assert all(got[y, x] > 1700.0 for y, x in holes), f"variant {nm} window {win}: a one-unit weight was lost"
            assert_depth_close(got, ref, RTOL, ill=env, what=f"last denormal unit, variant {nm} window {win}")
            assert_k1_stagewise(jbf.params, depth, bgr, got, variant=v, what=f"last denormal unit, variant {nm} window {win}", band_max=0.1)


def test_weights_below_one_denormal_unit_fixture(torch_cuda, F, oracle):
    """tests/golden/k1_grid_numerator.npz (stress seed 204, case 192): holes whose surviving weights are below one unit of the
    float32 denormal grid.  The generic kernel (float32 arithmetic like the reference's) returns round(d) there, the tuned
    kernels (weights at 2^24 scale) the unquantised mean; both lie in the GRID interval of the stage-wise check."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "k1_grid_numerator.npz"))
    win, ss, cs, ds = z["params"]
    depth, bgr = z["depth"], z["bgr"]
    h, w = depth.shape
    for v in (-1, 0):
        p = params(F, int(win), float(ss), float(cs), float(ds), pre=0)
        jbf = F.JointBilateralFilter(w, h, p)
        jbf.set_variant(v)
        out = torch_cuda.empty((1, h, w), dtype=torch_cuda.float32, device="cuda")
        jbf.filter_batch(dev(torch_cuda, depth[None]), dev(torch_cuda, bgr[None]), out)
        r = assert_k1_stagewise(p, depth, bgr, host(out)[0], variant=v, what=f"sub-unit weights, variant {v}", band_max=0.2)
        assert r["grid"] >= 2


def test_non_finite_and_huge_depth_samples(torch_cuda, F, oracle, frame):
    """What is guaranteed for depth values no sensor produces (input domain).  A tap of +inf or of
    3e38 mm is "valid" (> 50) in the reference and turns its window into NaN / inf / 1e37.  (a) The reference-shaped
    kernel reproduces that class for class.  (b) The tuned kernels sum weights at 2^24 scale, so their guarantee ends at
    2^64 mm: a 2^64 sample still matches everywhere; beyond it the pixels whose window holds such a sample are
    non-finite garbage in both implementations but not necessarily the same garbage -- every OTHER pixel is held to
    the usual bar.  -inf and NaN are "not > 50" and simply absent, as in the reference."""
    bgr, depth = frame(5, 96, 64)
    h, w = depth.shape
    hostile = depth.copy()
    spots = {(10, 10): np.inf, (30, 50): np.inf, (31, 52): -np.inf, (50, 20): 3.0e38, (12, 70): np.nan}
    for (y, x), v in spots.items():
        hostile[y, x] = v
    big = depth.copy()
    big[40, 40] = 2.0 ** 64

    def classes(a):
        return np.where(np.isnan(a), 2, np.where(np.isinf(a), 3, 0))

    for win in (5, 11, 19):
        cfg = (win, 3.0, 7.65, 20.0) if win > 5 else (5, 70.0, 50.0, 20.0)
        jbf = F.JointBilateralFilter(w, h, params(F, *cfg, pre=0))
        out = torch_cuda.empty((1, h, w), dtype=torch_cuda.float32, device="cuda")

        def run(v, d):
            jbf.set_variant(v)
            jbf.filter_batch(dev(torch_cuda, d[None]), dev(torch_cuda, bgr[None]), out)
            return host(out)[0].copy()

        ref, env = oracle.jbf_kernel(hostile, bgr, *cfg, return_ill=True)
        touched = np.zeros((h, w), bool)                        # pixels whose window holds a +inf / 3e38 sample
        r = win // 2
        for (y, x), v in spots.items():
            if v > 1e30:
                touched[max(0, y - r):y + r + 1, max(0, x - r):x + r + 1] = True
        assert np.isfinite(ref[~touched]).all() and (~np.isfinite(ref[touched])).sum() > 0
        g0 = run(0, hostile)                                    # (a) reference-shaped kernel
        assert np.array_equal(classes(g0), classes(ref))
        fin = np.isfinite(ref)
        assert_depth_close(np.where(fin, g0, 0), np.where(fin, ref, 0), RTOL, ill=env, what=f"hostile depth, generic, window {win}")
        g1 = run(-1, hostile)                                   # (b) tuned kernel: every untouched pixel as usual
        assert_k1_stagewise(jbf.params, hostile, bgr, g1, what=f"hostile depth, tuned, window {win}", ignore=touched)
        assert (~np.isfinite(g1[touched & ~fin])).all()        # where the reference is non-finite, so is the tuned kernel
        refb, envb = oracle.jbf_kernel(big, bgr, *cfg, return_ill=True)
        assert np.isfinite(refb).all()
        assert_k1_stagewise(jbf.params, big, bgr, run(-1, big), what=f"2^64 mm sample, tuned, window {win}")


@pytest.mark.parametrize("win", [9, 11, 13, 15, 19, 21, 25, 31])
def test_rule_elision_bodies_all_match_the_oracle(torch_cuda, F, oracle, win):
    """K1's tuned kernels (windows >= 9) pick, per tile, a body without the colour and / or depth Q1 rule when the
    tile's colour / depth ranges prove the rule cannot trip.  Four quadrants force each of the four bodies: smooth,
    colour edges only (> 63 levels per channel: cd >= cd_skip at sigma_c 7.65), depth steps only (> 288 mm), both."""
    h, w = 192, 256
    rng = np.random.default_rng(win)
    yy, xx = np.mgrid[0:h, 0:w]
    depth = (1500.0 + 0.2 * xx + 0.1 * yy + rng.normal(0, 1.0, (h, w))).astype(np.float32)
    bgr = np.full((h, w, 3), 90, np.int32) + rng.integers(-3, 4, (h, w, 3))
    stripes = ((xx // 9) % 2).astype(bool)
    cq = stripes & (((xx >= w // 2) & (yy < h // 2)) | ((xx >= w // 2) & (yy >= h // 2)))     # right half: colour edges
    bgr[cq] += 120
    dq = (((yy // 7) % 2) == 1) & (yy >= h // 2)                                            # bottom half: depth steps
    depth[dq] += 400.0
    depth[rng.random((h, w)) < 0.01] = 0
    bgr = np.clip(bgr, 0, 255).astype(np.uint8)
    from tools.hooks import stage
    for v, nm, vw in [(-1, "auto", win)] + [t for t in _variant_windows(F) if t[2] == win and "-pk" in t[1]]:
        p = params(F, win, 3.0, 7.65, 20.0, pre=0)
        jbf = F.JointBilateralFilter(w, h, p)
        jbf.set_variant(v)
        out = torch_cuda.empty((1, h, w), dtype=torch_cuda.float32, device="cuda")
        jbf.filter_batch(dev(torch_cuda, depth[None]), dev(torch_cuda, bgr[None]), out)
        got = host(out)[0]
        #  the elided bodies are algebraically identical to the full-rule body where their precondition holds:
        # with elision forced off the output must not change by a bit, and the counters prove which bodies ran
        _, _, bodies = stage.jbf_stage_run(p, depth[None], bgr[None], v)
        forced, _, bodies_forced = stage.jbf_stage_run(p, depth[None], bgr[None], v, force_full_rules=True)
        assert stage.bits_equal(forced[0], got), f"variant {nm}: forcing the full-rule body changed the output"
        assert bodies_forced[3] == bodies[:4].sum() and bodies_forced[:3].sum() == 0
        if "noelide" in nm:
            assert bodies[3] == bodies[:4].sum() and bodies[3] > 0, bodies[:4]                           # the full-rule body everywhere
        elif win >= 15:
            assert bodies[3] > 0 and bodies[1] == 0 and bodies[2] == 0, bodies[:4]                       # {none, both} only
            assert bodies[0] > 0 or v != -1, bodies[:4]        # (the 128-pixel-wide tiles of some variants all straddle an edge here)
        else:
            assert (bodies[:4] > 0).all(), f"variant {nm}: not every rule-specialised body ran: {bodies[:4]}"
        assert_k1_stagewise(p, depth, bgr, got, variant=v, what=f"rule elision, variant {nm} (bodies {bodies[:4].tolist()})")


def test_full_size_properties_1080p(torch_cuda, F):
    """BASELINE-size frames, checked through size-independent properties (no oracle run)."""
    H, W, n = 1080, 1920, 3
    t = torch_cuda
    g = t.Generator(device="cuda").manual_seed(0)
    bgr = t.randint(0, 256, (n, H, W, 3), dtype=t.uint8, device="cuda", generator=g)
    jbf = F.JointBilateralFilter(W, H, params(F, 19, 3.0, 7.65, 20.0), max_batch=n)
    flat = t.full((n, H, W), 1500.0, dtype=t.float32, device="cuda")
    out = jbf.process_batch(flat, bgr)
    assert t.allclose(out, flat, rtol=1e-5, atol=0)                          # constant depth -> identity
    assert t.count_nonzero(jbf.process_batch(t.zeros_like(flat), bgr)) == 0  # all invalid -> zeros
    depth = 500.0 + 3000.0 * t.rand((n, H, W), device="cuda", generator=g)
    depth[:, ::7, ::5] = 0
    a = jbf.process_batch(depth, bgr).clone()
    valid = depth > 50
    lo = t.where(valid, depth, t.full_like(depth, 1e9)).amin()
    hi = depth.amax()
    nz = a != 0
    assert a[nz].min() >= lo * (1 - 1e-5) and a[nz].max() <= hi * (1 + 1e-5)   # convex combination of valid taps
    b = jbf.process_batch(depth.flip(0).contiguous(), bgr.flip(0).contiguous()).flip(0)
    assert t.equal(a, b)                                                      # frames independent, deterministic
    # mirror symmetry (symmetric spatial table): filtering the x-flipped frame == x-flipping the filtered
    # frame up to summation order.  Checked on a smooth surface: on wild data a tap sitting on the Q1
    # underflow threshold (|d - avg| = 288.41 mm) legitimately flips with the last ulp of the average.
    yy, xx = t.meshgrid(t.arange(H, device="cuda"), t.arange(W, device="cuda"), indexing="ij")
    smooth = (1000.0 + 0.3 * xx + 0.2 * yy)[None].repeat(n, 1, 1) + 6.0 * t.rand((n, H, W), device="cuda", generator=g) - 3.0
    smooth[:, ::7, ::5] = 0
    s1 = jbf.process_batch(smooth.contiguous(), bgr).clone()
    s2 = jbf.process_batch(smooth.flip(2).contiguous(), bgr.flip(2).contiguous()).flip(2)
    rel = ((s2 - s1).abs() / s1.abs().clamp_min(1)).max().item()
    assert rel < 1e-4 and t.equal(s2 != 0, s1 != 0)


def test_mrf_sibling_filter(torch_cuda, F, oracle, frame):
    bgr, depth = frame(2, 320, 240)
    mrf = F.MarkovRandomField(320, 240)
    mrf.Process(dev(torch_cuda, depth), dev(torch_cuda, bgr))
    assert_mrf_close(host(mrf.getFiltered_Device()), oracle.mrf_kernel(depth, bgr), "MRF")
    # float* getFiltered_Host() (MarkovRandomField.h:16): the object's own Filtered_Device, never a caller's buffer
    assert np.array_equal(mrf.getFiltered_Host(), host(mrf.getFiltered_Device()))
    own = mrf.getFiltered_Host().copy()
    big = torch_cuda.empty((2, 240, 320), dtype=torch_cuda.float32, device="cuda")
    mrf.process_batch(dev(torch_cuda, np.stack([depth, depth * 0.5])), dev(torch_cuda, np.stack([bgr, bgr])), big)   # caller-owned output
    assert np.array_equal(mrf.getFiltered_Host(), own)


@pytest.mark.parametrize("cfg", [
    dict(size=(320, 240), window=5, cs=0.002, ss=2.0),       # tuned kernel, weights that actually vary
    dict(size=(203, 77), window=5, cs=0.0005, ss=150.0),     # ragged size, odd width (half-empty last pixel pair)
    dict(size=(320, 240), window=5, cs=50.0, ss=0.0),        # smooth_sigma 0: every tap vanishes, output = input
    dict(size=(160, 120), window=7, cs=0.002, ss=2.0),       # other windows: generic kernel
    dict(size=(160, 120), window=5, cs=0.0, ss=2.0),         # color_sigma 0: generic kernel (taps get weight 0)
])
def test_mrf_other_parameters_and_batches(torch_cuda, F, oracle, synth, frame, cfg):
    w, h = cfg["size"]
    bgr, depth = synth.make_batch(40, 3, w, h)
    depth[0][::7, ::5] = 0                                       # holes: the centre still enters with weight 1
    mrf = F.MarkovRandomField(w, h, max_batch=3, window=cfg["window"], color_sigma=cfg["cs"], smooth_sigma=cfg["ss"])
    out = torch_cuda.empty((3, h, w), dtype=torch_cuda.float32, device="cuda")
    mrf.process_batch(dev(torch_cuda, depth), dev(torch_cuda, bgr), out)
    for f in range(3):
        ref = oracle.mrf_kernel(depth[f], bgr[f], cfg["window"], cfg["cs"], cfg["ss"])
        assert_mrf_close(host(out)[f], ref, f"MRF {cfg} frame {f}")
    if cfg["ss"] == 0.0:
        assert np.array_equal(host(out), depth)


def test_errors_are_reported_not_fatal(torch_cuda, F):
    from kinectdepthmapenhancement_amd import KdeError
    jbf = F.JointBilateralFilter(64, 48)
    with pytest.raises(ValueError):
        jbf.Process(torch_cuda.zeros((48, 63), device="cuda"), torch_cuda.zeros((48, 64, 3), dtype=torch_cuda.uint8, device="cuda"))
    with pytest.raises(KdeError):
        jbf.process_batch(torch_cuda.zeros((2, 48, 64), device="cuda"),
                          torch_cuda.zeros((2, 48, 64, 3), dtype=torch_cuda.uint8, device="cuda"))   # n > max_batch
    with pytest.raises(KdeError):
        F.JointBilateralFilter(64, 48, params(F, w=6))


def _variant_windows(F):
    names = F.JointBilateralFilter.variants()
    out = []
    for v, nm in enumerate(names):
        if v == 0:
            continue
        out.append((v, nm, int(nm.split("-")[0][1:])))
    return out


@pytest.mark.parametrize("regime", ["reference-sigmas", "colour-underflow", "depth-outliers"])
def test_every_tuned_variant_matches_oracle(torch_cuda, F, oracle, frame, regime):
    """all LDS-tile / pixels-per-thread variants of K1 (BASELINE config 3's sweep space) against the oracle,
    in the parameter regimes that select different code paths (colour-factor skip on/off)."""
    ss, cs, ds = {"reference-sigmas": (70.0, 50.0, 20.0), "colour-underflow": (3.0, 7.65, 20.0),
                  "depth-outliers": (5.0, 20.0, 4.0)}[regime]
    for size, seed in (((320, 240), 2), ((70, 50), 9)):
        w, h = size
        bgr, depth = frame(seed, w, h)
        for v, nm, win in _variant_windows(F):
            jbf = F.JointBilateralFilter(w, h, params(F, win, ss, cs, ds, pre=0))
            jbf.set_variant(v)
            out = torch_cuda.empty((1, h, w), dtype=torch_cuda.float32, device="cuda")
            jbf.filter_batch(dev(torch_cuda, depth[None]), dev(torch_cuda, bgr[None]), out)
            # sigma_d = 4 mm is a stress regime for the COMPOSITE filter: d(ln weight)/d(avg) = delta/sigma_d^2, so the
            # last ulp of the window average already moves single weights by 1e-3.  Stage by stage there is nothing
            # ill-conditioned: the average is held to its float32 bound and the final value to 1e-4 from that average.
            assert_k1_stagewise(jbf.params, depth, bgr, host(out)[0], variant=v, what=f"variant {nm} {regime} {w}x{h}")


def test_variant_selection_errors(torch_cuda, F):
    from kinectdepthmapenhancement_amd import KdeError
    jbf = F.JointBilateralFilter(64, 48, params(F, 1, pre=0))      # no tuned kernel for window 1: generic path
    assert jbf.active_variant() == "generic-32x8-1px"             # ... and the handle says so (kde_jbf_active_variant)
    d = torch_cuda.full((1, 48, 64), 900.0, device="cuda")
    c = torch_cuda.zeros((1, 48, 64, 3), dtype=torch_cuda.uint8, device="cuda")
    o = torch_cuda.empty_like(d)
    jbf.filter_batch(d, c, o)
    assert torch_cuda.equal(o, d)                                  # (one tap: the pixel itself)
    jbf.set_variant(1)                                             # a window-5 kernel cannot serve window 1
    with pytest.raises(KdeError):
        jbf.filter_batch(d, c, o)
    with pytest.raises(KdeError):
        jbf.set_variant(10 ** 6)


def test_tuned_kernels_serve_every_window(torch_cuda, F, oracle, frame):
    """the ABI takes any odd window <= 31 (the reference's window_size is a run-time argument,
    JointBilateralFilter.cu:10,18-19); every window from 3 to 31 selects a tuned packed kernel (23..31 read their log2(S)
    table from a device copy: it no longer fits the kernel-argument block), window 1, zero sigmas and exotic colour sigmas
    the generic one -- and kde_jbf_active_variant tells which"""
    for win in range(1, 32, 2):
        name = F.JointBilateralFilter(64, 48, params(F, win, 3.0, 7.65, 20.0, pre=0)).active_variant()
        if win >= 3:
            assert name.startswith(f"w{win}-pk"), (win, name)
        else:
            assert name == "generic-32x8-1px", (win, name)
    # the wide windows against the oracle through the default path (Process with pre-smoothing, batch of 2)
    bgr, depth = frame(5, 96, 72)
    for win in (23, 27, 31):
        p = params(F, win, 6.0, 20.0, 30.0)
        jbf = F.JointBilateralFilter(96, 72, p, max_batch=2)
        out = host(jbf.process_batch(dev(torch_cuda, np.stack([depth, depth[::-1].copy()])), dev(torch_cuda, np.stack([bgr, bgr[::-1].copy()]))))
        smooth = host(jbf.getSmoothImage_Device(2))
        assert np.array_equal(smooth[0], oracle.cv_bilateral(bgr, 5, 30.0, 30.0))
        assert_k1_stagewise(p, depth, smooth[0], out[0], what=f"window {win} through the device table")
        assert_k1_stagewise(p, depth[::-1].copy(), smooth[1], out[1], what=f"window {win}, second frame of the batch")
    assert F.JointBilateralFilter(64, 48, params(F, 11, 3.0, 0.0, 20.0, pre=0)).active_variant() == "generic-32x8-1px"      # colour term off
    j = F.JointBilateralFilter(64, 48, params(F, 11, 3.0, 7.65, 20.0, pre=0))
    j.set_variant(0)
    assert j.active_variant() == "generic-32x8-1px"
    j.set_variant(F.JointBilateralFilter.variants().index("w11-sc2-16x16-false"))
    assert j.active_variant() == "w11-sc2-16x16-false"


def test_randomised_parity_sweep(torch_cuda, oracle):
    """tools/stress_parity.py: random sizes / windows / sigmas / hole densities through K1 (every variant), K0, MRF and
    the RegionGrowingBilateralFilter pipeline; 900 cases over three seeds were clean when this was written."""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("stress_parity", os.path.join(ROOT, "tools", "stress_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(cases=80, seed=7) == 0
