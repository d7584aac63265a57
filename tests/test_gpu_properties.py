"""Size-independent properties at BASELINE.json's full sizes — whole batches, every frame and every pixel, where the oracle
comparison of tests/test_gpu_fullsize.py looks at a few frames.  (The property checks themselves use torch on the GPU; the
thing checked is the HIP library's output through the C ABI.)

K1 (`JointBilateralFilter.cu:16-78`) and K10 (`EdgeRefinedSuperpixel.cu:104-205`) return, per pixel, a weighted mean with
non-negative weights of the VALID (> 50 mm) depths of its window, or 0 when no tap qualifies (or every weight underflowed),
or NaN through K10's 0/0 quirk.  Hence:
  * convexity   — a non-zero output lies between the smallest and largest valid depth of the window;
  * no-data     — a window without a valid tap gives exactly 0;
  * constancy   — a constant valid depth map comes back unchanged (to float32 rounding), whatever the colours are;
  * frames are independent — permuting the frames of a batch permutes the outputs, bit for bit.
"""
import numpy as np
import pytest

from gpu_util import dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F(torch_cuda):
    from kinectdepthmapenhancement_amd import filters
    return filters


def _params(F, window, ss, cs, ds):
    p = F.JointBilateralFilter.default_params()
    p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma = window, ss, cs, ds
    return p


def _mean_tol(window):
    """rounding allowance of a float32 weighted mean over window^2 taps (a quarter of the first-order worst case
    window^2 * 2^-24, which no input here approaches), at least 2e-6"""
    return max(2e-6, 0.25 * window * window * 2.0 ** -24)


def _window_range(t, depth, radius):
    """per pixel: (min, max, any) over the valid (> 50) depths of the (2 radius + 1)^2 window, zero padding = invalid"""
    import torch.nn.functional as nnf
    valid = depth > 50.0
    k = 2 * radius + 1
    big = t.where(valid, depth, t.full_like(depth, float("inf")))
    small = t.where(valid, depth, t.full_like(depth, float("-inf")))
    lo = -nnf.max_pool2d(-big[:, None], k, 1, radius)[:, 0]           # -inf padding of max_pool2d = +inf for the min
    hi = nnf.max_pool2d(small[:, None], k, 1, radius)[:, 0]
    return lo, hi, hi > 0.0


def _assert_convex(t, out, depth, radius, what, allow_nan=False):
    lo, hi, has = _window_range(t, depth, radius)
    nan = out != out
    if not allow_nan:
        assert not bool(nan.any()), f"{what}: NaN in the output"
    assert bool((out[~has] == 0.0).all()), f"{what}: a window without valid depth must give 0"
    nz = has & (out != 0.0) & ~nan
    eps = _mean_tol(2 * radius + 1)
    bad = nz & ((out < lo * (1 - eps)) | (out > hi * (1 + eps)))
    assert int(bad.sum()) == 0, f"{what}: {int(bad.sum())} outputs outside the range of their window's valid depths"
    return float(nz.float().mean())


@pytest.mark.parametrize("w,h,n,window", [(640, 480, 64, 11), (1920, 1080, 4, 19), (640, 480, 64, 5)])
def test_jbf_output_is_a_convex_combination_of_its_window(torch_cuda, F, synth, w, h, n, window):
    """config 4's per-GPU shard (64 x 640x480, window 11), config 3 (1080p, window 19), the reference's window 5"""
    t = torch_cuda
    bgr, depth = synth.make_batch(100, min(n, 8), w, h)
    reps = -(-n // bgr.shape[0])
    depth = np.tile(depth, (reps, 1, 1))[:n].copy()
    bgr = np.tile(bgr, (reps, 1, 1, 1))[:n].copy()
    for f in range(n):                                  # make the frames distinct and punch holes of several sizes
        depth[f] = np.roll(depth[f], 7 * f, axis=1)
        depth[f, 40 + f:40 + f + 3 * (f % 9), 100:100 + 5 * (f % 13)] = 0.0
    depth[0, :60, :80] = 0.0                            # a hole larger than every window: exact zeros inside
    jbf = F.JointBilateralFilter(w, h, _params(F, window, 3.0, 7.65, 20.0), max_batch=n)
    d = dev(t, depth)
    out = jbf.process_batch(d, dev(t, bgr))
    frac = _assert_convex(t, out, d, window // 2, f"K1 {n} x {w}x{h} window {window}")
    assert frac > 0.9
    assert bool((out[0, :60 - window // 2 - 1, :80 - window // 2 - 1] == 0.0).all())


@pytest.mark.parametrize("w,h,n,window,cs,ds", [(640, 480, 64, 11, 7.65, 20.0), (1920, 1080, 2, 19, 7.65, 20.0), (640, 480, 8, 5, 50.0, 20.0)])
def test_jbf_constant_depth_comes_back_unchanged(torch_cuda, F, synth, w, h, n, window, cs, ds):
    t = torch_cuda
    bgr, _ = synth.make_batch(300, min(n, 4), w, h)
    bgr = np.tile(bgr, (-(-n // bgr.shape[0]), 1, 1, 1))[:n]
    rng = np.random.default_rng(5)
    bgr = (bgr.astype(np.int32) + rng.integers(-20, 21, bgr.shape)).clip(0, 255).astype(np.uint8)     # textured guide
    for d0 in (800.0, 4321.125):
        depth = t.full((n, h, w), d0, dtype=t.float32, device="cuda")
        jbf = F.JointBilateralFilter(w, h, _params(F, window, 3.0, cs, ds), max_batch=n)
        out = jbf.process_batch(depth, dev(t, bgr))
        err = float(((out - d0).abs() / d0).max())
        assert err <= _mean_tol(window), (d0, err)


def test_jbf_frames_of_a_batch_are_independent(torch_cuda, F, synth):
    """the unit north_star shards over GPUs: reversing the 64 frames of a shard reverses the 64 outputs, bit for bit"""
    t = torch_cuda
    bgr, depth = synth.make_batch(400, 64, 640, 480)
    jbf = F.JointBilateralFilter(640, 480, _params(F, 11, 3.0, 7.65, 20.0), max_batch=64)
    d, c = dev(t, depth), dev(t, bgr)
    fwd = jbf.process_batch(d, c).clone()
    rev = jbf.process_batch(d.flip(0).contiguous(), c.flip(0).contiguous())
    assert t.equal(rev.flip(0), fwd)


@pytest.mark.parametrize("w,h,n", [(640, 480, 16), (1920, 1080, 2)])
def test_rgbf_refined_depth_stays_inside_its_neighbourhood(torch_cuda, F, synth, w, h, n):
    """config 5's chain on a batch: K9 copies depths from at most 5 px away and K10 averages a 7x7 window of those, so a
    refined depth is 0, NaN (Q6) or inside the range of the valid filtered depths within 8 px"""
    t = torch_cuda
    bgr, depth = synth.make_batch(500, n, w, h)
    K = synth.intrinsics(w, h)
    conv = F.DimensionConvertor(); conv.setCameraParameters(K, w, h)
    jbf = F.JointBilateralFilter(w, h, max_batch=n)
    rg = F.RegionGrowingBilateralFilter(w, h, max_batch=n); rg.SetParametor(15, 20, K)
    color, d = dev(t, bgr), dev(t, depth)
    filt = t.empty((n, h, w), dtype=t.float32, device="cuda")
    pts = t.empty((n, h, w, 3), dtype=t.float32, device="cuda")
    jbf.process_batch(d, color, filt)
    conv.projectiveToReal(filt, pts)
    rg.process_batch(filt, pts, color)
    out = rg.getRefinedDepth_Device().reshape(n, h, w)
    lo, hi, has = _window_range(t, filt, 8)
    nan = out != out
    nz = has & (out != 0.0) & ~nan
    bad = nz & ((out < lo * (1 - 2e-6)) | (out > hi * (1 + 2e-6)))
    assert int(bad.sum()) == 0
    assert float(nz.float().mean()) > 0.9 and float(nan.float().mean()) < 1e-3
    labels = rg.getRefinedLabels_Device().reshape(n, h, w)
    assert int(labels.min()) >= -1 and int(labels.max()) < 15 * 20
