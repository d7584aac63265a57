"""BASELINE.json's own workloads at their own sizes, HIP (through the C ABI) against the CPU oracle.

config 2: one 640x480 frame, JointBilateralFilter::Process, radius 5 -> window 11, sigma_s 3, sigma_r 0.03 -> 7.65
config 3: 1920x1080, radius 9 -> window 19 (the pass north_star's roofline target names)
config 4: one rank's shard of the 512-frame batch: 64 x 640x480 through the batched entry point
config 5: DimensionConvertor -> JointBilateralFilter -> RegionGrowingBilateralFilter on one 1080p frame
No pixel is excluded.  K1 / K10 depth is checked stage by stage (conftest.assert_k1_stagewise / assert_k10_stagewise): the
GPU's own first-pass average against binary64 within its float32 bound, the final value at 1e-4 against the last pass
evaluated in binary64 from that average; pixels with a tap on a Q1 decision at that average (BAND) must be <= 0.3 % of
the frame and are held to the interval of both outcomes.  u8 images and labels bit-exact."""
import numpy as np
import pytest

from conftest import assert_depth_close, assert_k1_stagewise, assert_k10_stagewise
from gpu_util import dev, host, pts_as_f32

pytestmark = pytest.mark.gpu

BENCH = dict(window=11, ss=3.0, cs=7.65, ds=20.0)        # bench.py's headline parameters (SURVEY 8d mapping)


@pytest.fixture(scope="module")
def F(torch_cuda):
    from kinectdepthmapenhancement_amd import filters
    return filters


def _params(F, window, ss, cs, ds):
    p = F.JointBilateralFilter.default_params()
    p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma = window, ss, cs, ds
    return p


def test_config2_vga_process_window11(torch_cuda, F, oracle, color_fixture, synth):
    """the reference's colour frame (input/color.jpg decode) + synthetic depth seed 1 (depth.xml is absent)"""
    _, depth = synth.make_frame(1, 640, 480)
    jbf = F.JointBilateralFilter(640, 480, _params(F, **BENCH))
    jbf.Process(dev(torch_cuda, depth), dev(torch_cuda, color_fixture))
    ref, smooth, env = oracle.jbf_process(depth, color_fixture, BENCH["window"], BENCH["ss"], BENCH["cs"], BENCH["ds"],
                                          return_all=True)
    assert np.array_equal(host(jbf.getSmoothImage_Device()), smooth)
    got = host(jbf.getFiltered_Device())
    assert_k1_stagewise(jbf.params, depth, smooth, got, what="config 2 (640x480, window 11)", band_max=0.001, decision_max=1e-4)
    assert_depth_close(got, ref, 1e-4, ill=env, what="config 2 vs the float32 restatement (cross-check)", max_flagged=0.05)


def test_config3_fhd_process_window19(torch_cuda, F, oracle, synth):
    bgr, depth = synth.make_frame(3, 1920, 1080)
    jbf = F.JointBilateralFilter(1920, 1080, _params(F, 19, 3.0, 7.65, 20.0))
    jbf.Process(dev(torch_cuda, depth), dev(torch_cuda, bgr))
    smooth = oracle.cv_bilateral(bgr, 5, 30.0, 30.0)
    assert np.array_equal(host(jbf.getSmoothImage_Device()), smooth)            # K0 u8 image at 1080p: bit-exact
    got = host(jbf.getFiltered_Device())
    assert_k1_stagewise(jbf.params, depth, smooth, got, what="config 3 (1920x1080, window 19)", band_max=0.003, decision_max=1e-4)
    ref, env = oracle.jbf_kernel(depth, smooth, 19, 3.0, 7.65, 20.0, return_ill=True)
    assert_depth_close(got, ref, 1e-4, ill=env, what="config 3 vs the float32 restatement (cross-check)", max_flagged=0.08)


def test_config4_one_ranks_shard_64_vga_frames(torch_cuda, F, oracle, synth):
    """the per-GPU shard of BASELINE config 4 through kde_jbf_process_batch; frames 0, 31 and 63 against the oracle"""
    n = 64
    bgr, depth = synth.make_batch(0, n, 640, 480)
    jbf = F.JointBilateralFilter(640, 480, _params(F, **BENCH), max_batch=n)
    out = host(jbf.process_batch(dev(torch_cuda, depth), dev(torch_cuda, bgr)))
    smooth = host(jbf.getSmoothImage_Device(n))
    for f in (0, 31, 63):
        assert np.array_equal(smooth[f], oracle.cv_bilateral(bgr[f], 5, 30.0, 30.0))
        assert_k1_stagewise(jbf.params, depth[f], smooth[f], out[f], what=f"config 4 shard frame {f}", band_max=0.003, decision_max=1e-4)
    # frames are independent units: a frame filtered alone is bit-identical to the same frame inside the batch
    single = F.JointBilateralFilter(640, 480, _params(F, **BENCH))
    single.Process(dev(torch_cuda, depth[17]), dev(torch_cuda, bgr[17]))
    assert np.array_equal(host(single.getFiltered_Device()), out[17])


def test_config5_fhd_chain_against_the_oracle(torch_cuda, F, oracle, synth):
    """projectiveToReal -> JBF.Process -> projectiveToReal -> RGBF.Process on 1920x1080 (rows 15, cols 20).  Each stage is
    compared on the GPU's own upstream output (labels are discontinuous in their input), as in the 640x480 test."""
    w, h = 1920, 1080
    t = torch_cuda
    bgr, depth = synth.make_frame(21, w, h)
    K = synth.intrinsics(w, h)
    conv = F.DimensionConvertor(); conv.setCameraParameters(K, w, h)
    jbf = F.JointBilateralFilter(w, h)
    rg = F.RegionGrowingBilateralFilter(w, h); rg.SetParametor(15, 20, K)
    color, d = dev(t, bgr), dev(t, depth)
    pts0 = t.empty((h, w, 3), dtype=t.float32, device="cuda")
    conv.projectiveToReal(d, pts0)
    assert np.array_equal(host(pts0), pts_as_f32(oracle.p2r_depth(depth, K)))   # K2 at 1080p: bit-exact
    jbf.Process(d, color)
    filt = jbf.getFiltered_Device()
    pts = t.empty((h, w, 3), dtype=t.float32, device="cuda")
    conv.projectiveToReal(filt, pts)
    rg.Process(filt, pts, color)
    got_filt = host(filt)
    smooth = oracle.cv_bilateral(bgr, 5, 30.0, 30.0)
    assert np.array_equal(host(jbf.getSmoothImage_Device()), smooth)
    assert_k1_stagewise(jbf.params, depth, smooth, got_filt, what="config 5 JBF (1080p)", band_max=0.001, decision_max=1e-4)
    opts = oracle.p2r_depth(got_filt, K)
    assert np.array_equal(host(pts), pts_as_f32(opts))
    ref = oracle.rgbf_process(got_filt, opts, bgr, 15, 20, K)
    assert np.array_equal(host(rg.getSPLabels_Device()), ref["sp_labels"])
    assert np.array_equal(host(rg.getDASPLabels_Device()), ref["dasp_labels"])
    assert np.array_equal(host(rg.getRefinedLabels_Device()), ref["refined_labels"])
    assert_k10_stagewise(ref["sp_labels"], ref["dasp_labels"], got_filt, bgr, host(rg.getRefinedDepth_Device()),
                         what="config 5 RGBF (1080p)", band_max=0.001, decision_max=1e-4)
