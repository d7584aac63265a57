"""BASELINE.json's own workloads at their own sizes, HIP (through the C ABI) against the CPU oracle.

config 2: one 640x480 frame, JointBilateralFilter::Process, radius 5 -> window 11, sigma_s 3, sigma_r 0.03 -> 7.65
config 3: 1920x1080, radius 9 -> window 19 (the pass north_star's roofline target names)
config 4: one rank's shard of the 512-frame batch: 64 x 640x480 through the batched entry point
config 5: DimensionConvertor -> JointBilateralFilter -> RegionGrowingBilateralFilter on one 1080p frame
No pixel is excluded.  K1 / K10 depth is checked stage by stage (conftest.assert_k1_stagewise / assert_k10_stagewise): the
GPU's own first-pass average against binary64 within its float32 bound, the final value at 1e-4 against the last pass
evaluated in binary64 from that average; pixels with a tap on a Q1 decision at that average (BAND) must be <= 0.3 % of
the frame and are held to the interval of both outcomes.  u8 images and labels bit-exact.

Next to the stage-wise bar every config is ALSO counted end to end against the float32 restatement (oracle.deviation_census:
pixels more than 1e-4 away, pixels whose zero mask differs, the same for the denormal-grid class) under a ceiling of about
1.5 x the values measured in round 5 (CENSUS below); and the reference-shaped kernels (kde_jbf_set_variant(h, 0),
kde_ers_set_variant(h, 3)) are shown to have NO zero-mask difference at all against it."""
import numpy as np
import pytest

from conftest import assert_depth_close, assert_k1_stagewise, assert_k10_stagewise
from gpu_util import dev, host, pts_as_f32

pytestmark = pytest.mark.gpu

BENCH = dict(window=11, ss=3.0, cs=7.65, ds=20.0)        # bench.py's headline parameters (SURVEY 8d mapping)

# ceilings of the end-to-end census against the float32 restatement: (fraction of pixels > 1e-4 away, fraction whose zero mask
# differs, fraction flagged by the envelope) = about 1.5 x what round 5 measured (profiles/r05_census.txt)
# measured:        > 1e-4 away            zero mask differs   flagged
#   config 2        3 px  (9.8e-06)        0                   2.75e-02
#   config 3        55 px (2.7e-05)        0                   1.74e-02      (47 of the 55 are GRID pixels)
#   config 4        1-2 px per frame       0                   1.5-2.2e-02
#   config 5 JBF    0                      0                   2.06e-03
#   config 5 RGBF   1 px  (4.8e-07)        0                   4.96e-04
CENSUS = {"config2": (2e-5, 0.0, 0.04), "config3": (4e-5, 0.0, 0.026), "config4": (1.3e-5, 0.0, 0.033),
          "config5_jbf": (2e-6, 0.0, 0.0031), "config5_rgbf": (2e-6, 0.0, 7.5e-4)}


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
    r = assert_k1_stagewise(jbf.params, depth, smooth, got, what="config 2 (640x480, window 11)", band_max=0.001, decision_max=1e-4)
    far, zd, fl = CENSUS["config2"]
    assert_depth_close(got, ref, 1e-4, ill=env, what="config 2 vs the float32 restatement (cross-check)", max_flagged=fl, max_far=far,
                       max_zero_diff=zd, grid=r["grid_map"])
    # the reference-shaped kernel (one pixel per thread, the reference's loop and quantisation): no zero-mask difference
    jbf.set_variant(0)
    jbf.Process(dev(torch_cuda, depth), dev(torch_cuda, color_fixture))
    assert_depth_close(host(jbf.getFiltered_Device()), ref, 1e-4, ill=env, what="config 2, kde_jbf_set_variant(h, 0), vs the float32 restatement",
                       max_zero_diff=0.0)


def test_config3_fhd_process_window19(torch_cuda, F, oracle, synth):
    bgr, depth = synth.make_frame(3, 1920, 1080)
    jbf = F.JointBilateralFilter(1920, 1080, _params(F, 19, 3.0, 7.65, 20.0))
    jbf.Process(dev(torch_cuda, depth), dev(torch_cuda, bgr))
    smooth = oracle.cv_bilateral(bgr, 5, 30.0, 30.0)
    assert np.array_equal(host(jbf.getSmoothImage_Device()), smooth)            # K0 u8 image at 1080p: bit-exact
    got = host(jbf.getFiltered_Device())
    r = assert_k1_stagewise(jbf.params, depth, smooth, got, what="config 3 (1920x1080, window 19)", band_max=0.003, decision_max=1e-4)
    ref, env = oracle.jbf_kernel(depth, smooth, 19, 3.0, 7.65, 20.0, return_ill=True)
    far, zd, fl = CENSUS["config3"]
    assert_depth_close(got, ref, 1e-4, ill=env, what="config 3 vs the float32 restatement (cross-check)", max_flagged=fl, max_far=far,
                       max_zero_diff=zd, grid=r["grid_map"])
    jbf.set_variant(0)
    jbf.Process(dev(torch_cuda, depth), dev(torch_cuda, bgr))
    assert_depth_close(host(jbf.getFiltered_Device()), ref, 1e-4, ill=env, what="config 3, kde_jbf_set_variant(h, 0), vs the float32 restatement",
                       max_zero_diff=0.0)


def test_config4_one_ranks_shard_64_vga_frames(torch_cuda, F, oracle, synth):
    """the per-GPU shard of BASELINE config 4 through kde_jbf_process_batch; frames 0, 31 and 63 against the oracle"""
    n = 64
    bgr, depth = synth.make_batch(0, n, 640, 480)
    jbf = F.JointBilateralFilter(640, 480, _params(F, **BENCH), max_batch=n)
    out = host(jbf.process_batch(dev(torch_cuda, depth), dev(torch_cuda, bgr)))
    smooth = host(jbf.getSmoothImage_Device(n))
    for f in (0, 31, 63):
        assert np.array_equal(smooth[f], oracle.cv_bilateral(bgr[f], 5, 30.0, 30.0))
        r = assert_k1_stagewise(jbf.params, depth[f], smooth[f], out[f], what=f"config 4 shard frame {f}", band_max=0.003, decision_max=1e-4)
        ref, env = oracle.jbf_kernel(depth[f], smooth[f], BENCH["window"], BENCH["ss"], BENCH["cs"], BENCH["ds"], return_ill=True)
        far, zd, fl = CENSUS["config4"]
        assert_depth_close(out[f], ref, 1e-4, ill=env, what=f"config 4 shard frame {f} vs the float32 restatement (cross-check)", max_flagged=fl,
                           max_far=far, max_zero_diff=zd, grid=r["grid_map"])
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
    r = assert_k1_stagewise(jbf.params, depth, smooth, got_filt, what="config 5 JBF (1080p)", band_max=0.001, decision_max=1e-4)
    ref1, env1 = oracle.jbf_kernel(depth, smooth, 5, 70.0, 50.0, 20.0, return_ill=True)
    far, zd, fl = CENSUS["config5_jbf"]
    assert_depth_close(got_filt, ref1, 1e-4, ill=env1, what="config 5 JBF vs the float32 restatement (cross-check)", max_flagged=fl, max_far=far,
                       max_zero_diff=zd, grid=r["grid_map"])
    ref_jbf = F.JointBilateralFilter(w, h)
    ref_jbf.set_variant(0)
    ref_jbf.Process(d, color)
    assert_depth_close(host(ref_jbf.getFiltered_Device()), ref1, 1e-4, ill=env1, what="config 5 JBF, kde_jbf_set_variant(h, 0), vs the float32 restatement",
                       max_zero_diff=0.0)
    opts = oracle.p2r_depth(got_filt, K)
    assert np.array_equal(host(pts), pts_as_f32(opts))
    with oracle.ers_flags((h, w)) as env10:
        ref = oracle.rgbf_process(got_filt, opts, bgr, 15, 20, K)
    assert np.array_equal(host(rg.getSPLabels_Device()), ref["sp_labels"])
    assert np.array_equal(host(rg.getDASPLabels_Device()), ref["dasp_labels"])
    assert np.array_equal(host(rg.getRefinedLabels_Device()), ref["refined_labels"])
    got10 = host(rg.getRefinedDepth_Device())
    r10 = assert_k10_stagewise(ref["sp_labels"], ref["dasp_labels"], got_filt, bgr, got10, what="config 5 RGBF (1080p)", band_max=0.001,
                               decision_max=1e-4)
    far, zd, fl = CENSUS["config5_rgbf"]
    assert_depth_close(got10, ref["refined_depth"], 1e-4, ill=env10, what="config 5 RGBF vs the float32 restatement (cross-check)", max_flagged=fl,
                       max_far=far, max_zero_diff=zd, grid=r10["grid_map"])
    # EdgeRefinedSuperpixel's reference-shaped kernels on the same inputs: no zero-mask difference against the restatement
    ers = F.EdgeRefinedSuperpixel(w, h)
    ers.set_variant(3)
    ers.EdgeRefining(dev(t, ref["sp_labels"]), dev(t, ref["dasp_labels"]), filt, color)
    assert np.array_equal(host(ers.getRefinedLabels_Device()), ref["refined_labels"])
    assert_depth_close(host(ers.getRefinedDepth_Device()), ref["refined_depth"], 1e-4, ill=env10,
                       what="config 5 RGBF, kde_ers_set_variant(h, 3), vs the float32 restatement", max_zero_diff=0.0)
