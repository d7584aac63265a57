"""BASELINE.json's own workloads at their own sizes, HIP (through the C ABI) against the CPU oracle.

config 2: one 640x480 frame, JointBilateralFilter::Process, radius 5 -> window 11, sigma_s 3, sigma_r 0.03 -> 7.65
config 3: 1920x1080, radius 9 -> window 19 (the pass north_star's roofline target names)
config 4: one rank's shard of the 512-frame batch: 64 x 640x480 through the batched entry point
config 5: DimensionConvertor -> JointBilateralFilter -> RegionGrowingBilateralFilter on one 1080p frame
No pixel is excluded (conftest.assert_depth_close): unflagged pixels 1e-4 against the float32 restatement, flagged
ones inside the oracle's binary64 envelope; u8 images and labels bit-exact."""
import numpy as np
import pytest

from conftest import assert_depth_close
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
    assert_depth_close(host(jbf.getFiltered_Device()), ref, 1e-4, ill=env, what="config 2 (640x480, window 11)",
                       max_flagged=0.05)


def test_config3_fhd_process_window19(torch_cuda, F, oracle, synth):
    bgr, depth = synth.make_frame(3, 1920, 1080)
    jbf = F.JointBilateralFilter(1920, 1080, _params(F, 19, 3.0, 7.65, 20.0))
    jbf.Process(dev(torch_cuda, depth), dev(torch_cuda, bgr))
    ref, smooth, env = oracle.jbf_process(depth, bgr, 19, 3.0, 7.65, 20.0, return_all=True)
    assert np.array_equal(host(jbf.getSmoothImage_Device()), smooth)            # K0 u8 image at 1080p: bit-exact
    assert_depth_close(host(jbf.getFiltered_Device()), ref, 1e-4, ill=env, what="config 3 (1920x1080, window 19)",
                       max_flagged=0.05)


def test_config4_one_ranks_shard_64_vga_frames(torch_cuda, F, oracle, synth):
    """the per-GPU shard of BASELINE config 4 through kde_jbf_process_batch; frames 0, 31 and 63 against the oracle"""
    n = 64
    bgr, depth = synth.make_batch(0, n, 640, 480)
    jbf = F.JointBilateralFilter(640, 480, _params(F, **BENCH), max_batch=n)
    out = host(jbf.process_batch(dev(torch_cuda, depth), dev(torch_cuda, bgr)))
    smooth = host(jbf.getSmoothImage_Device(n))
    for f in (0, 31, 63):
        ref, sm, env = oracle.jbf_process(depth[f], bgr[f], BENCH["window"], BENCH["ss"], BENCH["cs"], BENCH["ds"],
                                          return_all=True)
        assert np.array_equal(smooth[f], sm)
        assert_depth_close(out[f], ref, 1e-4, ill=env, what=f"config 4 shard frame {f}", max_flagged=0.05)
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
    ref_filt, smooth, env = oracle.jbf_process(depth, bgr, return_all=True)
    assert np.array_equal(host(jbf.getSmoothImage_Device()), smooth)
    assert_depth_close(got_filt, ref_filt, 1e-4, ill=env, what="config 5 JBF (1080p)", max_flagged=0.05)
    opts = oracle.p2r_depth(got_filt, K)
    assert np.array_equal(host(pts), pts_as_f32(opts))
    with oracle.ers_flags((h, w)) as env2:
        ref = oracle.rgbf_process(got_filt, opts, bgr, 15, 20, K)
    assert np.array_equal(host(rg.getSPLabels_Device()), ref["sp_labels"])
    assert np.array_equal(host(rg.getDASPLabels_Device()), ref["dasp_labels"])
    assert np.array_equal(host(rg.getRefinedLabels_Device()), ref["refined_labels"])
    assert_depth_close(host(rg.getRefinedDepth_Device()), ref["refined_depth"], 1e-4, ill=env2,
                       what="config 5 RGBF (1080p)", max_flagged=0.05)
