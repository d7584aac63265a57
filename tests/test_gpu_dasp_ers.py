"""DASP (K5-K8), ERS (K9, K10) and the RGBF / SPDSR pipelines: HIP vs CPU oracle.
Integer outputs (labels, cluster records) must be exact; float cluster centres use the same summation
order as the oracle and are compared exactly; K10 depth is checked stage by stage (conftest.assert_k10_stagewise: the
GPU's own label-restricted average and mean absolute deviation against binary64, then the final value at 1e-4 against the
last pass evaluated in binary64 from them; only pixels with a tap on a Q1 decision at that average keep an interval)."""
import numpy as np
import pytest

from conftest import assert_depth_close, assert_k1_stagewise, assert_k10_stagewise
from gpu_util import dev, host, ld_records, mean_records, pts_as_f32

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F(torch_cuda):
    from kinectdepthmapenhancement_amd import filters
    return filters


def _inputs(oracle, synth, frame, seed, w, h):
    bgr, depth = frame(seed, w, h)
    K = synth.intrinsics(w, h)
    pts = oracle.p2r_depth(depth, K)
    return bgr, depth, K, pts


@pytest.mark.parametrize("cfg", [
    dict(size=(640, 480), rows=15, cols=20, sig=(200.0, 40.0, 0.0), it=1),     # RGBF's SP call
    dict(size=(640, 480), rows=15, cols=20, sig=(100.0, 20.0, 200.0), it=1),   # RGBF's DASP call
    dict(size=(640, 480), rows=15, cols=20, sig=(0.0, 10.0, 200.0), it=5),     # SPDSR's DASP call
    dict(size=(320, 240), rows=7, cols=9, sig=(200.0, 10.0, 0.0), it=3),       # non-dividing grid (ragged windows)
    dict(size=(70, 50), rows=3, cols=5, sig=(100.0, 20.0, 200.0), it=2),
])
def test_dasp_segmentation_exact(torch_cuda, F, oracle, synth, frame, cfg):
    w, h = cfg["size"]
    bgr, depth, K, pts = _inputs(oracle, synth, frame, 11, w, h)
    d = F.DepthAdaptiveSuperpixel(w, h)
    d.SetParametor(cfg["rows"], cfg["cols"], K)
    d.Segmentation(dev(torch_cuda, bgr), dev(torch_cuda, pts_as_f32(pts)), *cfg["sig"], cfg["it"])
    labels, ld, mean, centers = oracle.dasp_segmentation(bgr, pts, cfg["rows"], cfg["cols"], K, *cfg["sig"], cfg["it"])
    assert np.array_equal(host(d.getLabelDevice()), labels)
    gl = ld_records(d.getLDDevice())
    assert np.array_equal(gl["l"], ld["l"]) and np.array_equal(gl["d"], ld["d"])
    gm = mean_records(d.getMeanDataDevice())
    for f in ("r", "g", "b", "x", "y", "size"):
        assert np.array_equal(gm[f], mean[f]), f
    assert np.array_equal(host(d.getCentersDevice()), pts_as_f32(centers), equal_nan=True)


def test_dasp_cluster_centre_kept_below_the_image(torch_cuda, F, oracle, synth, frame):
    """analyzeClusters keeps a projected centre whose row lies below the image (pixel.y <= height [sic], .cu:549): mean.y can
    be any int above the height, INT_MAX included (saturating float -> int).  calculateLD then forms (float)(y - mean.y),
    which a float subtraction reproduces only below 2^24: such a table takes the integer path."""
    w, h = 160, 120
    bgr, depth, K, pts = _inputs(oracle, synth, frame, 15, w, h)
    p = pts_as_f32(pts).copy()
    p[20:50, 30:70] = (0.0, -1.0e9, 100.0)          # projects to column cx, row cy + 1e7 * fy (centroids mix with ordinary points: rows up to 1.4e9)
    p[80:100, 100:140] = (0.0, -40000.0, 100.0)     # row cy + 400 * fy = 57 000: above the height, below 2^24
    pv = np.ascontiguousarray(p).view(oracle.FLOAT3).reshape(h, w)
    for sig in ((200.0, 40.0, 0.0), (100.0, 20.0, 200.0)):
        labels, ld, mean, centers = oracle.dasp_segmentation(bgr, pv, 6, 8, K, *sig, 4)
        assert (mean["y"] >= (1 << 24)).any() and ((mean["y"] > h) & (mean["y"] < (1 << 24))).any()
        d = F.DepthAdaptiveSuperpixel(w, h)
        d.SetParametor(6, 8, K)
        d.Segmentation(dev(torch_cuda, bgr), dev(torch_cuda, p), *sig, 4)
        assert np.array_equal(host(d.getLabelDevice()), labels)
        gl = ld_records(d.getLDDevice())
        assert np.array_equal(gl["l"], ld["l"]) and np.array_equal(gl["d"], ld["d"])
        gm = mean_records(d.getMeanDataDevice())
        for f in ("r", "g", "b", "x", "y", "size"):
            assert np.array_equal(gm[f], mean[f]), f


def test_dasp_geometry_guard(torch_cuda, F, synth):
    from kinectdepthmapenhancement_amd import KdeError
    d = F.DepthAdaptiveSuperpixel(64, 48)
    with pytest.raises(KdeError):
        d.SetParametor(15, 20, synth.intrinsics(64, 48))       # 3x3 windows: 4x4 candidate grid leaves the image
    with pytest.raises(KdeError):
        d.Segmentation(torch_cuda.zeros((48, 64, 3), dtype=torch_cuda.uint8, device="cuda"),
                       torch_cuda.zeros((48, 64, 3), device="cuda"), 1.0, 1.0, 1.0, 1)   # SetParametor not called


@pytest.mark.parametrize("size", [(640, 480), (70, 50)])
def test_ers_edge_refining_and_enhancement(torch_cuda, F, oracle, synth, frame, size):
    w, h = size
    rows, cols = (15, 20) if w == 640 else (3, 5)
    bgr, depth, K, pts = _inputs(oracle, synth, frame, 12, w, h)
    sp = oracle.dasp_segmentation(bgr, pts, rows, cols, K, 200.0, 40.0, 0.0, 1)[0]
    da = oracle.dasp_segmentation(bgr, pts, rows, cols, K, 100.0, 20.0, 200.0, 1)[0]
    ers = F.EdgeRefinedSuperpixel(w, h)
    ers.EdgeRefining(dev(torch_cuda, sp), dev(torch_cuda, da), dev(torch_cuda, depth), dev(torch_cuda, bgr))
    rl, rd9 = oracle.ers_edge_refining(sp, da, depth)
    assert np.array_equal(host(ers.getRefinedLabels_Device()), rl)            # K9 labels exact
    assert np.array_equal(host(ers.getEdgeStageDepth_Device()), rd9)          # K9 depth exact (only zeroing)
    assert (rl != da).sum() > 0 and (rd9 != depth).sum() > 0                  # the case actually exercises K9
    got = host(ers.getRefinedDepth_Device())
    assert_k10_stagewise(sp, da, depth, bgr, got, what=f"K10 {w}x{h}", band_max=0.003 if w == 640 else 0.05)
    with oracle.ers_flags((h, w)) as ill:      # cross-check against the float32 restatement and its own envelope
        ref = oracle.ers_enhance(rd9, bgr, rl)
    assert_depth_close(got, ref, 1e-4, ill=ill, what="K10 vs the float32 restatement (cross-check)", max_flagged=2e-2)
    assert np.array_equal(ers.getRefinedLabels_Host(), rl)
    assert np.array_equal(ers.getRefinedDepth_Host(), got, equal_nan=True)


def test_ers_crafted_label_boundaries(torch_cuda, F, oracle):
    """left / right branch, cascade, overlapping write sets in both scan directions."""
    rng = np.random.default_rng(5)
    H, W = 64, 96
    cl = (np.arange(W)[None, :] // 7 + 13 * (np.arange(H)[:, None] // 5)).astype(np.int32)
    dl = ((np.arange(W)[None, :] + 2) // 7 + 13 * ((np.arange(H)[:, None] + 3) // 5)).astype(np.int32)
    dl[rng.random((H, W)) < 0.02] = 999                      # speckle: many overlapping sources
    depth = (800 + 40 * dl + rng.normal(0, 30, (H, W))).astype(np.float32)
    depth[rng.random((H, W)) < 0.05] = 0
    bgr = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    ers = F.EdgeRefinedSuperpixel(W, H)
    ers.EdgeRefining(dev(torch_cuda, cl), dev(torch_cuda, dl), dev(torch_cuda, depth), dev(torch_cuda, bgr))
    rl, rd9 = oracle.ers_edge_refining(cl, dl, depth)
    assert np.array_equal(host(ers.getRefinedLabels_Device()), rl)
    assert np.array_equal(host(ers.getEdgeStageDepth_Device()), rd9)
    assert_k10_stagewise(cl, dl, depth, bgr, host(ers.getRefinedDepth_Device()), what="K10 crafted")


def test_k10_flat_patch_nan_quirk(torch_cuda, F, oracle):
    """Q6: exactly flat depth + flat colour -> 0/0 after 47 valid taps: NaNs must coincide with the oracle's."""
    H, W = 24, 40
    depth = np.full((H, W), 1024.0, np.float32)
    bgr = np.full((H, W, 3), 9, np.uint8)
    lab = np.zeros((H, W), np.int32)
    ers = F.EdgeRefinedSuperpixel(W, H)
    ers.EdgeRefining(dev(torch_cuda, lab), dev(torch_cuda, lab), dev(torch_cuda, depth), dev(torch_cuda, bgr))
    ref = oracle.ers_enhance(depth, bgr, lab)
    got = host(ers.getRefinedDepth_Device())
    assert np.isnan(ref).sum() > 0
    assert_depth_close(got, ref, 1e-4, what="K10 NaN quirk")
    assert_k10_stagewise(lab, lab, depth, bgr, got, what="K10 NaN quirk")


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_k10_every_kernel_variant(torch_cuda, F, oracle, synth, frame, variant):
    """packed-pair / scalar tuned / generic depthmap_enhancement kernels against the oracle: natural-looking
    frame with holes, the crafted speckle case, the flat-patch NaN quirk and a ragged size."""
    cases = []
    bgr, depth, K, pts = _inputs(oracle, synth, frame, 31, 203, 77)
    sp = oracle.dasp_segmentation(bgr, pts, 5, 7, K, 200.0, 40.0, 0.0, 1)[0]
    da = oracle.dasp_segmentation(bgr, pts, 5, 7, K, 100.0, 20.0, 200.0, 1)[0]
    cases.append(("synth", sp, da, depth, bgr))
    rng = np.random.default_rng(17)
    H, W = 51, 97                                                  # odd width: the last pixel pair is half outside
    cl = (np.arange(W)[None, :] // 7 + 13 * (np.arange(H)[:, None] // 5)).astype(np.int32)
    dl = ((np.arange(W)[None, :] + 2) // 7 + 13 * ((np.arange(H)[:, None] + 3) // 5)).astype(np.int32)
    d2 = (800 + 40 * dl + rng.normal(0, 30, (H, W))).astype(np.float32)
    d2[rng.random((H, W)) < 0.25] = 0                              # many invalid taps: ranks differ from tap indices
    b2 = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    b2[:, 40:60] = 77                                              # a flat-colour band (cd == 0 taps)
    cases.append(("crafted", cl, dl, d2, b2))
    flat_d = np.full((24, 40), 1024.0, np.float32)
    flat_d[5:9, 7:30] = 0
    cases.append(("flat", np.zeros((24, 40), np.int32), np.zeros((24, 40), np.int32), flat_d, np.full((24, 40, 3), 9, np.uint8)))
    for name, cl_, dl_, d_, b_ in cases:
        h, w = d_.shape
        ers = F.EdgeRefinedSuperpixel(w, h)
        ers.set_variant(variant)
        ers.EdgeRefining(dev(torch_cuda, cl_), dev(torch_cuda, dl_), dev(torch_cuda, d_), dev(torch_cuda, b_))
        rl, rd9 = oracle.ers_edge_refining(cl_, dl_, d_)
        assert np.array_equal(host(ers.getRefinedLabels_Device()), rl)
        assert np.array_equal(host(ers.getEdgeStageDepth_Device()), rd9)
        got = host(ers.getRefinedDepth_Device())
        if name == "flat":
            assert np.isnan(got).sum() > 0 and np.array_equal(np.isnan(got), np.isnan(oracle.ers_enhance(rd9, b_, rl)))
        assert_k10_stagewise(cl_, dl_, d_, b_, got, variant=variant, what=f"K10 variant {variant} {name}")
    from kinectdepthmapenhancement_amd import KdeError
    with pytest.raises(KdeError):
        F.EdgeRefinedSuperpixel(32, 32).set_variant(9)


def test_k10_depth_rule_elision_both_bodies(torch_cuda, F, oracle):
    """depthmap_enhancement's colour-free rows drop the depth-factor underflow rule when the tile's staged depth range
    proves it cannot trip (< 1009 mm at DepthSigma 70).  Left half: a gentle slope (rule elided).  Right half: stripes
    1.7 m apart, so taps on the other side of a stripe edge DO underflow and are skipped (Q1) -- the full body.  Both
    against the oracle, every kernel variant."""
    h, w = 96, 256
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:h, 0:w]
    depth = (1500.0 + 0.3 * xx + 0.2 * yy + rng.normal(0, 1.5, (h, w))).astype(np.float32)
    stripes = (xx >= w // 2) & (((xx // 5) % 2) == 1)
    depth[stripes] += 1700.0
    depth[rng.random((h, w)) < 0.02] = 0
    bgr = np.clip(np.full((h, w, 3), 100, np.int32) + rng.integers(-20, 21, (h, w, 3)), 0, 255).astype(np.uint8)
    lab = ((yy // 24) * 8 + xx // 32).astype(np.int32)
    rl, rd9 = oracle.ers_edge_refining(lab, lab, depth)
    from tools.hooks import stage
    ers = F.EdgeRefinedSuperpixel(w, h)
    for v in (0, 1, 2, 3):
        ers.set_variant(v)
        ers.EdgeRefining(dev(torch_cuda, lab), dev(torch_cuda, lab), dev(torch_cuda, depth), dev(torch_cuda, bgr))
        assert np.array_equal(host(ers.getEdgeStageDepth_Device()), rd9)
        got = host(ers.getRefinedDepth_Device())
        assert np.isfinite(got).all()
        assert_k10_stagewise(lab, lab, depth, bgr, got, variant=v, what=f"K10 depth-rule elision, variant {v}")
        if v in (0, 1):
            # the packed kernel's tile-level shortcuts (no depth rule in the colour-free rows; no deviation pass) are
            # algebraically identical to the full body: forcing them off must not change a bit, and both kinds of tile ran
            s0 = stage.ers_stage_run(lab, lab, depth, bgr, v)
            sf = stage.ers_stage_run(lab, lab, depth, bgr, v, force_full_rules=True)
            assert stage.bits_equal(sf["depth"], got)
            assert s0["counters"][6] > 0 and s0["counters"][7] > 0, s0["counters"]
            assert sf["counters"][6] == 0 and sf["counters"][4] == 0


def test_ers_non_finite_and_huge_depth_samples(torch_cuda, F, oracle, synth, frame):
    """Input domain of depthmap_enhancement (as tests/test_gpu_jbf.py::test_non_finite_and_huge_depth_samples for K1):
    +inf / 3e38 mm samples make their 7x7 window non-finite in the reference.  The generic kernels (variant 3) reproduce
    that class for class; the tuned ones (weights at 2^24 scale) guarantee every pixel whose window holds no such sample,
    and samples up to 2^64 mm everywhere.  Labels and edge_refining are exact regardless."""
    bgr, depth = frame(6, 120, 80)
    h, w = depth.shape
    K = synth.intrinsics(w, h)
    pts = oracle.p2r_depth(depth, K)
    cl = oracle.dasp_segmentation(bgr, pts, 5, 6, K, 200.0, 40.0, 0.0, 1)[0]
    dl = oracle.dasp_segmentation(bgr, pts, 5, 6, K, 100.0, 20.0, 200.0, 1)[0]
    spots = {(10, 10): np.inf, (30, 50): np.inf, (31, 52): -np.inf, (50, 20): 3.0e38, (12, 70): np.nan}
    hostile = depth.copy()
    for (y, x), v in spots.items():
        hostile[y, x] = v
    big = depth.copy()
    big[40, 40] = 2.0 ** 64
    classes = lambda a: np.where(np.isnan(a), 2, np.where(np.isinf(a), 3, 0))
    ers = F.EdgeRefinedSuperpixel(w, h)

    def run(v, d):
        ers.set_variant(v)
        ers.EdgeRefining(dev(torch_cuda, cl), dev(torch_cuda, dl), dev(torch_cuda, d), dev(torch_cuda, bgr))
        return host(ers.getRefinedLabels_Device()).copy(), host(ers.getEdgeStageDepth_Device()).copy(), host(ers.getRefinedDepth_Device()).copy()

    rl, rd9 = oracle.ers_edge_refining(cl, dl, hostile)
    with oracle.ers_flags((h, w)) as env:
        ref = oracle.ers_enhance(rd9, bgr, rl)
    touched = np.zeros((h, w), bool)
    for y, x in zip(*np.nonzero(rd9 > 1e30)):
        touched[max(0, y - 3):y + 4, max(0, x - 3):x + 4] = True
    assert np.isfinite(ref[~touched]).all()
    for v in (0, 1, 2, 3):
        gl, g9, g = run(v, hostile)
        assert np.array_equal(gl, rl) and np.array_equal(g9, rd9, equal_nan=True)
        if v == 3:
            assert np.array_equal(classes(g), classes(ref))
        assert_depth_close(np.where(touched, 0, g), np.where(touched, 0, ref), 1e-4, ill=env, what=f"hostile depth, K10 variant {v}")
        assert_k10_stagewise(cl, dl, hostile, bgr, g, variant=v, what=f"hostile depth, K10 variant {v}", ignore=touched)
    rlb, rd9b = oracle.ers_edge_refining(cl, dl, big)
    with oracle.ers_flags((h, w)) as envb:
        refb = oracle.ers_enhance(rd9b, bgr, rlb)
    for v in (0, 1, 2, 3):
        gl, g9, g = run(v, big)
        assert np.array_equal(gl, rlb) and np.array_equal(g9, rd9b)
        assert_k10_stagewise(cl, dl, big, bgr, g, variant=v, what=f"2^64 mm sample, K10 variant {v}")


def test_rgbf_pipeline_on_reference_color_fixture(torch_cuda, F, oracle, color_fixture, synth):
    _, depth = synth.make_frame(1, 640, 480)
    K = synth.intrinsics(640, 480)
    pts = oracle.p2r_depth(depth, K)
    rg = F.RegionGrowingBilateralFilter(640, 480)
    rg.SetParametor(15, 20, K)
    rg.Process(dev(torch_cuda, depth), dev(torch_cuda, pts_as_f32(pts)), dev(torch_cuda, color_fixture))
    with oracle.ers_flags((480, 640)) as ill:
        ref = oracle.rgbf_process(depth, pts, color_fixture, 15, 20, K)
    assert np.array_equal(host(rg.getSPLabels_Device()), ref["sp_labels"])
    assert np.array_equal(host(rg.getDASPLabels_Device()), ref["dasp_labels"])
    assert np.array_equal(host(rg.getRefinedLabels_Device()), ref["refined_labels"])
    got = host(rg.getRefinedDepth_Device())
    assert_k10_stagewise(ref["sp_labels"], ref["dasp_labels"], depth, color_fixture, got, what="RGBF (colour fixture)", band_max=0.003)
    assert_depth_close(got, ref["refined_depth"], 1e-4, ill=ill, what="RGBF vs the float32 restatement (cross-check)", max_flagged=0.01)
    assert np.array_equal(rg.getRefinedDepth_Host(), got, equal_nan=True)


def test_full_chain_config5_vga(torch_cuda, F, oracle, synth, frame):
    """BASELINE config 5 composition at a size the oracle finishes in seconds:
    projectiveToReal -> JBF.Process -> RGBF.Process fed with the JBF output and its back-projection."""
    w, h = 640, 480
    bgr, depth = frame(13, w, h)
    K = synth.intrinsics(w, h)
    t = torch_cuda
    conv = F.DimensionConvertor(); conv.setCameraParameters(K, w, h)
    jbf = F.JointBilateralFilter(w, h)
    rg = F.RegionGrowingBilateralFilter(w, h); rg.SetParametor(15, 20, K)
    color = dev(t, bgr)
    jbf.Process(dev(t, depth), color)
    filt = jbf.getFiltered_Device()
    pts = t.empty((h, w, 3), dtype=t.float32, device="cuda")
    conv.projectiveToReal(filt, pts)
    rg.Process(filt, pts, color)
    got_filt = host(filt)
    smooth = oracle.cv_bilateral(bgr, 5, 30.0, 30.0)
    assert np.array_equal(host(jbf.getSmoothImage_Device()), smooth)
    assert_k1_stagewise(jbf.params, depth, smooth, got_filt, what="chain JBF", band_max=0.003)
    # downstream stages are compared on the GPU's own JBF output (labels are discontinuous in their input)
    opts = oracle.p2r_depth(got_filt, K)
    assert np.array_equal(host(pts), pts_as_f32(opts))
    ref = oracle.rgbf_process(got_filt, opts, bgr, 15, 20, K)
    assert np.array_equal(host(rg.getRefinedLabels_Device()), ref["refined_labels"])
    assert_k10_stagewise(ref["sp_labels"], ref["dasp_labels"], got_filt, bgr, host(rg.getRefinedDepth_Device()), what="chain RGBF",
                         band_max=0.003)


def test_full_chain_config5_1080p_properties(torch_cuda, F, synth):
    """the 1080p chain (rows=15, cols=20) runs and satisfies size-independent properties; init_LD's
    H/32 grid in the reference leaves rows 1056-1079 uninitialised (F8) — here every row is labelled."""
    w, h = 1920, 1080
    t = torch_cuda
    bgr, depth = synth.make_frame(21, w, h)
    K = synth.intrinsics(w, h)
    conv = F.DimensionConvertor(); conv.setCameraParameters(K, w, h)
    jbf = F.JointBilateralFilter(w, h)
    rg = F.RegionGrowingBilateralFilter(w, h); rg.SetParametor(15, 20, K)
    color, d = dev(t, bgr), dev(t, depth)
    jbf.Process(d, color)
    filt = jbf.getFiltered_Device()
    pts = t.empty((h, w, 3), dtype=t.float32, device="cuda")
    conv.projectiveToReal(filt, pts)
    rg.Process(filt, pts, color)
    lab = rg.getRefinedLabels_Device()
    assert int(lab.min()) >= -1 and int(lab.max()) < 300
    assert int((lab[1056:] >= 0).sum()) > 0.5 * lab[1056:].numel()
    out = rg.getRefinedDepth_Device()
    fin = out[t.isfinite(out)]
    assert fin.max() <= filt.max() * 1.00001 and (out != 0).float().mean() > 0.8
    again = out.clone()
    rg.Process(filt, pts, color)
    assert t.equal(t.nan_to_num(rg.getRefinedDepth_Device()), t.nan_to_num(again))      # deterministic (race-free)


def test_spdsr_process_head_and_tail(torch_cuda, F, oracle, synth, frame):
    """SPDepthSuperResolution::Process end to end: head (DASPx2 with 5 iterations, ERS, back-projection) and tail
    (per-superpixel PCA plane + Projection_GPU::PlaneProjection with 20 sweeps, SPDepthSuperResolution.cpp:65-190)."""
    w, h = 320, 240
    bgr, depth, K, pts = _inputs(oracle, synth, frame, 14, w, h)
    sp = F.SPDepthSuperResolution(w, h)
    sp.SetParametor(6, 8, K)
    sp.Process(dev(torch_cuda, depth), dev(torch_cuda, pts_as_f32(pts)), dev(torch_cuda, bgr))
    with oracle.ers_flags((h, w)) as ill:
        rl, rd, rp = oracle.spdsr_head(depth, pts, bgr, 6, 8, K)
    assert np.array_equal(host(sp.getRefinedLabels_Device()), rl)
    got = host(sp.getRefinedDepth_Device())
    sp_l = oracle.dasp_segmentation(bgr, pts, 6, 8, K, 200.0, 10.0, 0.0, 5)[0]       # SPDepthSuperResolution.cpp:59-64
    da_l = oracle.dasp_segmentation(bgr, pts, 6, 8, K, 0.0, 10.0, 200.0, 5)[0]
    assert_k10_stagewise(sp_l, da_l, depth, bgr, got, what="SPDSR head depth")
    assert_depth_close(got, rd, 1e-4, ill=ill, what="SPDSR head depth vs the float32 restatement (cross-check)", max_flagged=0.01)
    gpts = host(sp.getEdgeEnhanced3DPoints_Device())
    assert np.array_equal(gpts, pts_as_f32(oracle.p2r_depth(got, K)), equal_nan=True)
    # tail, evaluated by the oracle on the GPU's own head output (plane fits are continuous in it)
    gp = np.ascontiguousarray(gpts).view(oracle.FLOAT3).reshape(h, w)
    nd_ref = oracle.spdsr_cluster_planes(rl, gp, 48)
    nd = host(sp.getClusterND_Device())
    assert np.all(np.abs(nd_ref[:, 0]) < 1.0)                    # every cluster of this frame has a plane
    assert np.allclose(nd, nd_ref, rtol=2e-5, atol=2e-6), np.abs(nd - nd_ref).max()
    assert np.allclose(np.linalg.norm(nd[:, :3], axis=1), 1.0, atol=1e-5)
    pf_ref, opt_ref = oracle.projection_plane(nd, rl, gp, K, 20)        # same planes -> isolates the projection kernels
    pf = host(sp.getPlaneFitted3D_Device())
    fin = np.isfinite(pts_as_f32(pf_ref)).all(-1) & np.isfinite(pf).all(-1)
    assert np.allclose(pf[fin], pts_as_f32(pf_ref)[fin], rtol=1e-5, atol=1e-3)
    opt = host(sp.getOptimizedPoints_Device())
    ro = pts_as_f32(opt_ref)
    fin = np.isfinite(ro).all(-1) & np.isfinite(opt).all(-1)
    assert fin.mean() > 0.99
    assert_depth_close(opt[..., 2][fin], ro[..., 2][fin], 1e-4, what="optimized z")
    assert np.allclose(opt[fin], ro[fin], rtol=2e-4, atol=2e-2)
    assert (ro[..., 2] != gpts[..., 2]).mean() > 0.05                  # the sweeps actually moved points
    assert np.array_equal(sp.getOptimizedPoints_Host(), opt, equal_nan=True)


def test_spdsr_tail_degenerate_clusters(torch_cuda, F, oracle, synth):
    """clusters with fewer than 3 labelled points get the (5,5,5) marker and leave their pixels untouched"""
    w, h = 64, 48
    K = synth.intrinsics(w, h)
    depth = np.zeros((h, w), np.float32)            # every pixel invalid -> DASP labels -1 (depth sigma != 0)
    depth[10:30, 10:50] = 1500.0
    bgr = np.full((h, w, 3), 60, np.uint8)
    pts = oracle.p2r_depth(depth, K)
    sp = F.SPDepthSuperResolution(w, h)
    sp.SetParametor(3, 4, K)
    sp.Process(dev(torch_cuda, depth), dev(torch_cuda, pts_as_f32(pts)), dev(torch_cuda, bgr))
    rl = host(sp.getRefinedLabels_Device())
    gpts = host(sp.getEdgeEnhanced3DPoints_Device())
    gp = np.ascontiguousarray(gpts).view(oracle.FLOAT3).reshape(h, w)
    nd_ref = oracle.spdsr_cluster_planes(rl, gp, 12)
    nd = host(sp.getClusterND_Device())
    assert (nd_ref[:, 0] == 5.0).any()
    marker = nd_ref[:, 0] == 5.0
    assert np.array_equal(nd[marker, :3], nd_ref[marker, :3])
    # (the flat patch makes K10 emit NaNs (Q6); they propagate into the moments of their cluster on both sides)
    assert np.allclose(nd[~marker], nd_ref[~marker], rtol=2e-5, atol=2e-6, equal_nan=True)


def test_chain_is_hip_graph_capturable(torch_cuda, F, oracle, synth, frame):
    """every entry point is launch-only (no allocation / synchronisation), so a launch-bound single-frame
    chain can be captured once into a hipGraph and replayed (cdna_hip_programming.md Guideline 9)."""
    t = torch_cuda
    w, h = 640, 480
    bgr, depth = frame(13, w, h)
    K = synth.intrinsics(w, h)
    conv = F.DimensionConvertor(); conv.setCameraParameters(K, w, h)
    jbf = F.JointBilateralFilter(w, h)
    rg = F.RegionGrowingBilateralFilter(w, h); rg.SetParametor(15, 20, K)
    color, d = dev(t, bgr), dev(t, depth)
    pts = t.empty((h, w, 3), dtype=t.float32, device="cuda")
    filt = jbf.getFiltered_Device()

    def chain():
        jbf.Process(d, color)
        conv.projectiveToReal(filt, pts)
        rg.Process(filt, pts, color)

    chain()
    t.cuda.synchronize()
    eager_depth = rg.getRefinedDepth_Device().clone()
    eager_labels = rg.getRefinedLabels_Device().clone()
    side = t.cuda.Stream()
    side.wait_stream(t.cuda.current_stream())
    with t.cuda.stream(side):
        chain()                                        # warm-up on the capture stream
        g = t.cuda.CUDAGraph()
        with t.cuda.graph(g, stream=side):
            chain()
    t.cuda.current_stream().wait_stream(side)
    rg.getRefinedDepth_Device().zero_()
    d2 = dev(t, frame(15, w, h)[1])                    # new input through the same device buffer
    d.copy_(d2)
    g.replay()
    t.cuda.synchronize()
    jbf2 = F.JointBilateralFilter(w, h)
    rg2 = F.RegionGrowingBilateralFilter(w, h); rg2.SetParametor(15, 20, K)
    jbf2.Process(d, color)
    p2 = t.empty_like(pts)
    conv.projectiveToReal(jbf2.getFiltered_Device(), p2)
    rg2.Process(jbf2.getFiltered_Device(), p2, color)
    assert t.equal(rg.getRefinedLabels_Device(), rg2.getRefinedLabels_Device())
    assert t.equal(t.nan_to_num(rg.getRefinedDepth_Device()), t.nan_to_num(rg2.getRefinedDepth_Device()))
    assert not t.equal(rg.getRefinedLabels_Device(), eager_labels) or not t.equal(t.nan_to_num(eager_depth), t.nan_to_num(rg.getRefinedDepth_Device()))


def test_k7_integer_sqrt_is_sqrtf_on_its_whole_domain(torch_cuda):
    """calculateLD's spatial distance is sqrtf(px*px + py*py) of integer pixel offsets; the kernels compute it with a
    6-instruction square root that is only claimed exact for integer arguments below 2^24 -- all of them are checked."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("kde_hooks", os.path.join(ROOT, "tools", "hooks", "hooks.py"))
    hooks = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hooks)          # tools/hooks/libkde_hooks.so compiles the SAME kde_device_math.h the kernels use
    n = 1 << 24
    out = torch_cuda.empty(n, dtype=torch_cuda.float32, device="cuda")
    assert hooks.lib().kde_test_sqrt_int24(0, n, out.data_ptr(), torch_cuda.cuda.current_stream().cuda_stream) == 0
    got = out.cpu().numpy()
    ref = np.sqrt(np.arange(n, dtype=np.float32))        # IEEE correctly rounded
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), int((got != ref).sum())


@pytest.mark.parametrize("sig", [(100.0, 0.0, 0.0), (100.0, 0.0, 50.0), (0.0, 0.0, 200.0)])
def test_k7_tie_break_follows_the_reference_tree(torch_cuda, F, oracle, synth, sig):
    """With a constant colour image and no spatial term every in-grid candidate has the same distance: the label is
    decided by the tie-break of the reference's 16-way tree (strict '>' at every level = bit-reversed index order),
    which the kernel reproduces with a scan in that order -- including after analyzeClusters moved the centres."""
    w, h = 160, 120
    _, depth = synth.make_frame(3, w, h)
    depth = (np.round(depth / 500.0) * 500.0).astype(np.float32)          # few distinct depths: ties in the depth term too
    bgr = np.full((h, w, 3), 77, np.uint8)
    K = synth.intrinsics(w, h)
    pts = oracle.p2r_depth(depth, K)
    for it in (1, 3):
        d = F.DepthAdaptiveSuperpixel(w, h)
        d.SetParametor(6, 8, K)
        d.Segmentation(dev(torch_cuda, bgr), dev(torch_cuda, pts_as_f32(pts)), *sig, it)
        labels, ld, mean, centers = oracle.dasp_segmentation(bgr, pts, 6, 8, K, *sig, it)
        assert np.array_equal(host(d.getLabelDevice()), labels), (sig, it)
        gl = ld_records(d.getLDDevice())
        assert np.array_equal(gl["l"], ld["l"]) and np.array_equal(gl["d"], ld["d"])
        assert len(np.unique(labels)) > 3                                    # not a degenerate single label
    with pytest.raises(Exception):
        F.DepthAdaptiveSuperpixel(w, h).Segmentation(dev(torch_cuda, bgr), dev(torch_cuda, pts_as_f32(pts)), 0.0, 0.0, 0.0, 1)


@pytest.mark.parametrize("size,n,grid", [((160, 120), 5, (6, 8)), ((203, 77), 3, (5, 7)), ((640, 480), 4, (15, 20))])
def test_rgbf_batch_is_bit_identical_to_single_frame_calls(torch_cuda, F, oracle, synth, size, n, grid):
    """kde_rgbf_process_batch: every kernel of the chain takes the whole batch per launch; each frame's
    SP / DASP / refined labels and refined depth equal its single-frame Process to the bit, in any batch position."""
    w, h = size
    bgr, depth = synth.make_batch(700, n, w, h)
    K = synth.intrinsics(w, h)
    t = torch_cuda
    pts = np.stack([pts_as_f32(oracle.p2r_depth(depth[f], K)) for f in range(n)])
    rgb = F.RegionGrowingBilateralFilter(w, h, max_batch=n)
    rgb.SetParametor(grid[0], grid[1], K)
    rgb.process_batch(dev(t, depth), dev(t, pts), dev(t, bgr))
    got = {k: host(getattr(rgb, g)()).copy() for k, g in (("sp", "getSPLabels_Device"), ("da", "getDASPLabels_Device"),
                                                           ("rl", "getRefinedLabels_Device"), ("rd", "getRefinedDepth_Device"))}
    assert got["rd"].shape == (n, h, w)
    rg1 = F.RegionGrowingBilateralFilter(w, h)
    rg1.SetParametor(grid[0], grid[1], K)
    for f in range(n):
        rg1.Process(dev(t, depth[f]), dev(t, pts[f]), dev(t, bgr[f]))
        assert np.array_equal(host(rg1.getSPLabels_Device()), got["sp"][f])
        assert np.array_equal(host(rg1.getDASPLabels_Device()), got["da"][f])
        assert np.array_equal(host(rg1.getRefinedLabels_Device()), got["rl"][f])
        assert np.array_equal(host(rg1.getRefinedDepth_Device()).view(np.uint32), got["rd"][f].view(np.uint32))
    assert np.array_equal(rgb.getRefinedDepth_Host().view(np.uint32), got["rd"].view(np.uint32))      # host getter: n frames
    # a shorter batch on the same object, in another order: frames are independent units
    perm = list(range(n - 1))[::-1]
    rgb.process_batch(dev(t, depth[perm]), dev(t, pts[perm]), dev(t, bgr[perm]))
    assert np.array_equal(host(rgb.getRefinedDepth_Device()).view(np.uint32), got["rd"][perm].view(np.uint32))
    # and against the oracle (frame 0): labels exact, depth stage-wise
    ref = oracle.rgbf_process(depth[0], oracle.p2r_depth(depth[0], K), bgr[0], grid[0], grid[1], K)
    assert np.array_equal(got["rl"][0], ref["refined_labels"])
    assert_k10_stagewise(ref["sp_labels"], ref["dasp_labels"], depth[0], bgr[0], got["rd"][0], what=f"RGBF batch frame 0 {w}x{h}")
    from kinectdepthmapenhancement_amd import KdeError
    with pytest.raises(KdeError):
        rgb.process_batch(dev(t, np.concatenate([depth, depth])), dev(t, np.concatenate([pts, pts])), dev(t, np.concatenate([bgr, bgr])))


def test_spdsr_batch_matches_single_frame_calls(torch_cuda, F, oracle, synth):
    """kde_spdsr_process_batch: head (5 assignment iterations, ERS, back-projection) bit-identical per frame; the tail's
    plane fit sums double-precision moments with atomics (order not fixed), so planes / optimised points agree to 1e-4"""
    w, h, n = 160, 120, 3
    bgr, depth = synth.make_batch(720, n, w, h)
    K = synth.intrinsics(w, h)
    t = torch_cuda
    pts = np.stack([pts_as_f32(oracle.p2r_depth(depth[f], K)) for f in range(n)])
    srb = F.SPDepthSuperResolution(w, h, max_batch=n)
    srb.SetParametor(6, 8, K)
    srb.process_batch(dev(t, depth), dev(t, pts), dev(t, bgr))
    rl, rd = host(srb.getRefinedLabels_Device()).copy(), host(srb.getRefinedDepth_Device()).copy()
    ep, nd = host(srb.getEdgeEnhanced3DPoints_Device()).copy(), host(srb.getClusterND_Device()).copy()
    opt = host(srb.getOptimizedPoints_Device()).copy()
    assert nd.shape == (n, 48, 4) and opt.shape == (n, h, w, 3)
    for f in range(n):
        # a fresh object per frame: a cluster without a plane keeps the distance its ClusterND slot held before
        # (SPDepthSuperResolution.cpp:139-142 leaves the pinned entry alone), and every slot of the batch has its own table
        sr1 = F.SPDepthSuperResolution(w, h)
        sr1.SetParametor(6, 8, K)
        sr1.Process(dev(t, depth[f]), dev(t, pts[f]), dev(t, bgr[f]))
        assert np.array_equal(host(sr1.getRefinedLabels_Device()), rl[f])
        assert np.array_equal(host(sr1.getRefinedDepth_Device()).view(np.uint32), rd[f].view(np.uint32))
        assert np.array_equal(host(sr1.getEdgeEnhanced3DPoints_Device()).view(np.uint32), ep[f].view(np.uint32))
        assert np.allclose(host(sr1.getClusterND_Device()), nd[f], rtol=2e-5, atol=2e-6)
        o1 = host(sr1.getOptimizedPoints_Device())
        fin = np.isfinite(o1).all(-1) & np.isfinite(opt[f]).all(-1)
        assert np.array_equal(np.isfinite(o1), np.isfinite(opt[f])) and np.allclose(o1[fin], opt[f][fin], rtol=1e-4, atol=1e-2)
    assert np.array_equal(srb.getOptimizedPoints_Host(), opt, equal_nan=True)


def test_fastdiv24_device_function_equals_integer_division(torch_cuda):
    """tests/test_oracle_micro.py re-implements the magic-number formula in Python; this calls the real
    make_fastdiv24 (host) + fastdiv24 (device) of csrc/kde_device_math.h through tools/hooks: every divisor a launcher can
    form (tiles per frame, tiles per row, cell sizes, grid columns) at the edges of its dividend range, random pairs, and
    the fallback (ok == 0: divisor or dividend bound >= 2^24), which must be a plain division."""
    import ctypes
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("kde_hooks", os.path.join(ROOT, "tools", "hooks", "hooks.py"))
    hooks = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hooks)
    t = torch_cuda
    rng = np.random.default_rng(3)
    st = t.cuda.current_stream().cuda_stream

    def probe(d, maxdiv, xs):
        xs = np.asarray(xs, np.uint32)
        x = t.from_numpy(xs.view(np.int32)).cuda()
        out = t.empty_like(x)
        info = (ctypes.c_uint32 * 3)()
        assert hooks.lib().kde_test_fastdiv24(int(d), int(maxdiv), xs.size, x.data_ptr(), out.data_ptr(), info, st) == 0
        return out.cpu().numpy().view(np.uint32), tuple(info)

    divisors = sorted(set([1, 2, 3, 5, 7, 10, 20, 32, 40, 60, 63, 64, 65, 72, 96, 300, 1200, 4080, 8100, 65535, 65536, 65537,
                           (1 << 23) - 1, 1 << 23, (1 << 24) - 1] + rng.integers(1, 1 << 24, 40).tolist()))
    for d in divisors:
        top = (1 << 24) - 1
        xs = np.unique(np.concatenate([[0, 1, d - 1, d, d + 1, top - 1, top], (np.arange(1, 40) * d - 1) % (top + 1),
                                       (np.arange(1, 40) * d) % (top + 1), rng.integers(0, top + 1, 4000)])).astype(np.uint32)
        got, (m, sh, ok) = probe(d, top, xs)
        assert ok == 1, d
        assert np.array_equal(got, xs // np.uint32(d)), d
    # exhaustive for a few divisors over the WHOLE 24-bit dividend range
    allx = np.arange(1 << 24, dtype=np.uint32)
    for d in (3, 40, 8100, 4079):
        got, _ = probe(d, (1 << 24) - 1, allx)
        assert np.array_equal(got, allx // np.uint32(d)), d
    # fallback: the proven range is left -> ok == 0 and the result is the hardware division, for any 32-bit dividend
    big = np.concatenate([[0, 1, (1 << 24), (1 << 31), 0xFFFFFFFF], rng.integers(0, 1 << 32, 4000, dtype=np.uint64)]).astype(np.uint32)
    for d, maxdiv in ((7, 1 << 24), (40, 1 << 30), (1 << 24, (1 << 24) - 1), (3_000_000_000 % (1 << 32), 1 << 20)):
        got, (m, sh, ok) = probe(d, maxdiv, big)
        assert ok == 0, (d, maxdiv)
        assert np.array_equal(got, big // np.uint32(d)), (d, maxdiv)


def test_spdsr_object_reuse_starts_from_clean_moment_tables(torch_cuda, F, oracle, synth):
    """The per-cluster moment tables are accumulated with atomics into zeroed memory and cleared by the kernel that consumes
    them (no memset launches per frame): a second, different frame — and a smaller batch after a larger one — on the SAME
    object must give the planes a fresh object gives."""
    w, h = 160, 120
    t = torch_cuda
    K = synth.intrinsics(w, h)
    bgr, depth = synth.make_batch(760, 4, w, h)
    pts = np.stack([pts_as_f32(oracle.p2r_depth(depth[f], K)) for f in range(4)])

    def fresh(f):
        sr = F.SPDepthSuperResolution(w, h)
        sr.SetParametor(6, 8, K)
        sr.Process(dev(t, depth[f]), dev(t, pts[f]), dev(t, bgr[f]))
        return host(sr.getClusterND_Device()).copy(), host(sr.getOptimizedPoints_Device()).copy()

    def same(a, b):
        nd_a, o_a = a
        nd_b, o_b = b
        planes = (np.abs(nd_a[..., 0]) < 1.0) & (np.abs(nd_b[..., 0]) < 1.0)
        assert np.array_equal(np.abs(nd_a[..., 0]) < 1.0, np.abs(nd_b[..., 0]) < 1.0)
        assert np.allclose(nd_a[planes], nd_b[planes], rtol=2e-5, atol=2e-6)
        fin = np.isfinite(o_a).all(-1) & np.isfinite(o_b).all(-1)
        assert np.array_equal(np.isfinite(o_a), np.isfinite(o_b)) and np.allclose(o_a[fin], o_b[fin], rtol=1e-4, atol=1e-2)

    one = F.SPDepthSuperResolution(w, h)
    one.SetParametor(6, 8, K)
    for f in (0, 1, 0):                                  # A, B, A again on one object
        one.Process(dev(t, depth[f]), dev(t, pts[f]), dev(t, bgr[f]))
        same((host(one.getClusterND_Device()), host(one.getOptimizedPoints_Device())), fresh(f))
    many = F.SPDepthSuperResolution(w, h, max_batch=4)
    many.SetParametor(6, 8, K)
    many.process_batch(dev(t, depth), dev(t, pts), dev(t, bgr))                    # 4 frames ...
    many.process_batch(dev(t, depth[2:4]), dev(t, pts[2:4]), dev(t, bgr[2:4]))      # ... then 2 others in slots 0, 1
    nd, opt = host(many.getClusterND_Device()), host(many.getOptimizedPoints_Device())
    for slot, f in ((0, 2), (1, 3)):
        same((nd[slot], opt[slot]), fresh(f))


def test_spdsr_two_sweeps_per_launch_equal_single_sweeps_bitwise(torch_cuda, tmp_path):
    """The measured-and-rejected two-sweeps-per-launch form of Projection_GPU's 20 mrf_optimization sweeps
    (mrf_sweep2_kernel; only in the measurement build tools/hooks/libkde_hip_ab.so, selected there by KDE_SPDSR_TWO_SWEEPS=1,
    read once per process): per pixel the arithmetic is the single-sweep kernel's, so the optimised cloud must not change by
    a bit against the PRODUCT library's.  Ragged size: partial tiles and an odd width."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import os, sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "if os.environ.get('KDE_SPDSR_TWO_SWEEPS'):\n"
        "    sys.path.insert(0, os.path.join(sys.path[0], 'tools', 'hooks')); import ab; ab.use_ab_library()\n"
        "from kinectdepthmapenhancement_amd import filters as F, synth\n"
        "from oracle import oracle as O\n"
        "outs = []\n"
        "for (w, h, g) in ((203, 131, (5, 7)), (320, 240, (6, 8))):\n"
        "    bgr, depth = synth.make_frame(33, w, h); K = synth.intrinsics(w, h)\n"
        "    pts = O.p2r_depth(depth, K).view(np.float32).reshape(h, w, 3)\n"
        "    sr = F.SPDepthSuperResolution(w, h); sr.SetParametor(g[0], g[1], K)\n"
        "    sr.Process(torch.from_numpy(depth).cuda(), torch.from_numpy(np.ascontiguousarray(pts)).cuda(), torch.from_numpy(bgr).cuda())\n"
        "    outs.append(sr.getOptimizedPoints_Device().cpu().numpy().ravel())\n"
        "np.save(sys.argv[1], np.concatenate(outs))\n") % ROOT
    res = []
    for tag, extra in (("two", {"KDE_SPDSR_TWO_SWEEPS": "1"}), ("one", {})):
        path = str(tmp_path / f"opt_{tag}.npy")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(path))
    assert res[0].shape == res[1].shape and np.isfinite(res[0]).mean() > 0.9
    assert np.array_equal(res[0].view(np.uint32), res[1].view(np.uint32))


_RESIDENT_CHILD = (
    "import os, sys, threading, numpy as np, torch; sys.path.insert(0, %r)\n"
    "sys.path.insert(0, os.path.join(sys.path[0], 'tools', 'hooks')); import ab; ab.use_ab_library()\n"
    "from kinectdepthmapenhancement_amd import filters as F, synth\n"
    "from oracle import oracle as O\n"
    "def run(seed, w, h, g, reps, stream=None):\n"
    "    with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):\n"
    "        bgr, depth = synth.make_frame(seed, w, h); K = synth.intrinsics(w, h)\n"
    "        pts = O.p2r_depth(depth, K).view(np.float32).reshape(h, w, 3)\n"
    "        sr = F.SPDepthSuperResolution(w, h); sr.SetParametor(g[0], g[1], K)\n"
    "        a = (torch.from_numpy(depth).cuda(), torch.from_numpy(np.ascontiguousarray(pts)).cuda(), torch.from_numpy(bgr).cuda())\n"
    "        for _ in range(reps): sr.Process(*a)\n"
    "        return np.asarray(sr.getOptimizedPoints_Host()).copy().ravel()\n"
    "cases, threads = %r, int(sys.argv[2])\n"
    "outs = [None] * len(cases)\n"
    "if threads:\n"
    "    run(*cases[0], 1)\n"
    "    th = [threading.Thread(target=lambda i=i: outs.__setitem__(i, run(*cases[i], 25, torch.cuda.Stream()))) for i in range(len(cases))]\n"
    "    [t.start() for t in th]; [t.join(120) for t in th]\n"
    "    assert not any(t.is_alive() for t in th)\n"
    "else:\n"
    "    outs = [run(*c, 4) for c in cases]\n"
    "np.save(sys.argv[1], np.concatenate(outs))\n")


def _resident_child(tmp_path, cases, mode, threads=0):
    import os
    import subprocess
    import sys
    from conftest import ROOT
    path = str(tmp_path / f"opt_resident_{mode}_{threads}.npy")
    r = subprocess.run([sys.executable, "-c", _RESIDENT_CHILD % (ROOT, cases), path, str(threads)], env=dict(os.environ, KDE_SPDSR_RESIDENT=str(mode)),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(path)


def test_spdsr_resident_sweeps_equal_the_20_launches_bitwise(torch_cuda, F, oracle, synth, tmp_path):
    """The measured-and-rejected LDS-resident form of Projection_GPU's 20 mrf_optimization sweeps (mrf_sweeps_resident_kernel:
    one launch, one block of the frame per CU, only the 2-pixel rims go through memory between sweeps; measurement build only,
    KDE_SPDSR_RESIDENT = 1 cooperative / 2 plain launch).  Per pixel the arithmetic is mrf_sweep_kernel's, so the optimised
    cloud must equal the PRODUCT library's 20 launches bit for bit, call after call (the announcement counters run on).
    Sizes: 1080p (blocks of 120 x 68), 640x480, and a ragged frame with an odd width and partial last blocks."""
    cases = ((33, 1920, 1080, (15, 20)), (33, 640, 480, (15, 20)), (33, 203, 131, (5, 7)))
    want = []
    for seed, w, h, g in cases:
        bgr, depth = synth.make_frame(seed, w, h)
        K = synth.intrinsics(w, h)
        pts = oracle.p2r_depth(depth, K).view(np.float32).reshape(h, w, 3)
        sr = F.SPDepthSuperResolution(w, h)
        sr.SetParametor(g[0], g[1], K)
        sr.Process(dev(torch_cuda, depth), dev(torch_cuda, np.ascontiguousarray(pts)), dev(torch_cuda, bgr))
        want.append(host(sr.getOptimizedPoints_Device()).ravel())
    want = np.concatenate(want)
    assert np.isfinite(want).mean() > 0.9
    for mode in (1, 2):
        got = _resident_child(tmp_path, cases, mode)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"resident mode {mode}"


def test_spdsr_resident_cooperative_launches_from_three_host_threads(torch_cuda, F, oracle, synth, tmp_path):
    """three host threads, each with its own stream and SPDSR handle, call Process 25 times concurrently on the measurement build
    with the COOPERATIVE resident launch: the runtime runs cooperative launches of a device one after the other, so every
    workgroup of a launch is resident and no launch waits for another's (a spin that gave up would fail the child); results
    equal the product library's bit for bit"""
    cases = ((41, 640, 480, (15, 20)), (42, 640, 480, (15, 20)), (43, 640, 480, (15, 20)))
    want = _resident_child(tmp_path, cases, 0)
    got = _resident_child(tmp_path, cases, 1, threads=1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_k8_row_coalesced_form_is_bit_identical(torch_cuda, F, oracle, synth, tmp_path):
    """The measured-and-rejected row-coalesced form of analyzeClusters (analyze_clusters_rows_kernel; measurement build only,
    KDE_K8_ROWS=1): coalesced 4-pixel chunks per wavefront row, integer sums by the loading lane, the float sums by the owning
    thread from an LDS stage in the reference's order.  Labels, cluster records and float centres after 2 - 5 iterations must
    equal the PRODUCT kernel's bit for bit (1080p, 640x480, a ragged frame, two other grids)."""
    import os
    import subprocess
    import sys
    import zlib
    from conftest import ROOT
    cases = ((1920, 1080, (15, 20), 5), (640, 480, (15, 20), 5), (203, 131, (5, 7), 3), (640, 480, (10, 8), 2), (320, 240, (6, 8), 4))
    want = {}
    for (w, h, g, it) in cases:
        bgr, depth = synth.make_frame(33, w, h)
        K = synth.intrinsics(w, h)
        pts = oracle.p2r_depth(depth, K).view(np.float32).reshape(h, w, 3)
        d = F.DepthAdaptiveSuperpixel(w, h)
        d.SetParametor(g[0], g[1], K)
        d.Segmentation(dev(torch_cuda, bgr), dev(torch_cuda, np.ascontiguousarray(pts)), 100.0, 20.0, 200.0, it)
        want[f"{w}x{h}_{g}_{it}"] = tuple(zlib.crc32(host(t).tobytes()) for t in (d.getLabelDevice(), d.getMeanDataDevice(), d.getCentersDevice()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "k8_rows_check.py"), ROOT], env=dict(os.environ, KDE_K8_ROWS="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = eval(r.stdout.strip().splitlines()[-1])        # the tool prints one dict literal of CRC triples
    assert got == want
