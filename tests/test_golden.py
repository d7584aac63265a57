"""The CPU oracle against the committed golden vectors (tests/golden/, made by make_golden.py).
These pin the restatement itself; the reference has no vectors of its own (parity unpinned)."""
import json
import os
import zlib

import numpy as np

from conftest import GOLDEN


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def test_color_fixture_and_synth_depth_checksums(color_fixture, synth):
    full = json.load(open(os.path.join(GOLDEN, "golden_fullframe.json")))
    assert color_fixture.shape == (480, 640, 3) and color_fixture.dtype == np.uint8
    assert _crc(color_fixture) == full["color_crc32"]
    _, depth = synth.make_frame(1, 640, 480)
    assert _crc(depth) == full["depth_crc32"]


def test_oracle_full_frame_checksums(oracle, color_fixture, synth):
    """BASELINE config 1: the single 640x480 frame through the CPU path (plumbing, no GPU)."""
    full = json.load(open(os.path.join(GOLDEN, "golden_fullframe.json")))
    _, depth = synth.make_frame(1, 640, 480)
    K = synth.intrinsics(640, 480)
    filt, smooth, ill = oracle.jbf_process(depth, color_fixture, return_all=True)
    assert _crc(smooth) == full["smooth_crc32"]
    assert _crc(filt) == full["jbf"]["crc32"] and int((filt == 0).sum()) == full["jbf"]["zeros"]
    assert int(ill.flagged.sum()) == full["jbf_flagged"]
    pts = oracle.p2r_depth(depth, K)
    assert _crc(pts.view(np.float32)) == full["points"]["crc32"]
    rg = oracle.rgbf_process(depth, pts, color_fixture, 15, 20, K)
    assert _crc(rg["sp_labels"]) == full["sp_labels_crc32"]
    assert _crc(rg["dasp_labels"]) == full["dasp_labels_crc32"]
    assert _crc(rg["refined_labels"]) == full["refined_labels_crc32"]
    assert _crc(rg["refined_depth"]) == full["rgbf_refined_depth"]["crc32"]


def test_oracle_crops(oracle, synth):
    g = np.load(os.path.join(GOLDEN, "golden_crops.npz"))
    cb, cd = g["bgr"], g["depth"]
    Kc = synth.intrinsics(64, 48)
    assert np.array_equal(oracle.cv_bilateral(cb, 5, 30.0, 30.0), g["k0_smooth"])
    assert np.array_equal(oracle.jbf_kernel(cd, g["k0_smooth"]), g["k1_jbf_ref_params"])
    assert np.array_equal(oracle.jbf_process(cd, cb), g["jbf_process"])
    assert np.array_equal(oracle.jbf_kernel(cd, cb, 11, 3.0, 7.65, 20.0), g["k1_jbf_w11_s3_c7p65"])
    assert np.array_equal(oracle.mrf_kernel(cd, cb), g["mrf"])
    cp = oracle.p2r_depth(cd, Kc)
    assert np.array_equal(cp.view(np.float32).reshape(48, 64, 3), g["k2_points"])
    labels, ld, mean, centers = oracle.dasp_segmentation(cb, cp, 3, 4, Kc, 100.0, 20.0, 200.0, 1)
    assert np.array_equal(labels, g["k7_dasp_labels"])
    assert np.array_equal(mean.view(np.uint8).reshape(-1, 16), g["k8_dasp_mean"])
    rl, rd = oracle.ers_edge_refining(g["k7_sp_labels"], g["k7_dasp_labels"], cd)
    assert np.array_equal(rl, g["k9_labels"]) and np.array_equal(rd, g["k9_depth"])
    assert np.array_equal(oracle.ers_enhance(rd, cb, rl), g["k10_depth"], equal_nan=True)
