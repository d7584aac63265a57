#!/usr/bin/env python3
"""BASELINE's workloads at their own sizes on MANY synthetic frames (tests/test_gpu_fullsize.py checks one frame each):
JointBilateralFilter::Process at 640x480 / window 11 (config 2 / 4) and 1920x1080 / window 19 (config 3), and the config-5
chain at 640x480, every frame under the stage-wise check (the oracle is the checker, as in tests/).  Every pixel of every frame has to pass.  The
part of a frame that is interval-checked because a tap lies ON a Q1 decision (BAND without GRID) is ASSERTED <= 1e-4 per frame --
the bound the full-size tests use, here for the generator, not for a few seeds; the denormal-grid part (content: holes whose
whole window differs in colour) is reported.
    python tools/stress_fullsize.py [--vga 16] [--fhd 3] [--chain 6] [--seed 1000]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vga", type=int, default=16)
    ap.add_argument("--fhd", type=int, default=3)
    ap.add_argument("--chain", type=int, default=6)
    ap.add_argument("--seed", type=int, default=1000)
    a = ap.parse_args()
    import torch
    from conftest import assert_k1_stagewise, assert_k10_stagewise
    from gpu_util import dev, host, pts_as_f32
    from kinectdepthmapenhancement_amd import filters as F, synth
    from oracle import oracle as O
    O.build()
    O.set_threads(min(16, os.cpu_count() or 1))
    bad = 0
    fracs = {}       # what -> list of (BAND fraction, strict max rel err, decision fraction)
    DECISION_MAX = 1e-4

    def params(w):
        p = F.JointBilateralFilter.default_params()
        p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma = w, 3.0, 7.65, 20.0
        return p

    for (W, H, win, count, tag) in ((640, 480, 11, a.vga, "config 2/4"), (1920, 1080, 19, a.fhd, "config 3")):
        if count <= 0:
            continue
        bgr, depth = synth.make_batch(a.seed, count, W, H)
        jbf = F.JointBilateralFilter(W, H, params(win), max_batch=count)
        out = host(jbf.process_batch(dev(torch, depth), dev(torch, bgr)))
        smooth = host(jbf.getSmoothImage_Device(count))
        for f in range(count):
            try:
                assert np.array_equal(smooth[f], O.cv_bilateral(bgr[f], 5, 30.0, 30.0)), "K0 bytes differ"
                r = assert_k1_stagewise(jbf.params, depth[f], smooth[f], out[f], what=f"{tag} seed {a.seed + f}", band_max=1.0, decision_max=DECISION_MAX)
                fracs.setdefault(tag, []).append((r["band_frac"], r["max_rel_strict"], r["band_decision_frac"]))
                print(f"ok   {tag} {W}x{H} window {win} seed {a.seed + f}: BAND {r['band_frac']:.2e} (on a decision {r['band_decision_frac']:.2e}), strict {r['max_rel_strict']:.2e}", flush=True)
            except AssertionError as e:
                bad += 1
                print(f"FAIL {tag} seed {a.seed + f}: {str(e)[:300]}", flush=True)
    W, H = 640, 480
    K = synth.intrinsics(W, H)
    for f in range(a.chain):
        bgr, depth = synth.make_frame(a.seed + 500 + f, W, H)
        conv = F.DimensionConvertor(); conv.setCameraParameters(K, W, H)
        jbf = F.JointBilateralFilter(W, H)
        rg = F.RegionGrowingBilateralFilter(W, H); rg.SetParametor(15, 20, K)
        color, d = dev(torch, bgr), dev(torch, depth)
        jbf.Process(d, color)
        filt = jbf.getFiltered_Device()
        pts = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        conv.projectiveToReal(filt, pts)
        rg.Process(filt, pts, color)
        got_filt = host(filt)
        try:
            ref = O.rgbf_process(got_filt, O.p2r_depth(got_filt, K), bgr, 15, 20, K)
            assert np.array_equal(host(rg.getSPLabels_Device()), ref["sp_labels"]), "SP labels differ"
            assert np.array_equal(host(rg.getDASPLabels_Device()), ref["dasp_labels"]), "DASP labels differ"
            assert np.array_equal(host(rg.getRefinedLabels_Device()), ref["refined_labels"]), "refined labels differ"
            r = assert_k10_stagewise(ref["sp_labels"], ref["dasp_labels"], got_filt, bgr, host(rg.getRefinedDepth_Device()),
                                     what=f"config 5 chain seed {a.seed + 500 + f}", band_max=1.0, decision_max=DECISION_MAX)
            fracs.setdefault("config 5 (K10)", []).append((r["band_frac"], r["max_rel_strict"], r["band_decision_frac"]))
            print(f"ok   config 5 chain 640x480 seed {a.seed + 500 + f}: BAND {r['band_frac']:.2e} (on a decision {r['band_decision_frac']:.2e}), strict {r['max_rel_strict']:.2e}", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"FAIL chain seed {a.seed + 500 + f}: {str(e)[:300]}", flush=True)
    for tag, v in fracs.items():
        b = np.array([x[0] for x in v]); e = np.array([x[1] for x in v]); dcs = np.array([x[2] for x in v])
        print(f"{tag}: {len(v)} frames, BAND fraction median {np.median(b):.2e} p90 {np.percentile(b, 90):.2e} max {b.max():.2e} "
              f"({int((b > 0.003).sum())} frames above 0.3 %); of it on a decision (asserted <= {DECISION_MAX:g} per frame): median {np.median(dcs):.2e} "
              f"max {dcs.max():.2e}; strict-pixel max rel err median {np.median(e):.2e} max {e.max():.2e}")
    print(f"stress_fullsize: {a.vga} + {a.fhd} + {a.chain} frames, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
