"""Stage-wise parity probe (test infrastructure): K1 / K10 on the product library and on tools/hooks/libkde_hip_stage.so,
bit-identity of the two final outputs, then oracle.stage_check on the GPU's own first-pass average / deviation.
usage: python tools/stage_probe.py [--json out.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def brief(r):
    return {k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in r.items() if k not in ("bad", "rel", "avg_unchecked")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--fhd", action="store_true", help="also 1080p window 19 (config 3)")
    args = ap.parse_args()
    import torch
    from kinectdepthmapenhancement_amd import filters as F, synth
    from oracle import oracle as O
    from tools.hooks import stage
    O.build()
    O.set_threads(min(16, os.cpu_count() or 1))
    results = []
    names = F.JointBilateralFilter.variants()

    def k1(tag, depth, bgr, w, ss, cs, ds, variants, presmooth=True):
        h, wd = depth.shape
        p = F.JointBilateralFilter.default_params()
        p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma, p.presmooth = w, ss, cs, ds, 0
        guide = O.cv_bilateral(bgr, 5, 30.0, 30.0) if presmooth else bgr
        for v in variants:
            jbf = F.JointBilateralFilter(wd, h, p)
            jbf.set_variant(v)
            out = torch.empty((1, h, wd), dtype=torch.float32, device="cuda")
            jbf.filter_batch(torch.from_numpy(depth[None]).cuda(), torch.from_numpy(guide[None]).cuda(), out)
            got = out.cpu().numpy()[0]
            jbf.close()
            so, avg, cnt = stage.jbf_stage_run(p, depth[None], guide[None], v)
            same = stage.bits_equal(so[0], got)
            sf, _, cntf = stage.jbf_stage_run(p, depth[None], guide[None], v, force_full_rules=True)
            same_forced = stage.bits_equal(sf[0], got)
            t = time.time()
            st = O.jbf_stage(depth, guide, w, ss, cs, ds, avg_in=avg[0])
            r = O.stage_check(got, st)
            rec = dict(kernel="K1", case=tag, variant=names[v] if v >= 0 else "auto", stage_bits_equal=same,
                       forced_bits_equal=same_forced, bodies=cnt[:4].tolist(), bodies_forced=cntf[:4].tolist(),
                       bad=int(r["bad"].sum()), oracle_s=round(time.time() - t, 2), **brief(r))
            print(json.dumps(rec), flush=True)
            results.append(rec)

    bgr, depth = synth.make_frame(1, 640, 480)
    from PIL import Image
    fix = np.ascontiguousarray(np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "color_640x480.png")).convert("RGB"))[..., ::-1])
    vw = {w: [i for i, nm in enumerate(names) if i > 0 and nm.startswith(f"w{w}-")] for w in (5, 7, 11, 19)}
    k1("config2 fixture w11", depth, fix, 11, 3.0, 7.65, 20.0, [-1, 0] + vw[11])
    k1("synthetic w11", depth, bgr, 11, 3.0, 7.65, 20.0, [-1])
    k1("fixture w5 ref", depth, fix, 5, 70.0, 50.0, 20.0, [-1, 0] + vw[5])
    k1("synthetic w19", depth, bgr, 19, 3.0, 7.65, 20.0, [-1, 0] + vw[19])
    k1("synthetic w7", depth, bgr, 7, 30.0, 50.0, 70.0, [-1] + vw[7])
    k1("depth-outliers w11", depth, bgr, 11, 5.0, 20.0, 4.0, [-1, 0], presmooth=False)
    if args.fhd:
        b2, d2 = synth.make_frame(3, 1920, 1080)
        k1("config3 1080p w19", d2, b2, 19, 3.0, 7.65, 20.0, [-1])

    # ---- K10 through EdgeRefinedSuperpixel::EdgeRefining on segmentations of the oracle (labels are exact) ----
    def k10(tag, w, h, seed, rows, cols):
        bgr, depth = synth.make_frame(seed, w, h)
        K = synth.intrinsics(w, h)
        pts = O.p2r_depth(depth, K)
        ref = O.rgbf_process(depth, pts, bgr, rows, cols, K)
        for v in (0, 1, 2, 3):
            ers = F.EdgeRefinedSuperpixel(w, h)
            ers.set_variant(v)
            tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
            ers.EdgeRefining(tt(ref["sp_labels"]), tt(ref["dasp_labels"]), tt(depth), tt(bgr))
            got = ers.getRefinedDepth_Device().cpu().numpy()
            edge = ers.getEdgeStageDepth_Device().cpu().numpy()
            labels = ers.getRefinedLabels_Device().cpu().numpy()
            ers.close()
            s = stage.ers_stage_run(ref["sp_labels"], ref["dasp_labels"], depth, bgr, v)
            sf = stage.ers_stage_run(ref["sp_labels"], ref["dasp_labels"], depth, bgr, v, force_full_rules=True)
            t = time.time()
            st = O.ers_stage(edge, bgr, labels, avg_in=s["avg"], dev_in=s["dev"])
            r = O.stage_check(got, st)
            rec = dict(kernel="K10", case=tag, variant=v, stage_bits_equal=stage.bits_equal(s["depth"], got),
                       forced_bits_equal=stage.bits_equal(sf["depth"], got), tiles=s["counters"][4:].tolist(),
                       tiles_forced=sf["counters"][4:].tolist(), labels_exact=bool(np.array_equal(labels, ref["refined_labels"])),
                       bad=int(r["bad"].sum()), oracle_s=round(time.time() - t, 2), **brief(r))
            print(json.dumps(rec), flush=True)
            results.append(rec)

    k10("vga 15x20", 640, 480, 1, 15, 20)
    k10("qvga 6x8", 320, 240, 2, 6, 8)
    if args.fhd:
        k10("config5 1080p 15x20", 1920, 1080, 3, 15, 20)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(results, f, indent=1)
    nbad = sum(1 for r in results if r["bad"] or not r["stage_bits_equal"] or not r["forced_bits_equal"])
    print("cases", len(results), "failing", nbad)
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
