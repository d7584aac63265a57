#!/usr/bin/env python3
"""A/B of K1's partial-keep variants (r05): for windows 9, 11, 13 every variant of the window in
interleaved rounds on one process (clock drift cancels), each output compared BIT FOR BIT with the window's default kernel.
One JSON line per (workload, variant) -> profiles/r05_sweep_k1_variants.log."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hooks"))
import ab as _ab                                    # noqa: E402
_ab.use_ab_library()                                # the keep variants exist in tools/hooks/libkde_hip_ab.so only


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", default="9,11,13")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from wake import wake
    from kinectdepthmapenhancement_amd import filters, synth
    names = filters.JointBilateralFilter.variants()
    for (w, h, n, distinct) in ((640, 480, 64, 8), (1920, 1080, 16, 2)):
        bgr, depth = synth.make_batch(500, distinct, w, h)
        reps = -(-n // distinct)
        color = torch.from_numpy(np.tile(bgr, (reps, 1, 1, 1))[:n]).cuda()
        d = torch.from_numpy(np.tile(depth, (reps, 1, 1))[:n]).cuda()
        pre = filters.JointBilateralFilter(w, h, max_batch=n)
        guide = torch.empty_like(color)
        pre.presmooth_batch(color, guide)           # K1 runs on the K0-smoothed guide, as in Process
        pre.close()
        for win in [int(x) for x in a.windows.split(",")]:
            p = filters.JointBilateralFilter.default_params()
            p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma, p.presmooth = win, 3.0, 7.65, 20.0, 0
            jbf = filters.JointBilateralFilter(w, h, p, max_batch=n)
            todo = [v for v, nm in enumerate(names) if v and nm.startswith(f"w{win}-") and ("-pk" in nm) and "-v4" in nm and "noelide" not in nm]
            outs, times = {}, {v: [] for v in todo}
            wake(torch, 150.0)
            for rnd in range(a.rounds + 1):
                for v in todo:
                    jbf.set_variant(v)
                    out = torch.empty_like(d)
                    jbf.filter_batch(d, guide, out)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.iters):
                        jbf.filter_batch(d, guide, out)
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:
                        times[v].append(e0.elapsed_time(e1) / a.iters)
                    else:
                        outs[v] = out.view(torch.int32).clone()
            base = todo[0]
            ms0 = float(np.median(times[base]))
            for v in todo:
                ms = float(np.median(times[v]))
                print(json.dumps({"size": f"{n}x{w}x{h}", "window": win, "variant": names[v], "default": v == base, "ms_median": round(ms, 4),
                                  "ms_min": round(min(times[v]), 4), "vs_default": round(ms / ms0, 4),
                                  "bit_identical_to_default": bool(torch.equal(outs[v], outs[base]))}), flush=True)
            jbf.close()


if __name__ == "__main__":
    main()
