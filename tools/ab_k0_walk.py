#!/usr/bin/env python3
"""A/B of K0's tile walk (r05): 0 = linear, 1 = XCD bands, 2 = XCD bands with runs of four adjacent tiles per
workgroup.  The walk is forced with KDE_K0_BAND_WALK in the measurement build (tools/hooks/libkde_hip_ab.so; read once per
process, so every leg is a child process), K0 is timed INSIDE the headline step (K0 + K1 back to back on 64 x 640x480, window
11: its input is not cache-resident) and alone, on one and on eight 1080p frames; the smoothed bytes are hashed.
    python tools/ab_k0_walk.py [--rounds 3]        -> one JSON summary"""
import argparse
import json
import os
import subprocess
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "hooks"))


def child():
    import ab
    ab.use_ab_library()
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from wake import wake
    from kinectdepthmapenhancement_amd import filters as F, synth
    out = {}
    wake(torch, 150.0)
    for (w, h, n, distinct, tag) in ((640, 480, 64, 8, "vga64"), (1920, 1080, 1, 1, "fhd1"), (1920, 1080, 8, 2, "fhd8")):
        bgr, depth = synth.make_batch(500, distinct, w, h)
        reps = -(-n // distinct)
        color = torch.from_numpy(np.tile(bgr, (reps, 1, 1, 1))[:n]).cuda()
        d = torch.from_numpy(np.tile(depth, (reps, 1, 1))[:n]).cuda()
        p = F.JointBilateralFilter.default_params()
        p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma = 11, 3.0, 7.65, 20.0
        jbf = F.JointBilateralFilter(w, h, p, max_batch=n)
        smooth, res = torch.empty_like(color), torch.empty_like(d)
        for _ in range(5):
            jbf.presmooth_batch(color, smooth)
            jbf.filter_batch(d, smooth, res)
        iters = 30
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(iters)]
        for i in range(iters):
            ev[i][0].record()
            jbf.presmooth_batch(color, smooth)
            ev[i][1].record()
            jbf.filter_batch(d, smooth, res)
            ev[i][2].record()
        torch.cuda.synchronize()
        k0_in_step = float(np.median([e[0].elapsed_time(e[1]) for e in ev]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            jbf.presmooth_batch(color, smooth)
        e1.record()
        torch.cuda.synchronize()
        out[tag] = {"k0_ms_in_step": k0_in_step, "k0_ms_alone": e0.elapsed_time(e1) / iters,
                    "crc": zlib.crc32(smooth.cpu().numpy().tobytes())}
        jbf.close()
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child()
    modes = {"linear (0)": "0", "xcd bands (1)": "1", "xcd bands, runs of 4 (2)": "2"}
    res = {m: [] for m in modes}
    for _ in range(a.rounds):
        for m, v in modes.items():
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True,
                               env=dict(os.environ, KDE_K0_BAND_WALK=v))
            if r.returncode != 0:
                sys.exit(r.stderr[-2000:])
            res[m].append(json.loads(r.stdout.strip().splitlines()[-1]))
    summary = {}
    for m, runs in res.items():
        summary[m] = {tag: {k: (float(np.median([x[tag][k] for x in runs])) if k != "crc" else runs[0][tag][k]) for k in runs[0][tag]} for tag in runs[0]}
    first = next(iter(summary.values()))
    summary["bytes_identical_across_walks"] = all(summary[m][t]["crc"] == first[t]["crc"] for m in modes for t in first)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
