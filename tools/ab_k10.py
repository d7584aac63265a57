#!/usr/bin/env python3
"""A/B of K10's pass-1 label test (r04): the default enhance7_pk_kernel<true> (same-label test folded into the weight argument)
against r03's form (weight times a {0,1} mask, KDE_K10_MASK_PRODUCT=1).  The switch is read once per process, so every leg is a
child process; legs alternate over --rounds.  Timed: RegionGrowingBilateralFilter::Process (4 launches, K10 the largest) on one
1920x1080 frame and on a batch of 8, plus the outputs' CRC (the two forms must agree to the bit).
    python3 tools/ab_k10.py [--rounds 3]"""
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
import ab as _ab                                    # noqa: E402
_ab.use_ab_library()                                # both legs on tools/hooks/libkde_hip_ab.so: only that build has the switch


def child():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from wake import wake
    from kinectdepthmapenhancement_amd import filters as F, synth
    W, H = 1920, 1080
    K = synth.intrinsics(W, H)
    out = {}
    wake(torch, 150.0)
    for n in (1, 8):
        bgr, depth = synth.make_batch(77, n, W, H)
        color, d = torch.from_numpy(bgr).cuda(), torch.from_numpy(depth).cuda()
        conv = F.DimensionConvertor(); conv.setCameraParameters(K, W, H)
        pts = torch.empty((n, H, W, 3), dtype=torch.float32, device="cuda")
        conv.projectiveToReal(d, pts)
        rg = F.RegionGrowingBilateralFilter(W, H, max_batch=n); rg.SetParametor(15, 20, K)
        run = (lambda: rg.process_batch(d, pts, color)) if n > 1 else (lambda: rg.Process(d[0], pts[0], color[0]))
        for _ in range(5):
            run()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 / n)
        out[f"rgbf_ms_per_frame_batch{n}"] = float(np.median(ts))
        out[f"crc_batch{n}"] = zlib.crc32(rg.getRefinedDepth_Device().cpu().numpy().tobytes())
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child()
    legs = {"fused (default)": {}, "mask product (r03)": {"KDE_K10_MASK_PRODUCT": "1"}}
    res = {k: [] for k in legs}
    for _ in range(a.rounds):
        for name, env in legs.items():
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True, env=dict(os.environ, **env))
            if r.returncode != 0:
                sys.exit(r.stderr[-2000:])
            res[name].append(json.loads(r.stdout.strip().splitlines()[-1]))
    summary = {}
    for name, runs in res.items():
        summary[name] = {k: (float(np.median([x[k] for x in runs])) if k.startswith("rgbf") else runs[0][k]) for k in runs[0]}
    a_, b_ = summary["fused (default)"], summary["mask product (r03)"]
    summary["bit_identical"] = all(a_[k] == b_[k] for k in a_ if k.startswith("crc"))
    summary["gain_single_frame"] = b_["rgbf_ms_per_frame_batch1"] / a_["rgbf_ms_per_frame_batch1"] - 1.0
    summary["gain_batch8"] = b_["rgbf_ms_per_frame_batch8"] / a_["rgbf_ms_per_frame_batch8"] - 1.0
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
