#!/usr/bin/env python3
"""MarkovRandomField::Process (row f1) on a resident batch: Mpixel/s and the HBM fraction at 11 B/pixel."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--wakeup-ms", type=float, default=150.0, help="untimed load before anything is measured (tools/wake.py)")
    a = ap.parse_args()
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from wake import wake
    from kinectdepthmapenhancement_amd import filters as F, synth
    bgr, depth = synth.make_batch(500, min(a.frames, 2), a.width, a.height)
    reps = -(-a.frames // bgr.shape[0])
    color = torch.from_numpy(np.tile(bgr, (reps, 1, 1, 1))[:a.frames]).cuda()
    d = torch.from_numpy(np.tile(depth, (reps, 1, 1))[:a.frames]).cuda()
    out = torch.empty_like(d)
    res = {"size": f"{a.width}x{a.height}x{a.frames}"}
    px = a.frames * a.width * a.height
    wake(torch, a.wakeup_ms)
    for name, kw in (("reference_constants_w5", {}), ("generic_kernel_w7", {"window": 7})):
        mrf = F.MarkovRandomField(a.width, a.height, max_batch=a.frames, **kw)
        mrf.process_batch(d, color, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            mrf.process_batch(d, color, out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        res[name] = {"ms": ms, "mpix_s": px / ms / 1e3, "hbm_frac": 11.0 * px / (ms * 1e-3) / 8e12}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
